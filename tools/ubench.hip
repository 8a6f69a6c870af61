// Measurement infrastructure (not product): the three hardware rates the overlap kernels are priced against,
// measured on the box instead of taken from a table.
//
//   valu   independent integer VALU streams (v_alignbit_b32 / v_xor_b32 / v_lshrrev_b32: the instructions the scan
//          filter and the verify compare consist of) at 1, 2, 4, 8 waves per SIMD -> wave-instructions per second
//          per SIMD, i.e. cycles per wave64 instruction at the clock the chip holds under that load
//   lds    ds_read_b64 at random 8-byte-aligned addresses of a 128 KB region (the scan's filter lookups)
//   l2     global_load_dwordx4 over a footprint that stays in the XCD's L2 (the verify's b side)
//   lines  global_load_dwordx4 at RANDOM 64-byte lines of a table of 4 MB .. 2 GB, one line per lane (the wide index's
//          probe: one slot per word of a) or one line per four lanes (the narrow table's group probe) -> lines per second
//   stream a plain 16-byte-per-lane sweep (the byte count FETCH_SIZE is calibrated against, tools/fetch_calib.sh)
//
// Occupancy is pinned with dynamic LDS: a 256-thread workgroup (one wave per SIMD) that asks for 160 KB / w of LDS
// can only share its CU with w - 1 others.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return -1.0; } } while (0)

extern __shared__ uint64_t dyn_lds[];

template <int OP>
__global__ __launch_bounds__(256) void k_valu(uint32_t* out, int iters, uint32_t sh) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5,
             a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    const uint32_t c = blockIdx.x * 2654435761u + 12345u;
    const uint64_t m64 = 0x5555555555555555ull ^ (uint64_t)sh;
    if (threadIdx.x == 1023) dyn_lds[0] = 0;  // (keeps the dynamic LDS request alive)
    for (int i = 0; i < iters; ++i) {
#define STEP(r)                                                                                              \
    if (OP == 0) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(sh));                  \
    else if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    else if (OP == 2) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(r) : "v"(sh));                          \
    else if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r) : "v"(c));                       \
    else if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "s"(sh));             \
    else if (OP == 5) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(sh));                 \
    else if (OP == 6) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(sh));               \
    else if (OP == 7) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(c));                        \
    else if (OP == 8) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(r));                                     \
    else if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(sh));                 \
    else if (OP == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(sh));                \
    else if (OP == 11) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(r) : "v"(c), "v"(sh));  \
    else if (OP == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "s"(sh));                             \
    else if (OP == 13) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "s"(sh));                \
    else if (OP == 14) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(sh));             \
    else if (OP == 15) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));   \
    else if (OP == 16) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));  \
    else if (OP == 17) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(c)); \
    else if (OP == 18) asm volatile("v_add_u32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(c)); \
    else if (OP == 19) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    else if (OP == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : "vcc");             \
    else if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(c), "s"(m64));         \
    else if (OP == 22) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r) : : "vcc");                 \
    else if (OP == 23) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(r) : "v"(c) : "vcc"); \
    else if (OP == 24) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    else if (OP == 25) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    else if (OP == 26) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    else if (OP == 27) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(c));                                \
    else if (OP == 28) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    else if (OP == 29) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r) : "v"(sh));                          \
    else if (OP == 30) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(c));                                   \
    else asm volatile("v_min_i32 %0, %0, %1" : "+v"(r) : "v"(c));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
#undef STEP
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ __launch_bounds__(256) void k_lds(uint32_t* out, int iters, uint32_t region_mask) {
    uint32_t* s = reinterpret_cast<uint32_t*>(dyn_lds);
    for (uint32_t i = threadIdx.x; i <= region_mask / 4; i += blockDim.x) s[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = (threadIdx.x + 1) * 2654435761u + blockIdx.x;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint64_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // eight independent reads in flight, addresses from a per-lane LCG
            x = x * 1664525u + 1013904223u;
            v[k] = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(s) + ((x >> 8) & region_mask & ~7u));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_l2(const u32x4* __restrict__ buf, uint32_t* out, int iters, uint32_t per_xcd_vecs) {
    // workgroup i runs on XCD i mod 8: every XCD sweeps its own slice again and again
    const u32x4* base = buf + (size_t)(blockIdx.x & 7u) * per_xcd_vecs;
    uint32_t idx = ((blockIdx.x >> 3) * 256u + threadIdx.x) % per_xcd_vecs;
    u32x4 acc = {0, 0, 0, 0};
    if (threadIdx.x == 1023) dyn_lds[0] = 0;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = base[idx];
            idx += 256u * 61u;
            if (idx >= per_xcd_vecs) idx -= per_xcd_vecs;
            if (idx >= per_xcd_vecs) idx %= per_xcd_vecs;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}


// ---- random 64-byte lines (VERDICT r3 #3a / #4) ---------------------------------------------------------------------
// Every lane draws line numbers from its own xorshift stream and loads 16 bytes of the line; LPL lanes share a line
// (LPL = 1: 64 lines per wave-instruction, what k_wide_scan's table probe issues; LPL = 4: 16 lines per
// wave-instruction, the four slots of a group in k_scan_probe).  DEPTH independent loads are in flight per lane.
template <int LPL, int DEPTH>
__global__ __launch_bounds__(256) void k_rand_lines(const u32x4* __restrict__ table, uint64_t n_lines, uint32_t* out, int iters) {
    const uint32_t gl = (blockIdx.x * 256u + threadIdx.x) / LPL;   // lanes of one group draw the same numbers
    uint64_t x = 0x9E3779B97F4A7C15ull * (gl + 1);
    const uint32_t sub = threadIdx.x % LPL;
    u32x4 acc = {0, 0, 0, 0};
    if (threadIdx.x == 1023) dyn_lds[0] = 0;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            const uint64_t line = (uint64_t)(((__uint128_t)(x & 0xFFFFFFFFFFFFull) * n_lines) >> 48);
            v[k] = table[line * 4 + sub];
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc ^= v[k];
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// 16 bytes per lane, grid-stride, `passes` sweeps over n_vecs vectors
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ buf, uint64_t n_vecs, uint32_t* out, int passes) {
    u32x4 acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (int p = 0; p < passes; ++p)
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_vecs; i += stride) acc ^= buf[i];
    out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

static double time_ms(hipEvent_t e0, hipEvent_t e1) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

extern "C" {

// wave-instructions per second per SIMD (1024 SIMDs); op: see the STEP macro
double ub_valu(int op, int waves_per_simd, int iters) {
    const int n_cu = 256, rounds = 4;
    const int grid = n_cu * waves_per_simd * rounds;
    const size_t lds = (size_t)(160 * 1024) / waves_per_simd - 1024;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    void (*kerns[])(uint32_t*, int, uint32_t) = {k_valu<0>, k_valu<1>, k_valu<2>, k_valu<3>, k_valu<4>, k_valu<5>, k_valu<6>, k_valu<7>,
                                                 k_valu<8>, k_valu<9>, k_valu<10>, k_valu<11>, k_valu<12>, k_valu<13>, k_valu<14>,
                                                 k_valu<15>, k_valu<16>, k_valu<17>, k_valu<18>, k_valu<19>, k_valu<20>,
                                                 k_valu<21>, k_valu<22>, k_valu<23>, k_valu<24>, k_valu<25>, k_valu<26>, k_valu<27>,
                                                 k_valu<28>, k_valu<29>, k_valu<30>, k_valu<31>};
    if (op < 0 || op > 31) return -1.0;
    auto kern = kerns[op];
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, iters / 8, 7u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, iters, 7u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    const double ms = time_ms(e0, e1);
    (void)hipFree(out);
    const double insts = (double)grid * 4.0 * (double)iters * 16.0;  // wave-instructions
    return insts / (ms * 1e-3) / 1024.0;
}

// bytes per second, whole chip: random ds_read_b64
double ub_lds(int waves_per_simd, int iters) {
    const int n_cu = 256, rounds = 4;
    const int grid = n_cu * waves_per_simd * rounds;
    size_t lds = (size_t)(160 * 1024) / waves_per_simd - 1024;
    uint32_t mask = 1;
    while ((size_t)(mask + 1) * 2 <= lds) mask = mask * 2 + 1;  // largest power of two that fits
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), lds, 0, out, iters / 8, mask);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), lds, 0, out, iters, mask);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    const double ms = time_ms(e0, e1);
    (void)hipFree(out);
    return (double)grid * 256.0 * (double)iters * 8.0 * 8.0 / (ms * 1e-3);
}

// bytes per second, whole chip: 16-byte loads over `per_xcd_bytes` per XCD (<= 2 MB stays in its L2)
double ub_l2(int waves_per_simd, int iters, uint32_t per_xcd_bytes) {
    const int n_cu = 256, rounds = 4;
    const int grid = n_cu * waves_per_simd * rounds;
    const size_t lds = (size_t)(160 * 1024) / waves_per_simd - 1024;
    const uint32_t per_xcd_vecs = per_xcd_bytes / 16;
    u32x4* buf = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&buf, (size_t)per_xcd_bytes * 8));
    CHECK(hipMemset(buf, 1, (size_t)per_xcd_bytes * 8));
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_l2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_l2, dim3(grid), dim3(256), lds, 0, buf, out, iters / 8, per_xcd_vecs);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_l2, dim3(grid), dim3(256), lds, 0, buf, out, iters, per_xcd_vecs);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    const double ms = time_ms(e0, e1);
    (void)hipFree(buf);
    (void)hipFree(out);
    return (double)grid * 256.0 * (double)iters * 8.0 * 16.0 / (ms * 1e-3);
}


// random 64-byte lines per second, whole chip: table_bytes of table, lanes_per_line 1 or 4, depth 4 or 8 loads in flight
double ub_rand_lines(uint64_t table_bytes, int waves_per_simd, int iters, int lanes_per_line, int depth) {
    const int n_cu = 256, rounds = 4;
    const int grid = n_cu * waves_per_simd * rounds;
    const size_t lds = (size_t)(160 * 1024) / waves_per_simd - 1024;
    u32x4* buf = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&buf, table_bytes));
    CHECK(hipMemset(buf, 1, table_bytes));
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    void (*kern)(const u32x4*, uint64_t, uint32_t*, int) =
        lanes_per_line == 4 ? (depth == 4 ? k_rand_lines<4, 4> : k_rand_lines<4, 8>) : (depth == 4 ? k_rand_lines<1, 4> : k_rand_lines<1, 8>);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t n_lines = table_bytes / 64;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, buf, n_lines, out, iters / 4 + 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, buf, n_lines, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    const double ms = time_ms(e0, e1);
    (void)hipFree(buf);
    (void)hipFree(out);
    const int dd = depth == 4 ? 4 : 8;
    return (double)grid * 256.0 / (lanes_per_line == 4 ? 4.0 : 1.0) * (double)iters * dd / (ms * 1e-3);
}

}  // extern "C"

#ifdef UBENCH_MAIN
// FETCH_SIZE calibration (tools/fetch_calib.sh): every pattern once, each its own kernel NAME, with the byte count the
// pattern requests printed beside it -- `rocprofv3 --pmc FETCH_SIZE` of this program gives the counter per kernel.
template <int TAG>
__global__ __launch_bounds__(256) void k_cal_stream(const u32x4* __restrict__ buf, uint64_t n_vecs, uint32_t* out, int passes) {
    u32x4 acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (int p = 0; p < passes; ++p)
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_vecs; i += stride) acc ^= buf[i];
    out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
template <int TAG, int LPL>
__global__ __launch_bounds__(256) void k_cal_lines(const u32x4* __restrict__ table, uint64_t n_lines, uint32_t* out, int iters) {
    const uint32_t gl = (blockIdx.x * 256u + threadIdx.x) / LPL;
    uint64_t x = 0x9E3779B97F4A7C15ull * (gl + 1);
    const uint32_t sub = threadIdx.x % LPL;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            const uint64_t line = (uint64_t)(((__uint128_t)(x & 0xFFFFFFFFFFFFull) * n_lines) >> 48);
            v[k] = table[line * 4 + sub];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const uint64_t big = 2ull << 30;
    u32x4* buf = nullptr;
    uint32_t* out = nullptr;
    const int grid = 256 * 8;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\n");
    auto report = [&](const char* kernel, const char* what, double requests, double bytes_per_request, bool last) {
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf(" \"%s\": {\"pattern\": \"%s\", \"requests\": %.0f, \"bytes_requested\": %.0f, \"ms\": %.4f, \"requests_per_s\": %.4g, \"GBps\": %.1f}%s\n",
               kernel, what, requests, requests * bytes_per_request, ms, requests / (ms * 1e-3), requests * bytes_per_request / (ms * 1e-3) / 1e9, last ? "" : ",");
        return 0;
    };
    // (a) streaming 16 B per lane over 2 GiB, once: 2 GiB, nothing resident
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_cal_stream<0>, dim3(grid), dim3(256), 0, 0, buf, big / 16, out, 1);
    if (report("k_cal_stream<0>", "stream 16 B/lane, 2 GiB once (HBM)", (double)(big / 64), 64.0, false)) return 1;
    // (b) streaming over 64 MiB, 32 passes: the Infinity Cache holds it after the first pass, the L2s (32 MiB) do not
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_cal_stream<1>, dim3(grid), dim3(256), 0, 0, buf, (64ull << 20) / 16, out, 32);
    if (report("k_cal_stream<1>", "stream 16 B/lane, 64 MiB x 32 passes (Infinity Cache)", 32.0 * (64ull << 20) / 64, 64.0, false)) return 1;
    // (c..f) random 64-byte lines, one lane per line (16 B of it loaded), tables of 2 GiB / 512 MiB / 64 MiB / 8 MiB
    const int iters = 64;
    const double n_req = (double)grid * 256.0 * iters * 8.0;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_cal_lines<0, 1>), dim3(grid), dim3(256), 0, 0, buf, big / 64, out, iters);
    if (report("k_cal_lines<0, 1>", "random 64-B lines, 1 lane per line, 2 GiB table (HBM)", n_req, 64.0, false)) return 1;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_cal_lines<1, 1>), dim3(grid), dim3(256), 0, 0, buf, (512ull << 20) / 64, out, iters);
    if (report("k_cal_lines<1, 1>", "random 64-B lines, 1 lane per line, 512 MiB table", n_req, 64.0, false)) return 1;
    // (warm the 64 MiB table into the Infinity Cache first: the warm-up has its own kernel name)
    hipLaunchKernelGGL(k_cal_stream<2>, dim3(grid), dim3(256), 0, 0, buf, (64ull << 20) / 16, out, 2);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_cal_lines<2, 1>), dim3(grid), dim3(256), 0, 0, buf, (64ull << 20) / 64, out, iters);
    if (report("k_cal_lines<2, 1>", "random 64-B lines, 1 lane per line, 64 MiB table (Infinity Cache)", n_req, 64.0, false)) return 1;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_cal_lines<3, 1>), dim3(grid), dim3(256), 0, 0, buf, (8ull << 20) / 64, out, iters);
    if (report("k_cal_lines<3, 1>", "random 64-B lines, 1 lane per line, 8 MiB table (half of it in each XCD's L2)", n_req, 64.0, false)) return 1;
    // (g, h) four lanes per line (the narrow scan's group probe): 8 MiB and 64 MiB
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_cal_lines<4, 4>), dim3(grid), dim3(256), 0, 0, buf, (8ull << 20) / 64, out, iters);
    if (report("k_cal_lines<4, 4>", "random 64-B lines, 4 lanes per line, 8 MiB table", n_req / 4, 64.0, false)) return 1;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_cal_lines<5, 4>), dim3(grid), dim3(256), 0, 0, buf, (64ull << 20) / 64, out, iters);
    if (report("k_cal_lines<5, 4>", "random 64-B lines, 4 lanes per line, 64 MiB table", n_req / 4, 64.0, true)) return 1;
    printf("}\n");
    return 0;
}
#endif
