// Measurement infrastructure (not product): the three hardware rates the overlap kernels are priced against,
// measured on the box instead of taken from a table.
//
//   valu   independent integer VALU streams (v_alignbit_b32 / v_xor_b32 / v_lshrrev_b32: the instructions the scan
//          filter and the verify compare consist of) at 1, 2, 4, 8 waves per SIMD -> wave-instructions per second
//          per SIMD, i.e. cycles per wave64 instruction at the clock the chip holds under that load
//   lds    ds_read_b64 at random 8-byte-aligned addresses of a 128 KB region (the scan's filter lookups)
//   l2     global_load_dwordx4 over a footprint that stays in the XCD's L2 (the verify's b side)
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

}  // extern "C"
