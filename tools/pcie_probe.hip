// Developer probe: can this box move H2D and D2H at the same time, and what does a device->host copy (a blit kernel
// on this runtime: __amd_rocclr_copyBuffer, 256 workgroups x 512 threads) cost the kernels that run beside it?
// hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_stream(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        v.x ^= v.y;
        out[i] = v;
    }
}
__global__ void k_tiny(uint32_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

static int kernels_beside_copy(const char* what, hipStream_t copy_stream, void* h_down, void* d_down, size_t down, void* h_up, void* d_up,
                               size_t up, hipStream_t up_stream, uint4* a, uint4* b, size_t n16, hipStream_t ks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // kernel alone
    float alone = 0;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0, ks));
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, ks, a, b, n16);
        CK(hipEventRecord(e1, ks));
        CK(hipStreamSynchronize(ks));
        CK(hipEventElapsedTime(&alone, e0, e1));
    }
    double t0 = now();
    if (up_stream) CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, up_stream));
    CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, copy_stream));
    std::vector<float> ms;
    int tiny = 0;
    while (hipStreamQuery(copy_stream) == hipErrorNotReady) {
        CK(hipEventRecord(e0, ks));
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, ks, a, b, n16);
        CK(hipEventRecord(e1, ks));
        CK(hipStreamSynchronize(ks));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        for (int q = 0; q < 10; q++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, ks, (uint32_t*)b);
        CK(hipStreamSynchronize(ks));
        tiny += 10;
    }
    CK(hipStreamSynchronize(copy_stream));
    if (up_stream) CK(hipStreamSynchronize(up_stream));
    double t1 = now();
    float sum = 0, mx = 0;
    for (float t : ms) sum += t, mx = t > mx ? t : mx;
    printf("%-28s copy%s done in %.3f ms (%.1f GB/s down); HBM kernel alone %.3f ms, beside the copy: %zu launches, mean %.3f max %.3f ms (+ %d tiny kernels)\n",
           what, up_stream ? "+upload" : "", t1 - t0, down / (t1 - t0) * 1e-6, alone, ms.size(), ms.empty() ? 0.f : sum / ms.size(), mx, tiny);
    return 0;
}

int main() {
    const size_t up = 189600004, down = 168221232;
    void *h_up, *h_down, *d_up, *d_down;
    CK(hipHostMalloc(&h_up, up, hipHostMallocDefault));
    CK(hipHostMalloc(&h_down, down, hipHostMallocDefault));
    CK(hipMalloc(&d_up, up)); CK(hipMalloc(&d_down, down));
    memset(h_up, 1, up); memset(h_down, 2, down);
    CK(hipMemset(d_down, 3, down));
    hipStream_t s1, s2, ks;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double t1 = now();
        CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s2));
        double t2 = now();
        CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        double t3 = now();
        printf("rep %d: h2d %.3f ms (%.1f GB/s)  d2h %.3f ms (%.1f GB/s)  both %.3f ms\n", rep,
               t1 - t0, up / (t1 - t0) * 1e-6, t2 - t1, down / (t2 - t1) * 1e-6, t3 - t2);
    }
    // the same device->host bytes in 8 pieces, queued back to back on one stream, from a buffer a kernel has just written
    {
        double t0 = now();
        for (int k = 0; k < 8; k++) {
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, ks, (uint32_t*)d_down);
            CK(hipStreamSynchronize(ks));
            CK(hipMemcpyAsync((char*)h_down + down / 8 * k, (char*)d_down + down / 8 * k, down / 8, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s2));
        double t1 = now();
        printf("d2h in 8 pieces of %.1f MB: %.3f ms (%.1f GB/s)\n", down / 8 * 1e-6, t1 - t0, down / (t1 - t0) * 1e-6);
    }
    // an HBM-bound kernel (64 MB in, 64 MB out) next to the copies
    const size_t n16 = (64u << 20) / 16;
    uint4 *a, *b;
    CK(hipMalloc(&a, n16 * 16)); CK(hipMalloc(&b, n16 * 16));
    CK(hipMemset(a, 5, n16 * 16));
    if (kernels_beside_copy("plain copy stream", s2, h_down, d_down, down, h_up, d_up, up, nullptr, a, b, n16, ks)) return 1;
    if (kernels_beside_copy("plain copy stream", s2, h_down, d_down, down, h_up, d_up, up, s1, a, b, n16, ks)) return 1;
    // copy streams confined to a few CUs
    const int masks[] = {8, 16, 32, 64};
    for (int ncu : masks) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 256 CUs; spread the chosen ones over the XCDs (CU i of the mask <-> XCD i % 8 is a guess: just take every (256/ncu)-th)
        for (int i = 0; i < ncu; i++) {
            const int cu = i * (256 / ncu);
            mask[cu / 32] |= 1u << (cu % 32);
        }
        hipStream_t ms;
        if (hipExtStreamCreateWithCUMask(&ms, 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); (void)hipGetLastError(); continue; }
        char name[64];
        snprintf(name, sizeof name, "copy stream on %d CUs", ncu);
        if (kernels_beside_copy(name, ms, h_down, d_down, down, h_up, d_up, up, nullptr, a, b, n16, ks)) return 1;
        if (kernels_beside_copy(name, ms, h_down, d_down, down, h_up, d_up, up, s1, a, b, n16, ks)) return 1;
        CK(hipStreamDestroy(ms));
    }
    return 0;
}
