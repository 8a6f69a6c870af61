// Developer probe: does the KIND of page-locked host memory change the host->device rate?
// The packed read stores are page-aligned malloc + hipHostRegister (c_api.hip: RegAlloc); the streamed step uploads them in
// ~11 pieces with an event behind each.  Here: the same byte count in the same pieces out of (a) hipHostMalloc memory,
// (b) malloc + hipHostRegister, (c) the same with 2 MB alignment and MADV_HUGEPAGE before the first touch, (d) hipHostMalloc
// with the non-coherent flag.  Prints GB/s (first copy starts -> last piece landed, device events), best of 5.
//   hipcc --offload-arch=gfx950 -O2 tools/h2d_source_probe.hip -o /tmp/h2d_source_probe && /tmp/h2d_source_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(const char* what, const char* src, char* dst, size_t bytes, int pieces, hipStream_t st) {
    hipEvent_t e0, e1, ep[32];
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int k = 0; k < pieces; ++k) CK(hipEventCreate(&ep[k]));
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < pieces; ++k) {
            const size_t lo = bytes / pieces * k, hi = k + 1 == pieces ? bytes : bytes / pieces * (k + 1);
            CK(hipMemcpyAsync(dst + lo, src + lo, hi - lo, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ep[k], st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    printf("%-64s %d pieces: %.3f ms  %.1f GB/s\n", what, pieces, best, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main() {
    const size_t bytes = 188u << 20;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char* dst = nullptr;
    CK(hipMalloc(&dst, bytes));
    // (a)
    char* a = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&a), bytes, hipHostMallocDefault));
    std::memset(a, 1, bytes);
    // (b)
    void* b = nullptr;
    if (posix_memalign(&b, 4096, bytes)) return 1;
    std::memset(b, 2, bytes);
    CK(hipHostRegister(b, bytes, hipHostRegisterPortable));
    // (c)
    void* c = nullptr;
    if (posix_memalign(&c, 2u << 20, bytes)) return 1;
    (void)madvise(c, bytes, MADV_HUGEPAGE);
    std::memset(c, 3, bytes);
    CK(hipHostRegister(c, bytes, hipHostRegisterPortable));
    // (d)
    char* d = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&d), bytes, hipHostMallocNonCoherent));
    std::memset(d, 4, bytes);
    for (int pieces : {1, 11}) {
        if (run("hipHostMalloc (default)", a, dst, bytes, pieces, st)) return 1;
        if (run("malloc + hipHostRegister (what the stores are)", static_cast<char*>(b), dst, bytes, pieces, st)) return 1;
        if (run("2 MB aligned + MADV_HUGEPAGE + hipHostRegister", static_cast<char*>(c), dst, bytes, pieces, st)) return 1;
        if (run("hipHostMalloc (non-coherent)", d, dst, bytes, pieces, st)) return 1;
    }
    return 0;
}
