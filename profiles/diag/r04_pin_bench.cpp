// How should the host stage a few GB for one upload?  hipHostMalloc (what hypo_gpu_host_alloc does) against malloc + first touch on
// all threads + hipHostRegister, against a copy straight out of pageable memory.  (C4 at size: page-locking 3.9 GB of staging took
// 0.84 s of the 0.89 s "flatten".)   hipcc -O2 -fopenmp r04_pin_bench.cpp
#include <hip/hip_runtime.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? atoi(argv[1]) : 2;
    const size_t n = gb << 30;
    void* d = nullptr; CK(hipMalloc(&d, n));
    CK(hipMemset(d, 0, n)); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void* h = nullptr; CK(hipHostMalloc(&h, n, hipHostMallocDefault));
        double t1 = now();
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i += 4096) ((char*)h)[i] = 1;
        double t2 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t3 = now();
        CK(hipHostFree(h));
        double t4 = now();
        printf("hipHostMalloc %zu GB: alloc %.3f s, touch %.3f s, H2D %.3f s (%.1f GB/s), free %.3f s\n", gb, t1 - t0, t2 - t1, t3 - t2, gb * 1.074 / (t3 - t2), t4 - t3);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void* h = nullptr; CK(hipHostMalloc(&h, n, hipHostMallocNonCoherent));
        double t1 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t2 = now();
        CK(hipHostFree(h));
        printf("hipHostMalloc NonCoherent %zu GB: alloc %.3f s, H2D %.3f s (%.1f GB/s)\n", gb, t1 - t0, t2 - t1, gb * 1.074 / (t2 - t1));
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void* h = nullptr; if (posix_memalign(&h, 1 << 21, n)) return 1;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i += 4096) ((char*)h)[i] = 1;
        double t1 = now();
        CK(hipHostRegister(h, n, hipHostRegisterDefault));
        double t2 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t3 = now();
        CK(hipHostUnregister(h));
        double t4 = now();
        free(h);
        printf("malloc + touch on %d threads + hipHostRegister %zu GB: touch %.3f s, register %.3f s, H2D %.3f s (%.1f GB/s), unregister %.3f s\n", omp_get_max_threads(), gb, t1 - t0, t2 - t1, t3 - t2, gb * 1.074 / (t3 - t2), t4 - t3);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void* h = nullptr; if (posix_memalign(&h, 1 << 21, n)) return 1;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i += 4096) ((char*)h)[i] = 1;
        double t1 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t2 = now();
        free(h);
        printf("pageable %zu GB: touch %.3f s, H2D %.3f s (%.1f GB/s)\n", gb, t1 - t0, t2 - t1, gb * 1.074 / (t2 - t1));
    }
    {   // register in pieces on several threads
        void* h = nullptr; if (posix_memalign(&h, 1 << 21, n)) return 1;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i += 4096) ((char*)h)[i] = 1;
        const int P = 8; const size_t piece = n / P;
        double t1 = now();
        int bad = 0;
#pragma omp parallel for num_threads(P) reduction(| : bad)
        for (int p = 0; p < P; ++p) bad |= hipHostRegister((char*)h + p * piece, piece, hipHostRegisterDefault) != hipSuccess;
        double t2 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t3 = now();
        printf("hipHostRegister in %d pieces side by side: %.3f s (bad %d), H2D %.3f s (%.1f GB/s)\n", P, t2 - t1, bad, t3 - t2, gb * 1.074 / (t3 - t2));
        for (int p = 0; p < P; ++p) (void)hipHostUnregister((char*)h + p * piece);
        free(h);
    }
    return 0;
}
