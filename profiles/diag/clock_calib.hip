// clock_calib.hip — what clock64() counts on this GPU: ticks per microsecond (against wall_clock64, 100 MHz), and the latency of
// a dependent LDS read and a dependent global read in those ticks.  hipcc --offload-arch=gfx950 -O3 -o clock_calib clock_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void calib(unsigned long long* out, const int* chain, int n) {
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 17 + 1) & 1023;
    __syncthreads();
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    int x = threadIdx.x;
    for (int i = 0; i < n; ++i) x = lds[x];
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    int y = threadIdx.x;
    for (int i = 0; i < n; ++i) y = chain[y];
    unsigned long long c2 = clock64(), w2 = wall_clock64();
    int v = x;
    for (int i = 0; i < n; ++i) v = v * 3 + 1;          // dependent VALU
    unsigned long long c3 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = c2 - c1; out[3] = w2 - w1; out[4] = c3 - c2; out[5] = x + y + v; }
}
int main() {
    const int n = 20000, N = 1 << 24;
    int* h = (int*)malloc(N * 4);
    for (long i = 0; i < N; ++i) h[i] = (int)((i * 1000003L + 12345) & (N - 1));
    int* d; unsigned long long* o; hipMalloc(&d, N * 4); hipMalloc(&o, 64);
    hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib, dim3(1), dim3(64), 0, 0, o, d, n);
        unsigned long long r[6]; hipMemcpy(r, o, 48, hipMemcpyDeviceToHost);
        printf("LDS chain: %.1f clock64 ticks/read, %.1f ns/read -> %.0f ticks/us | global chain: %.1f ticks/read, %.1f ns/read | dependent VALU op: %.2f ticks\n",
               (double)r[0] / n, (double)r[1] * 10.0 / n, (double)r[0] / ((double)r[1] / 100.0), (double)r[2] / n, (double)r[3] * 10.0 / n, (double)r[4] / n);
    }
    return 0;
}
