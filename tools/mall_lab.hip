// Does the 256 MB Infinity Cache keep a chunk a kernel has just WRITTEN (or read) for the next kernel?  Producer / consumer
// pairs of the V-cycle (step 1 -> step 2, pre-smoothing -> residual) could be launched chunk by chunk if it does.
//   write X (S MB) | stream T MB of other data (read + write) | read X, timed -> GB/s against a cold read
// hipcc --offload-arch=gfx950 -O3 tools/mall_lab.hip -o tools/mall_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_write(double *x, size_t n, double v) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) x[i] = v + (double)i; }
__global__ void k_copy(const double *a, double *b, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) b[i] = a[i] * 1.0000001; }
__global__ void k_read(const double *x, size_t n, double *out)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) s += x[i];
    if (s == 123.456) out[0] = s;
}
int main()
{
    const size_t MB = 1 << 20;
    double *X, *Y, *Z, *out;
    CK(hipMalloc(&X, 1024 * MB));
    CK(hipMalloc(&Y, 2048 * MB));
    CK(hipMalloc(&Z, 2048 * MB));
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int G = 8192;
    for (int prod = 0; prod < 2; ++prod)       // 0: X written before, 1: X read before
        for (size_t S : {16, 32, 64, 128, 256, 512})
            for (size_t T : {0, 64, 128, 256, 1024}) {
                const size_t ns = S * MB / 8, nt = T * MB / 8 / 2;
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, 0, Y + (1024 * MB / 8), Z + (1024 * MB / 8), 512 * MB / 8);  // flush: 1 GB of other traffic
                    if (prod == 0) hipLaunchKernelGGL(k_write, dim3(G), dim3(256), 0, 0, X, ns, 1.0);
                    else hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, X, ns, out);
                    if (nt) hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, 0, Y, Z, nt);
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, X, ns, out);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                printf("%s X %4zu MB, then %4zu MB of other traffic: read X at %7.1f GB/s (%.3f ms)\n", prod ? "read " : "wrote", S, T, S * MB / 1e9 / (best * 1e-3), best);
            }
    return 0;
}
