// tools/skew_lab.hip -- does the RELATIVE placement of the streams of a vector kernel decide its rate?  (round 5)
// OpUpdateP of krylov.hip (x += a p ; p = z + b p: three reads, two writes of 16-byte packs, 1024 packs per workgroup) on
// n = 512^3 doubles, the three vectors inside ONE allocation at  z = base, p = base + S, x = base + 2 S  for a list of
// strides S = n * 8 + skew, and in three separate allocations.  Every case: 5 launches, the best and the mean.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/skew_lab.hip -o tools/skew_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_update(long long ng, const v2 *__restrict__ z, v2 *__restrict__ p, v2 *__restrict__ x, double a, double b)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) {
            const v2 vz = z[i], vp = p[i];
            v2 vx = x[i];
            vx += a * vp;
            x[i] = vx;
            p[i] = vz + b * vp;
        }
    }
}
// two reads, one write + one more read/write pair: the first march's pattern is not a flat stream; this is the flat twin of
// k_presmooth2<0,1> (r_old, w read; r_new, x2 written)
__global__ __launch_bounds__(256) void k_rw(long long ng, const v2 *__restrict__ a, const v2 *__restrict__ b, v2 *__restrict__ c, v2 *__restrict__ d, double s)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) {
            const v2 va = a[i], vb = b[i];
            const v2 r = va - s * vb;
            c[i] = r;
            d[i] = 0.9 * r;
        }
    }
}
static double run(int which, long long n, double *v0, double *v1, double *v2p, double *v3, double *best)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long ng = n / 2;
    const unsigned nb = (unsigned)((ng + 1023) / 1024);
    double sum = 0.0;
    *best = 1e30;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        if (which == 0) hipLaunchKernelGGL(k_update, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)v0, (v2 *)v1, (v2 *)v2p, 1e-3, 0.5);
        else hipLaunchKernelGGL(k_rw, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)v0, (const v2 *)v1, (v2 *)v2p, (v2 *)v3, 1e-3);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r == 0) continue;
        sum += ms;
        *best = std::min(*best, (double)ms);
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return sum / 5.0;
}
int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 134217728LL;
    const long long M = 1 << 20;
    const long long skews[] = {0, 4096, 2 * M, 3 * M, 4 * M, 6 * M, 8 * M, 10 * M, 16 * M, 18 * M, 32 * M, 34 * M, 48 * M, 64 * M, 66 * M, 96 * M, 128 * M, 130 * M, 192 * M, 256 * M,
                               258 * M, 320 * M, 384 * M, 512 * M, 514 * M, 640 * M, 768 * M};
    const long long maxs = 800 * M;
    double *pool;
    CK(hipMalloc(&pool, (size_t)(4 * (n * 8 + maxs) + 4096)));
    CK(hipMemset(pool, 0, (size_t)(4 * (n * 8 + maxs) + 4096)));
    printf("n = %lld doubles (%.3f GiB per vector); bytes moved: update 5 x, rw 4 x\n", n, n * 8.0 / (1 << 30));
    for (int rep = 0; rep < 2; ++rep)
        for (long long sk : skews) {
            const long long S = n * 8 + sk;
            char *b0 = (char *)pool;
            double b1, b2;
            const double m1 = run(0, n, (double *)b0, (double *)(b0 + S), (double *)(b0 + 2 * S), nullptr, &b1);
            const double m2 = run(1, n, (double *)b0, (double *)(b0 + S), (double *)(b0 + 2 * S), (double *)(b0 + 3 * S), &b2);
            printf("one pool, stride n*8 + %8lld : update mean %.3f best %.3f ms (%.2f TB/s) | rw mean %.3f best %.3f ms (%.2f TB/s)\n", sk, m1, b1,
                   5.0 * n * 8 / b1 / 1e9, m2, b2, 4.0 * n * 8 / b2 / 1e9);
        }
    // separate allocations, several times (what hipMalloc hands out)
    for (int rep = 0; rep < 4; ++rep) {
        double *a[4];
        std::vector<void *> junk;
        for (int q = 0; q < rep; ++q) {  // perturb the allocator between repetitions
            void *j;
            CK(hipMalloc(&j, (size_t)((q + 1) * 3 << 20)));
            junk.push_back(j);
        }
        for (int q = 0; q < 4; ++q) {
            CK(hipMalloc(&a[q], (size_t)(n * 8)));
            CK(hipMemset(a[q], 0, (size_t)(n * 8)));
        }
        double b1, b2;
        const double m1 = run(0, n, a[0], a[1], a[2], nullptr, &b1);
        const double m2 = run(1, n, a[0], a[1], a[2], a[3], &b2);
        printf("separate allocations #%d (%p %p %p %p): update mean %.3f best %.3f ms | rw mean %.3f best %.3f ms\n", rep, (void *)a[0], (void *)a[1],
               (void *)a[2], (void *)a[3], m1, b1, m2, b2);
        for (int q = 0; q < 4; ++q) CK(hipFree(a[q]));
        for (void *j : junk) CK(hipFree(j));
    }
    CK(hipFree(pool));
    return 0;
}
