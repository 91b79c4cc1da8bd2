// tools/place_lab.hip -- the rate of a flat 3-read-2-write stream against HOW its vectors were allocated (round 5).
// One process per mode (no allocator history):  ./place_lab MODE [n]
//   fresh      : 4 x hipMalloc(n * 8) first thing
//   padded     : 4 x hipMalloc(n * 8 + 2 MiB + 64)     (work vectors are not powers of two)
//   contiguous : 4 x hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   churn      : 40 allocations of 3..300 MiB, every other one freed, then 4 x hipMalloc(n * 8)
//   recycled   : 4 x hipMalloc, free them, 3 MiB junk, 4 x hipMalloc again (what a second solver in the process gets)
//   pool       : one hipMalloc of 4 n * 8 + 16 MiB, vectors at n * 8 + 4 KiB strides
//   pool2      : one hipMalloc of exactly 4 GiB, vectors at 1 GiB strides
//   shop       : 6 candidate sets of 3 vectors, each timed; prints the spread (is the rate a property of the allocation?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
__global__ __launch_bounds__(256) void k_read(long long ng, const v2 *__restrict__ z, double *out)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
    v2 s = {0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) s += z[i];
    }
    if (s[0] + s[1] == 12345.678) out[0] = s[0];
}
static void timeit(const char *tag, long long n, double *z, double *p, double *x)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long ng = n / 2;
    const unsigned nb = (unsigned)((ng + 1023) / 1024);
    double best = 1e30, sum = 0, bestr = 1e30;
    for (int r = 0; r < 6; ++r) {
        float ms;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_update, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)z, (v2 *)p, (v2 *)x, 1e-3, 0.5);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) sum += ms, best = std::min(best, (double)ms);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)z, x);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) bestr = std::min(bestr, (double)ms);
    }
    printf("%-40s update mean %.3f best %.3f ms = %.2f TB/s | read best %.3f ms = %.2f TB/s   (%p %p %p)\n", tag, sum / 5, best, 5.0 * n * 8 / best / 1e9, bestr,
           n * 8.0 / bestr / 1e9, (void *)z, (void *)p, (void *)x);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}
static double *alloc(size_t bytes, bool contiguous = false)
{
    double *p;
    if (contiguous) CK(hipExtMallocWithFlags((void **)&p, bytes, hipDeviceMallocContiguous));
    else CK(hipMalloc(&p, bytes));
    CK(hipMemset(p, 0, bytes));
    return p;
}
int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "fresh";
    const long long n = argc > 2 ? atoll(argv[2]) : 134217728LL;
    const size_t B = (size_t)n * 8;
    double *v[4];
    if (!strcmp(mode, "fresh")) {
        for (int q = 0; q < 4; ++q) v[q] = alloc(B);
        timeit("fresh hipMalloc", n, v[0], v[1], v[2]);
        timeit("fresh hipMalloc (other three)", n, v[1], v[2], v[3]);
    } else if (!strcmp(mode, "padded")) {
        for (int q = 0; q < 4; ++q) v[q] = alloc(B + (2 << 20) + 64);
        timeit("hipMalloc(n*8 + 2 MiB + 64)", n, v[0], v[1], v[2]);
        timeit("... vectors 64 B into their blocks", n, v[0] + 8, v[1] + 8, v[2] + 8);
    } else if (!strcmp(mode, "contiguous")) {
        for (int q = 0; q < 4; ++q) v[q] = alloc(B, true);
        timeit("hipDeviceMallocContiguous", n, v[0], v[1], v[2]);
    } else if (!strcmp(mode, "churn")) {
        std::vector<void *> keep;
        for (int q = 0; q < 40; ++q) {
            void *j;
            CK(hipMalloc(&j, (size_t)(3 + 7 * q) << 20));
            if (q & 1) keep.push_back(j);
            else CK(hipFree(j));
        }
        for (int q = 0; q < 4; ++q) v[q] = alloc(B);
        timeit("after churn", n, v[0], v[1], v[2]);
    } else if (!strcmp(mode, "recycled")) {
        for (int q = 0; q < 4; ++q) v[q] = alloc(B);
        timeit("first generation", n, v[0], v[1], v[2]);
        for (int q = 0; q < 4; ++q) CK(hipFree(v[q]));
        void *j;
        CK(hipMalloc(&j, 3 << 20));
        for (int q = 0; q < 4; ++q) v[q] = alloc(B);
        timeit("second generation", n, v[0], v[1], v[2]);
        for (int q = 0; q < 4; ++q) CK(hipFree(v[q]));
        for (int q = 3; q >= 0; --q) v[q] = alloc(B);
        timeit("third generation", n, v[0], v[1], v[2]);
    } else if (!strcmp(mode, "pool")) {
        char *b0 = (char *)alloc(4 * B + (16 << 20));
        const size_t S = B + 4096;
        timeit("one pool, n*8 + 4 KiB strides", n, (double *)b0, (double *)(b0 + S), (double *)(b0 + 2 * S));
    } else if (!strcmp(mode, "pool2")) {
        char *b0 = (char *)alloc(4 * B);
        timeit("one pool of 4 n*8, n*8 strides", n, (double *)b0, (double *)(b0 + B), (double *)(b0 + 2 * B));
        timeit("... vectors 1, 2, 3", n, (double *)(b0 + B), (double *)(b0 + 2 * B), (double *)(b0 + 3 * B));
    } else if (!strcmp(mode, "shop")) {
        double *c[6][3];
        for (int s = 0; s < 6; ++s)
            for (int q = 0; q < 3; ++q) c[s][q] = alloc(B + ((size_t)s << 21));
        for (int s = 0; s < 6; ++s) {
            char tag[64];
            snprintf(tag, sizeof tag, "candidate set %d", s);
            timeit(tag, n, c[s][0], c[s][1], c[s][2]);
        }
        timeit("mixed: 0.z 3.p 5.x", n, c[0][0], c[3][1], c[5][2]);
    }
    return 0;
}
