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
//   matrix     : 4 + 4 + 4 separate allocations, the update timed over all 64 (z, p, x) triples
//   perbuf     : 16 separate allocations timed one at a time (x *= a, read), then the update over the best / worst three
//   pairs      : a two-vector read-write stream over every ordered pair of NC (24) separate allocations
//   mix        : classes of allocations by the pair stream, then 7r1w / 2r2w / 3r1w / update streams inside one class and spread
//   far        : a candidate every 4 GiB up to 224 GiB, the pair stream against the first buffer and against the one before
//   vmm, vmm1g : the same through hipMemAddressReserve / hipMemCreate / hipMemMap, addresses reserved on 2 MiB / 1 GiB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
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
__global__ __launch_bounds__(256) void k_scale(long long ng, v2 *__restrict__ x, double a)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) x[i] = a * x[i];
    }
}
__global__ __launch_bounds__(256) void k_pair(long long ng, v2 *__restrict__ p, v2 *__restrict__ x, double a, double b)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) {
            const v2 vp = p[i];
            v2 vx = x[i];
            vx += a * vp;
            x[i] = vx;
            p[i] = b * vp;
        }
    }
}
// y = sum of seven read streams (the shape of the CSR product's traffic: many read streams, one written)
__global__ __launch_bounds__(256) void k_read7(long long ng, const v2 *__restrict__ a0, const v2 *__restrict__ a1, const v2 *__restrict__ a2, const v2 *__restrict__ a3,
                                               const v2 *__restrict__ a4, const v2 *__restrict__ a5, const v2 *__restrict__ a6, v2 *__restrict__ y)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) y[i] = a0[i] + a1[i] + a2[i] + a3[i] + a4[i] + a5[i] + a6[i];
    }
}
// two read, two written (the fused residual update's march: r_old, w read; r_new, c written)
__global__ __launch_bounds__(256) void k_r2w2(long long ng, const v2 *__restrict__ a0, const v2 *__restrict__ a1, v2 *__restrict__ y0, v2 *__restrict__ y1)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) {
            const v2 r = a0[i] - 0.5 * a1[i];
            y0[i] = r;
            y1[i] = 0.9 * r;
        }
    }
}
// three read, one written (z = z + e + r-ish: the prolongation's fine-level traffic)
__global__ __launch_bounds__(256) void k_r3w1(long long ng, const v2 *__restrict__ a0, const v2 *__restrict__ a1, const v2 *__restrict__ a2, v2 *__restrict__ y0)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + 256 * u;
        if (i < ng) y0[i] = a0[i] + 0.5 * a1[i] + 0.25 * a2[i];
    }
}
static double best_of(int reps, const std::function<void()> &launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 1e9;
    for (int r = 0; r <= reps; ++r) {
        float ms;
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, (double)ms);
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}
// one buffer through the virtual-memory API: the address reserved on `align`, one physical handle of the whole size
static double *alloc_vmm(size_t bytes, size_t align)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t sz = (bytes + gran - 1) / gran * gran;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, sz, align, nullptr, 0));
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, sz, &prop, 0));
    CK(hipMemMap(va, sz, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, sz, &acc, 1));
    CK(hipMemset(va, 0, sz));
    return (double *)va;
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
    } else if (!strcmp(mode, "matrix")) {
        // every (z, p, x) out of 4 + 4 + 4 separately allocated candidates: how many triples are in the fast mode?
        double *c[12];
        for (int q = 0; q < 12; ++q) c[q] = alloc(B);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const long long ng = n / 2;
        const unsigned nb = (unsigned)((ng + 1023) / 1024);
        int fast = 0, all = 0;
        for (int a = 0; a < 4; ++a)
            for (int b2 = 0; b2 < 4; ++b2) {
                printf("z%d p%d:", a, b2);
                for (int x2 = 0; x2 < 4; ++x2) {
                    double best = 1e30;
                    for (int r = 0; r < 4; ++r) {
                        float ms;
                        CK(hipEventRecord(e0, 0));
                        hipLaunchKernelGGL(k_update, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)c[a], (v2 *)c[4 + b2], (v2 *)c[8 + x2], 1e-3, 0.5);
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r) best = std::min(best, (double)ms);
                    }
                    printf(" %.3f", best);
                    ++all;
                    if (best < 0.91) ++fast;
                }
                printf("\n");
            }
        printf("fast triples (< 0.91 ms): %d of %d\n", fast, all);
        for (int q = 0; q < 12; ++q) printf("  c[%d] = %p\n", q, (void *)c[q]);
    } else if (!strcmp(mode, "perbuf") || !strcmp(mode, "vmm") || !strcmp(mode, "vmm1g")) {
        // is the mode a property of single buffers?  16 candidates, each timed alone (x *= a, read), then the
        // update over the three best and the three worst by the single-buffer time
        const int NC = 16;
        double *c[NC];
        double ts[NC], tr[NC];
        const bool vmm = mode[0] == 'v';
        for (int q = 0; q < NC; ++q) c[q] = vmm ? alloc_vmm(B, !strcmp(mode, "vmm1g") ? (size_t)1 << 30 : (size_t)2 << 20) : alloc(B);
        const long long ng = n / 2;
        const unsigned nb = (unsigned)((ng + 1023) / 1024);
        double *out = alloc(4096);
        for (int pass = 0; pass < 2; ++pass)
            for (int q = 0; q < NC; ++q) {
                ts[q] = best_of(6, [&] { hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, 0, ng, (v2 *)c[q], 1.0); });
                tr[q] = best_of(6, [&] { hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)c[q], out); });
                printf("pass %d c[%2d] = %p  scale %.3f ms = %.2f TB/s   read %.3f ms = %.2f TB/s\n", pass, q, (void *)c[q], ts[q], 2.0 * B / ts[q] / 1e9, tr[q],
                       1.0 * B / tr[q] / 1e9);
            }
        int ord[NC];
        for (int q = 0; q < NC; ++q) ord[q] = q;
        std::sort(ord, ord + NC, [&](int a, int b) { return ts[a] < ts[b]; });
        auto upd = [&](int a, int b, int x) {
            return best_of(6, [&] { hipLaunchKernelGGL(k_update, dim3(nb), dim3(256), 0, 0, ng, (const v2 *)c[a], (v2 *)c[b], (v2 *)c[x], 1e-3, 0.5); });
        };
        printf("update over the three best  (%d %d %d): %.3f ms\n", ord[0], ord[1], ord[2], upd(ord[0], ord[1], ord[2]));
        printf("update over the next three  (%d %d %d): %.3f ms\n", ord[3], ord[4], ord[5], upd(ord[3], ord[4], ord[5]));
        printf("update over the three worst (%d %d %d): %.3f ms\n", ord[NC - 3], ord[NC - 2], ord[NC - 1], upd(ord[NC - 3], ord[NC - 2], ord[NC - 1]));
        printf("update, x worst, z p best   (%d %d %d): %.3f ms\n", ord[0], ord[1], ord[NC - 1], upd(ord[0], ord[1], ord[NC - 1]));
        printf("update, z worst, p x best   (%d %d %d): %.3f ms\n", ord[NC - 1], ord[0], ord[1], upd(ord[NC - 1], ord[0], ord[1]));
        printf("update, p worst, z x best   (%d %d %d): %.3f ms\n", ord[0], ord[NC - 1], ord[1], upd(ord[0], ord[NC - 1], ord[1]));
        // the whole x column for z p best: which x candidates put the update in the fast mode
        printf("x column (z %d, p %d):", ord[0], ord[1]);
        for (int q = 0; q < NC; ++q)
            if (q != ord[0] && q != ord[1]) printf(" %d:%.3f", q, upd(ord[0], ord[1], q));
        printf("\n");
    } else if (!strcmp(mode, "pairs")) {
        // the two-vector read-write stream over every ordered pair out of NC separate allocations: is "slow" an equivalence
        // (classes of allocations) or a matter of distance?
        const int NC = argc > 3 ? atoi(argv[3]) : 24;
        std::vector<double *> c(NC);
        for (int q = 0; q < NC; ++q) c[q] = alloc(B);
        const long long ng = n / 2;
        const unsigned nb = (unsigned)((ng + 1023) / 1024);
        std::vector<double> t((size_t)NC * NC, 0.0);
        double lo = 1e9, hi = 0;
        for (int a = 0; a < NC; ++a)
            for (int b = 0; b < NC; ++b)
                if (a != b) {
                    const double v = best_of(3, [&] { hipLaunchKernelGGL(k_pair, dim3(nb), dim3(256), 0, 0, ng, (v2 *)c[a], (v2 *)c[b], 1e-3, 0.5); });
                    t[(size_t)a * NC + b] = v;
                    lo = std::min(lo, v), hi = std::max(hi, v);
                }
        printf("pair stream (p rows, x columns), best %.3f ms worst %.3f ms; F: within 4 %% of the best, s: slower, m: between\n", lo, hi);
        for (int a = 0; a < NC; ++a) {
            printf("p %2d %p ", a, (void *)c[a]);
            for (int b = 0; b < NC; ++b) {
                const double v = t[(size_t)a * NC + b];
                putchar(a == b ? '.' : v <= 1.04 * lo ? 'F' : v >= 0.96 * hi ? 's' : 'm');
            }
            printf("   ");
            for (int b = 0; b < NC; ++b) printf(" %.3f", t[(size_t)a * NC + b]);
            printf("\n");
        }
    } else if (!strcmp(mode, "mix")) {
        // classes of allocations by the pair stream, then other stream mixes with every vector in ONE class against spread over classes
        const int NC = argc > 3 ? atoi(argv[3]) : 32;
        std::vector<double *> c(NC);
        for (int q = 0; q < NC; ++q) c[q] = alloc(B);
        const long long ng = n / 2;
        const unsigned nb = (unsigned)((ng + 1023) / 1024);
        auto pair = [&](int a, int b) { return best_of(3, [&] { hipLaunchKernelGGL(k_pair, dim3(nb), dim3(256), 0, 0, ng, (v2 *)c[a], (v2 *)c[b], 1e-3, 0.5); }); };
        // fast / slow threshold from buffer 0 against all the others
        std::vector<double> t0(NC, 0.0);
        double lo = 1e9, hi = 0;
        for (int q = 1; q < NC; ++q) t0[q] = pair(0, q), lo = std::min(lo, t0[q]), hi = std::max(hi, t0[q]);
        const double mid = 0.5 * (lo + hi);
        printf("pair stream against buffer 0: best %.3f worst %.3f ms\n", lo, hi);
        std::vector<int> cls(NC, -1), rep;
        for (int q = 0; q < NC; ++q) {
            for (size_t k = 0; k < rep.size() && cls[q] < 0; ++k)
                if (rep[k] != q && pair(rep[k], q) > mid) cls[q] = (int)k;
            if (cls[q] < 0) cls[q] = (int)rep.size(), rep.push_back(q);
        }
        printf("classes:");
        for (int q = 0; q < NC; ++q) printf(" %d", cls[q]);
        printf("\n");
        std::vector<std::vector<int>> mem(rep.size());
        for (int q = 0; q < NC; ++q) mem[cls[q]].push_back(q);
        int big = 0;
        for (size_t k = 0; k < mem.size(); ++k)
            if (mem[k].size() > mem[big].size()) big = (int)k;
        if (mem[big].size() < 8 || mem.size() < 2) {
            printf("no class of 8\n");
            return 0;
        }
        std::vector<int> same(mem[big].begin(), mem[big].begin() + 8), spread;
        for (size_t r = 0; spread.size() < 8; ++r)
            for (size_t k = 0; k < mem.size() && spread.size() < 8; ++k)
                if (r < mem[k].size()) spread.push_back(mem[k][r]);
        auto run = [&](const char *tag, const std::vector<int> &v) {
            auto P = [&](int k) { return (v2 *)c[v[k]]; };
            const double a = best_of(5, [&] { hipLaunchKernelGGL(k_read7, dim3(nb), dim3(256), 0, 0, ng, P(0), P(1), P(2), P(3), P(4), P(5), P(6), P(7)); });
            const double b = best_of(5, [&] { hipLaunchKernelGGL(k_r2w2, dim3(nb), dim3(256), 0, 0, ng, P(0), P(1), P(2), P(3)); });
            const double d = best_of(5, [&] { hipLaunchKernelGGL(k_r3w1, dim3(nb), dim3(256), 0, 0, ng, P(0), P(1), P(2), P(3)); });
            const double e = best_of(5, [&] { hipLaunchKernelGGL(k_update, dim3(nb), dim3(256), 0, 0, ng, P(0), P(1), P(2), 1e-3, 0.5); });
            printf("%-28s (", tag);
            for (int k = 0; k < 8; ++k) printf(" %d", v[k]);
            printf(" ): 7r1w %.3f ms = %.2f TB/s | 2r2w %.3f ms = %.2f TB/s | 3r1w %.3f ms = %.2f TB/s | update %.3f ms = %.2f TB/s\n", a, 8.0 * B / a / 1e9, b,
                   4.0 * B / b / 1e9, d, 4.0 * B / d / 1e9, e, 5.0 * B / e / 1e9);
        };
        for (int r = 0; r < 2; ++r) {
            run("one class", same);
            run("spread over the classes", spread);
        }
        // written vectors in one class, read ones in another; and the other way round
        if (mem.size() >= 2) {
            int other = big == 0 ? 1 : 0;
            std::vector<int> v = {mem[big][0], mem[big][1], mem[other][0], mem[other % mem.size()].size() > 1 ? mem[other][1] : mem[other][0], 0, 0, 0, 0};
            auto P = [&](int k) { return (v2 *)c[v[k]]; };
            const double b1 = best_of(5, [&] { hipLaunchKernelGGL(k_r2w2, dim3(nb), dim3(256), 0, 0, ng, P(0), P(1), P(2), P(3)); });
            const double b2 = best_of(5, [&] { hipLaunchKernelGGL(k_r2w2, dim3(nb), dim3(256), 0, 0, ng, P(0), P(2), P(1), P(3)); });
            printf("2r2w: reads one class, writes another %.3f ms | each read with a write of its class, the two pairs apart %.3f ms\n", b1, b2);
        }
    } else if (!strcmp(mode, "far")) {
        // the class structure at the scale of the whole device: buffer 0, then a candidate every STEP GiB (spacers untouched)
        // up to 224 GiB; the pair stream of each candidate against buffer 0 and against the candidate before it
        const int stepg = argc > 3 ? atoi(argv[3]) : 4;
        double *b0 = alloc(B), *prev = b0;
        const long long ng = n / 2;
        const unsigned nb = (unsigned)((ng + 1023) / 1024);
        size_t total = B;
        for (int k = 0; total < ((size_t)224 << 30); ++k) {
            void *sp = nullptr;
            if (stepg > 1 && hipMalloc(&sp, ((size_t)stepg << 30) - B) != hipSuccess) break;
            double *c;
            if (hipMalloc(&c, B) != hipSuccess) break;
            CK(hipMemset(c, 0, B));
            total += (size_t)stepg << 30;
            const double t0 = best_of(3, [&] { hipLaunchKernelGGL(k_pair, dim3(nb), dim3(256), 0, 0, ng, (v2 *)b0, (v2 *)c, 1e-3, 0.5); });
            const double t1 = best_of(3, [&] { hipLaunchKernelGGL(k_pair, dim3(nb), dim3(256), 0, 0, ng, (v2 *)prev, (v2 *)c, 1e-3, 0.5); });
            printf("at %3zu GiB %p: against buffer 0 %.3f ms, against the one before %.3f ms\n", total >> 30, (void *)c, t0, t1);
            prev = c;
        }
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
