// gridsync_lab.hip -- what a grid-wide barrier costs on MI355X against a dependent kernel launch: the decision number for
// running the replicated coarse multigrid levels (~25 dependent phases of a few microseconds each) in ONE cooperative
// multi-workgroup kernel instead of one launch per phase.
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync_lab.hip -o tools/gridsync_lab && tools/gridsync_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

// sense-reversing barrier over `nb` co-resident workgroups: agent-scope release before arriving, acquire after leaving
// (on a multi-XCD part that is an L2 write-back and invalidate: data written by a workgroup on one XCD must reach one on another)
__device__ __forceinline__ void grid_sync(unsigned *bar, unsigned nb, unsigned &gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = gen + 1;
        if (atomicAdd(&bar[0], 1u) == nb - 1) {
            bar[0] = 0;
            __threadfence();
            __hip_atomic_store(&bar[1], g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != g) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    gen += 1;
    __syncthreads();
}

// `phases` phases of a small Jacobi-like update over n doubles with a grid barrier in between (ping-pong a / b)
__global__ __launch_bounds__(256) void k_phases(double *a, double *b, int n, int phases, unsigned *bar)
{
    unsigned gen = 0;
    const int gt = blockIdx.x * 256 + threadIdx.x, gs = gridDim.x * 256;
    for (int p = 0; p < phases; ++p) {
        for (int i = gt; i < n; i += gs) b[i] = 0.5 * (a[i > 0 ? i - 1 : i] + a[i < n - 1 ? i + 1 : i]);
        grid_sync(bar, gridDim.x, gen);
        double *t = a;
        a = b;
        b = t;
    }
}
__global__ __launch_bounds__(256) void k_one(const double *a, double *b, int n)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) b[i] = 0.5 * (a[i > 0 ? i - 1 : i] + a[i < n - 1 ? i + 1 : i]);
}

int main()
{
    const int phases = 200;
    unsigned *bar = nullptr;
    CHECK(hipMalloc(&bar, 2 * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipStream_t q;
    CHECK(hipStreamCreate(&q));
    std::printf("# %d dependent phases over n doubles: one launch per phase vs one kernel with grid barriers (us per phase)\n", phases);
    std::printf("%10s %8s %14s %14s\n", "n", "blocks", "launch/phase", "barrier/phase");
    for (int n : {4096, 32768, 262144, 2097152}) {
        double *a = nullptr, *b = nullptr;
        CHECK(hipMalloc(&a, sizeof(double) * n));
        CHECK(hipMalloc(&b, sizeof(double) * n));
        CHECK(hipMemset(a, 0, sizeof(double) * n));
        for (int nb : {8, 32, 64, 128, 256}) {
            if (nb * 256 > 4 * n) continue;
            float ms_l = 0, ms_b = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0, q));
                for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_one, dim3(nb), dim3(256), 0, q, (p & 1) ? b : a, (p & 1) ? a : b, n);
                CHECK(hipEventRecord(e1, q));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms_l, e0, e1));
                CHECK(hipMemsetAsync(bar, 0, 2 * sizeof(unsigned), q));
                CHECK(hipEventRecord(e0, q));
                hipLaunchKernelGGL(k_phases, dim3(nb), dim3(256), 0, q, a, b, n, phases, bar);
                CHECK(hipEventRecord(e1, q));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms_b, e0, e1));
            }
            std::printf("%10d %8d %11.2f us %11.2f us\n", n, nb, 1e3 * ms_l / phases, 1e3 * ms_b / phases);
        }
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }
    return 0;
}
