// stream_lab.hip -- what read / copy bandwidth does MI355X give to plain HIP
// streaming kernels, as a function of load width, unroll, grid size, access
// order and cache policy?  Calibrates the ceiling for the HBM-bound kernels
// of this backend (SpMV, vector updates).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_lab.hip -o tools/stream_lab
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef double double2_ __attribute__((ext_vector_type(2)));
typedef double double4_ __attribute__((ext_vector_type(4)));

// grid-stride read, VEC doubles per lane per load, UNROLL independent loads in flight
template <int VEC, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read_gs(const double *__restrict__ a, int64_t n, double *__restrict__ out)
{
    typedef double vt __attribute__((ext_vector_type(VEC)));
    const vt *p = reinterpret_cast<const vt *>(a);
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * 256;
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nv; i += UNROLL * stride) {
        vt v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc += v[u][k];
    }
    for (; i < nv; i += stride) {
        vt v = p[i];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc += v[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

// block-contiguous read: block b owns a contiguous span, walks it in 256*VEC*UNROLL pieces
template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void k_read_bc(const double *__restrict__ a, int64_t n, double *__restrict__ out)
{
    typedef double vt __attribute__((ext_vector_type(VEC)));
    const vt *p = reinterpret_cast<const vt *>(a);
    const int64_t nv = n / VEC;
    const int64_t per = (nv + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per, b1 = (b0 + per < nv) ? b0 + per : nv;
    double acc = 0.0;
    int64_t i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < b1; i += UNROLL * 256) {
        vt v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc += v[u][k];
    }
    for (; i < b1; i += 256) {
        vt v = p[i];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc += v[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void k_copy_gs(const double *__restrict__ a, double *__restrict__ b, int64_t n)
{
    typedef double vt __attribute__((ext_vector_type(VEC)));
    const vt *p = reinterpret_cast<const vt *>(a);
    vt *q = reinterpret_cast<vt *>(b);
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nv; i += UNROLL * stride) {
        vt v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) q[i + u * stride] = v[u];
    }
    for (; i < nv; i += stride) q[i] = p[i];
}

// three-stream read like the CSR floor: val (8 B/entry) + col (4 B/entry), one entry group per lane
template <int EPL /* entries per lane: 2 or 4 */>
__global__ __launch_bounds__(256) void k_read_csr(const double *__restrict__ val, const int *__restrict__ col, int64_t nnz,
                                                  double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    double acc = 0.0;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; (g + 1) * EPL <= nnz; g += stride) {
        if (EPL == 2) {
            const double2_ v = *reinterpret_cast<const double2_ *>(val + 2 * g);
            const int2 c = *reinterpret_cast<const int2 *>(col + 2 * g);
            acc += v[0] * (double)c.x + v[1] * (double)c.y;
        } else {
            const double2_ v0 = *reinterpret_cast<const double2_ *>(val + 4 * g);
            const double2_ v1 = *reinterpret_cast<const double2_ *>(val + 4 * g + 2);
            const int4 c = *reinterpret_cast<const int4 *>(col + 4 * g);
            acc += v0[0] * (double)c.x + v0[1] * (double)c.y + v1[0] * (double)c.z + v1[1] * (double)c.w;
        }
    }
    if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int64_t n = (int64_t)(gb * 1e9 / 8.0) / 4096 * 4096;
    double *a, *b, *out;
    int *c;
    CK(hipMalloc(&a, 8 * n));
    CK(hipMalloc(&b, 8 * n));
    CK(hipMalloc(&c, 4 * n));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, 8 * n));
    CK(hipMemset(b, 0, 8 * n));
    CK(hipMemset(c, 0, 4 * n));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto bench = [&](const char *name, double bytes, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("%-52s %8.3f ms %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
        fflush(stdout);
    };
    const double rb = 8.0 * n;
    printf("array %.2f GB\n", rb / 1e9);
#define RD(V, U, NT, G) bench("read gs vec" #V " unroll" #U " nt" #NT " grid" #G, rb, [&] { \
        hipLaunchKernelGGL((k_read_gs<V, U, NT>), dim3(G), dim3(256), 0, 0, a, n, out); })
    RD(1, 1, false, 2048); RD(1, 4, false, 2048); RD(2, 1, false, 2048); RD(2, 2, false, 2048); RD(2, 4, false, 2048);
    RD(2, 8, false, 2048); RD(4, 2, false, 2048); RD(4, 4, false, 2048);
    RD(2, 4, true, 2048); RD(2, 4, false, 256); RD(2, 4, false, 512); RD(2, 4, false, 1024); RD(2, 4, false, 4096);
    RD(2, 4, false, 8192); RD(2, 4, false, 16384); RD(2, 1, false, 65536); RD(2, 1, false, 262144);
#define RB(V, U, G) bench("read block-contig vec" #V " unroll" #U " grid" #G, rb, [&] { \
        hipLaunchKernelGGL((k_read_bc<V, U>), dim3(G), dim3(256), 0, 0, a, n, out); })
    RB(2, 4, 2048); RB(2, 4, 8192); RB(2, 8, 2048); RB(2, 4, 65536);
#define CP(V, U, G) bench("copy gs vec" #V " unroll" #U " grid" #G " (r+w bytes)", 2 * rb, [&] { \
        hipLaunchKernelGGL((k_copy_gs<V, U>), dim3(G), dim3(256), 0, 0, a, b, n); })
    CP(2, 1, 2048); CP(2, 4, 2048); CP(2, 4, 8192); CP(4, 2, 2048);
    bench("read csr-like val+col 2/lane grid2048", 12.0 * n, [&] {
        hipLaunchKernelGGL((k_read_csr<2>), dim3(2048), dim3(256), 0, 0, a, c, n, out); });
    bench("read csr-like val+col 4/lane grid2048", 12.0 * n, [&] {
        hipLaunchKernelGGL((k_read_csr<4>), dim3(2048), dim3(256), 0, 0, a, c, n, out); });
    bench("read csr-like val+col 4/lane grid8192", 12.0 * n, [&] {
        hipLaunchKernelGGL((k_read_csr<4>), dim3(8192), dim3(256), 0, 0, a, c, n, out); });
    bench("hipMemcpyDtoD (r+w bytes)", 2 * rb, [&] { CK(hipMemcpyAsync(b, a, 8 * n, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
