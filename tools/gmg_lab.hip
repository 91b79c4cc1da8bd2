// gmg_lab.hip -- A/B harness for the matrix-free level kernels of the multigrid (K2/K5) on gfx950.
// Variants of the damped-Jacobi sweep  xo = xi + omega (b - A xi)/diag  on an n^3 grid.
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gmg_lab.hip -o tools/gmg_lab
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct L {
    int nx, ny, nz;
    const double *wx, *wy, *wz, *gx, *gy, *gz;
};

// ---- V1: the product kernel as of round 1 (one cell per lane, 2-D grid)
__global__ __launch_bounds__(256) void k_v1(L l, double omega, const double *__restrict__ b, const double *__restrict__ xi,
                                            double *__restrict__ xo)
{
    const unsigned plane = (unsigned)l.nx * l.ny;
    const int k = blockIdx.y;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < plane; q += gridDim.x * 256u) {
        const int j = q / (unsigned)l.nx, i = q - j * l.nx;
        const int64_t p = (int64_t)k * plane + q;
        const double wxi = l.wx[i], wyj = l.wy[j], wzk = l.wz[k];
        const double ax = wyj * wzk, ay = wxi * wzk, az = wxi * wyj;
        const double c0 = (i > 0) ? ax * l.gx[i - 1] : 0.0, c1 = (i < l.nx - 1) ? ax * l.gx[i] : 0.0;
        const double c2 = (j > 0) ? ay * l.gy[j - 1] : 0.0, c3 = (j < l.ny - 1) ? ay * l.gy[j] : 0.0;
        const double c4 = (k > 0) ? az * l.gz[k - 1] : 0.0, c5 = (k < l.nz - 1) ? az * l.gz[k] : 0.0;
        const double xc = xi[p];
        double s = 0.0;
        if (i > 0) s += c0 * (xi[p - 1] - xc);
        if (i < l.nx - 1) s += c1 * (xi[p + 1] - xc);
        if (j > 0) s += c2 * (xi[p - l.nx] - xc);
        if (j < l.ny - 1) s += c3 * (xi[p + l.nx] - xc);
        if (k > 0) s += c4 * (xi[p - plane] - xc);
        if (k < l.nz - 1) s += c5 * (xi[p + plane] - xc);
        const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
        xo[p] = xc + omega * ((b[p] - s) / d);
    }
}

// ---- V2: C cells per lane along i (C = 2 or 4), wide loads, one (j,k) row segment per lane group
template <int C>
__global__ __launch_bounds__(256) void k_vc(L l, double omega, const double *__restrict__ b, const double *__restrict__ xi,
                                            double *__restrict__ xo)
{
    typedef double vt __attribute__((ext_vector_type(C)));
    const unsigned nxc = (unsigned)l.nx / C;  // lane groups per row
    const unsigned planec = nxc * l.ny;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int k = blockIdx.y;
    const double wzk = l.wz[k];
    const double gzm = (k > 0) ? l.gz[k - 1] : 0.0, gzp = (k < l.nz - 1) ? l.gz[k] : 0.0;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < planec; q += gridDim.x * 256u) {
        const int j = q / nxc, i0 = (q - j * nxc) * C;
        const int64_t p = (int64_t)k * plane + (int64_t)j * l.nx + i0;
        const vt xc = *reinterpret_cast<const vt *>(xi + p);
        const vt bv = *reinterpret_cast<const vt *>(b + p);
        const double xl = (i0 > 0) ? xi[p - 1] : 0.0;
        const double xr = (i0 + C < l.nx) ? xi[p + C] : 0.0;
        vt ym = xc, yp = xc, zm = xc, zp = xc;
        if (j > 0) ym = *reinterpret_cast<const vt *>(xi + p - l.nx);
        if (j < l.ny - 1) yp = *reinterpret_cast<const vt *>(xi + p + l.nx);
        if (k > 0) zm = *reinterpret_cast<const vt *>(xi + p - plane);
        if (k < l.nz - 1) zp = *reinterpret_cast<const vt *>(xi + p + plane);
        const double wyj = l.wy[j];
        const double gym = (j > 0) ? l.gy[j - 1] : 0.0, gyp = (j < l.ny - 1) ? l.gy[j] : 0.0;
        const double ax = wyj * wzk;
        vt out;
        double gxm = (i0 > 0) ? l.gx[i0 - 1] : 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int i = i0 + c;
            const double wxi = l.wx[i];
            const double gxp = (i < l.nx - 1) ? l.gx[i] : 0.0;
            const double ay = wxi * wzk, az = wxi * wyj;
            const double c0 = ax * gxm, c1 = ax * gxp, c2 = ay * gym, c3 = ay * gyp, c4 = az * gzm, c5 = az * gzp;
            const double left = (c == 0) ? xl : xc[c - 1];
            const double right = (c == C - 1) ? xr : xc[c + 1];
            double s = 0.0;
            if (i > 0) s += c0 * (left - xc[c]);
            if (i < l.nx - 1) s += c1 * (right - xc[c]);
            if (j > 0) s += c2 * (ym[c] - xc[c]);
            if (j < l.ny - 1) s += c3 * (yp[c] - xc[c]);
            if (k > 0) s += c4 * (zm[c] - xc[c]);
            if (k < l.nz - 1) s += c5 * (zp[c] - xc[c]);
            const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
            out[c] = xc[c] + omega * ((bv[c] - s) / d);
            gxm = gxp;
        }
        *reinterpret_cast<vt *>(xo + p) = out;
    }
}

// ---- V3: C cells per lane, one plane-row per workgroup row: blockIdx.y = k, blockIdx.z... 3-D grid:
//      x = lane groups along i, y = j, z = k: no division at all
template <int C>
__global__ __launch_bounds__(256) void k_v3d(L l, double omega, const double *__restrict__ b, const double *__restrict__ xi,
                                             double *__restrict__ xo)
{
    typedef double vt __attribute__((ext_vector_type(C)));
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int k = blockIdx.z;
    const double wzk = l.wz[k];
    const double gzm = (k > 0) ? l.gz[k - 1] : 0.0, gzp = (k < l.nz - 1) ? l.gz[k] : 0.0;
    // a 256-thread workgroup covers ROWS_PER_WG = 256*C/nx rows when nx <= 256*C
    const int lanes_per_row = l.nx / C;
    const int rows_per_wg = 256 / lanes_per_row > 0 ? 256 / lanes_per_row : 1;
    const int jr = threadIdx.x / lanes_per_row;
    const int j = blockIdx.y * rows_per_wg + jr;
    const int i0 = (threadIdx.x - jr * lanes_per_row + blockIdx.x * 256) * C;
    if (j >= l.ny || i0 >= l.nx || jr >= rows_per_wg) return;
    const int64_t p = (int64_t)k * plane + (int64_t)j * l.nx + i0;
    const vt xc = *reinterpret_cast<const vt *>(xi + p);
    const vt bv = *reinterpret_cast<const vt *>(b + p);
    const double xl = (i0 > 0) ? xi[p - 1] : 0.0;
    const double xr = (i0 + C < l.nx) ? xi[p + C] : 0.0;
    vt ym = xc, yp = xc, zm = xc, zp = xc;
    if (j > 0) ym = *reinterpret_cast<const vt *>(xi + p - l.nx);
    if (j < l.ny - 1) yp = *reinterpret_cast<const vt *>(xi + p + l.nx);
    if (k > 0) zm = *reinterpret_cast<const vt *>(xi + p - plane);
    if (k < l.nz - 1) zp = *reinterpret_cast<const vt *>(xi + p + plane);
    const double wyj = l.wy[j];
    const double gym = (j > 0) ? l.gy[j - 1] : 0.0, gyp = (j < l.ny - 1) ? l.gy[j] : 0.0;
    const double ax = wyj * wzk;
    vt out;
    double gxm = (i0 > 0) ? l.gx[i0 - 1] : 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = i0 + c;
        const double wxi = l.wx[i];
        const double gxp = (i < l.nx - 1) ? l.gx[i] : 0.0;
        const double ay = wxi * wzk, az = wxi * wyj;
        const double c0 = ax * gxm, c1 = ax * gxp, c2 = ay * gym, c3 = ay * gyp, c4 = az * gzm, c5 = az * gzp;
        const double left = (c == 0) ? xl : xc[c - 1];
        const double right = (c == C - 1) ? xr : xc[c + 1];
        double s = 0.0;
        if (i > 0) s += c0 * (left - xc[c]);
        if (i < l.nx - 1) s += c1 * (right - xc[c]);
        if (j > 0) s += c2 * (ym[c] - xc[c]);
        if (j < l.ny - 1) s += c3 * (yp[c] - xc[c]);
        if (k > 0) s += c4 * (zm[c] - xc[c]);
        if (k < l.nz - 1) s += c5 * (zp[c] - xc[c]);
        const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
        out[c] = xc[c] + omega * ((bv[c] - s) / d);
        gxm = gxp;
    }
    *reinterpret_cast<vt *>(xo + p) = out;
}

// ---- reference streams: read 2 vectors write 1 (the 24 B/row floor)
__global__ __launch_bounds__(256) void k_floor(int64_t n, const double *__restrict__ b, const double *__restrict__ xi,
                                               double *__restrict__ xo)
{
    typedef double vt __attribute__((ext_vector_type(2)));
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (2 * i + 1 < n) {
        const vt a = *reinterpret_cast<const vt *>(xi + 2 * i), c = *reinterpret_cast<const vt *>(b + 2 * i);
        *reinterpret_cast<vt *>(xo + 2 * i) = a + 0.9 * c;
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int64_t N = (int64_t)n * n * n;
    std::vector<double> hw(n, 1.0 / n), hg(n - 1, 5e-4 * n);
    double *w, *g, *b, *x, *y, *yref;
    CK(hipMalloc(&w, 8 * n));
    CK(hipMalloc(&g, 8 * n));
    CK(hipMemcpy(w, hw.data(), 8 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), 8 * (n - 1), hipMemcpyHostToDevice));
    CK(hipMalloc(&b, 8 * N));
    CK(hipMalloc(&x, 8 * N));
    CK(hipMalloc(&y, 8 * N));
    CK(hipMalloc(&yref, 8 * N));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int64_t o = 0; o < N; o += (int64_t)h.size()) {
        CK(hipMemcpy(b + o, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(x + o, h.data() + 7, 8 * (h.size() - 7), hipMemcpyHostToDevice));
    }
    L l{n, n, n, w, w, w, g, g, g};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> h0(4096), h1(4096);
    auto bench = [&](const char *name, auto launch, bool check) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        int bad = -1;
        if (check) {
            bad = 0;
            for (int64_t off : {int64_t(0), N / 3, N / 2 + 12345, N - 4096}) {
                CK(hipMemcpy(h0.data(), yref + off, 8 * 4096, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h1.data(), y + off, 8 * 4096, hipMemcpyDeviceToHost));
                for (int i = 0; i < 4096; ++i) bad += (h0[i] != h1[i]);
            }
        }
        printf("%-46s %8.3f ms  %7.1f GB/s (24 B/cell)  mismatches=%d\n", name, ms, 24.0 * N / ms / 1e6, bad);
        fflush(stdout);
    };
    const unsigned plane = (unsigned)n * n;
    hipLaunchKernelGGL(k_v1, dim3((plane + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x, yref);
    CK(hipDeviceSynchronize());
    bench("floor: 2 reads + 1 write, double2", [&] {
        hipLaunchKernelGGL(k_floor, dim3((unsigned)((N / 2 + 255) / 256)), dim3(256), 0, 0, N, b, x, y); }, false);
    bench("V1 one cell/lane 2-D grid", [&] {
        hipLaunchKernelGGL(k_v1, dim3((plane + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x, y); }, true);
    bench("V2 two cells/lane 2-D grid", [&] {
        hipLaunchKernelGGL(k_vc<2>, dim3((plane / 2 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x, y); }, true);
    bench("V2 four cells/lane 2-D grid", [&] {
        hipLaunchKernelGGL(k_vc<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x, y); }, true);
    {
        const int lpr2 = n / 2, rows2 = 256 / lpr2 > 0 ? 256 / lpr2 : 1;
        bench("V3 two cells/lane 3-D grid (no division)", [&] {
            hipLaunchKernelGGL(k_v3d<2>, dim3((lpr2 + 255) / 256, (n + rows2 - 1) / rows2, n), dim3(256), 0, 0, l, 0.9, b, x, y); }, true);
        const int lpr4 = n / 4, rows4 = 256 / lpr4 > 0 ? 256 / lpr4 : 1;
        bench("V3 four cells/lane 3-D grid (no division)", [&] {
            hipLaunchKernelGGL(k_v3d<4>, dim3((lpr4 + 255) / 256, (n + rows4 - 1) / rows4, n), dim3(256), 0, 0, l, 0.9, b, x, y); }, true);
    }
    return 0;
}
