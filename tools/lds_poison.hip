// tools/lds_poison.hip -- flake hunt (round 4): fill the LDS of every CU with a pattern (NaN / 1e30) so that a kernel reading LDS
// it never wrote shows up as a wrong result instead of depending on what ran before it.  hipcc --offload-arch=gfx950 -O2 -shared
// -fPIC tools/lds_poison.hip -o tools/liblds_poison.so ; used by tools/fuzz_debug.py (PIB_FUZZ_LDS=nan|big)
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void k_poison(double v)
{
    extern __shared__ double sh[];
    for (int i = threadIdx.x; i < 20000; i += 1024) sh[i] = v;
    __syncthreads();
    if (sh[(threadIdx.x * 7) % 20000] == 12345.678) printf("x");  // keep the stores
}
extern "C" int lds_poison(double v)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipLaunchKernelGGL(k_poison, dim3(2048), dim3(1024), 160000, nullptr, v);
    return (int)hipDeviceSynchronize();
}
