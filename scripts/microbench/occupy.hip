// Diagnostic (not part of the product): a kernel that just SITS on some CUs for a given time, to measure what a co-resident
// persistent launch (the LSTM sequence kernels: 32 workgroups x 256 threads x 84 KB of LDS) costs the image tower beside it.
//   mode 0: sleeps (holds its LDS and wave slots only)      mode 1: VALU-busy      mode 2: polls a global word between short sleeps
// hipcc --offload-arch=gfx950 -shared -fPIC -O2 occupy.hip -o liboccupy.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, int mode, unsigned *word, float *sink) {
    extern __shared__ float sh[];
    const long long t0 = wall_clock64();          // 100 MHz
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    unsigned seen = 0;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(127);
        } else if (mode == 1) {
#pragma unroll 16
            for (int i = 0; i < 512; ++i) a = a * b + 1e-7f;
        } else {
            seen += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (threadIdx.x == 0) sh[0] = a + (float)seen;
    __syncthreads();
    if (sink && sh[0] == 12345.678f) sink[blockIdx.x] = sh[0];
}

extern "C" int occupy_launch(int wgs, int lds_bytes, double ms, int mode, unsigned *word, float *sink, void *stream) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void *)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (long long)(ms * 1e5), mode, word, sink);
    return (int)hipGetLastError();
}
