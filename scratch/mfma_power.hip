// Does operand entropy / run length change the achievable fp32 MFMA rate (power-limited clocks)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ float rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f); }
template <int NT>
__global__ __launch_bounds__(256) void k(float* out, int iters, int random) {
  const int tid = threadIdx.x;
  f32x16 acc[NT];
  for (int b = 0; b < NT; ++b) for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  float af[8], bf[NT][8];
  for (int s = 0; s < 8; ++s) af[s] = random ? rnd(tid * 131 + s + blockIdx.x * 7919) : 0.01f * (s % 3);
  for (int b = 0; b < NT; ++b) for (int s = 0; s < 8; ++s) bf[b][s] = random ? rnd(tid * 977 + s * 13 + b * 101 + 5) * 0.01f : 0.02f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[b][s], acc[b], 0, 0, 0);
  }
  float s = 0; for (int b = 0; b < NT; ++b) for (int r = 0; r < 16; ++r) s += acc[b][r];
  out[blockIdx.x * 256 + tid] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int random = 0; random < 2; ++random)
    for (int iters : {4000, 400000}) {
      const int blocks = 768;
      k<2><<<blocks, 256>>>(out, 100, random); hipDeviceSynchronize();
      hipEventRecord(e0); k<2><<<blocks, 256>>>(out, iters, random); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 8 * 2 * 4096.0;
      printf("%s operands, %7d iters: %.1f TFLOP/s (%.2f ms)\n", random ? "random " : "constant", iters, flops / ms / 1e9, ms);
    }
  return 0;
}
