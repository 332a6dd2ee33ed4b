#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode 0: pure MFMA (4 acc), mode 1: + LDS frag reads each K-step, mode 2: + barrier each K-step
template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float smem[2*(256+192)*20];
  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6, li = lane & 31, lk = lane >> 5;
  for (int i = tid; i < 2*(256+192)*20; i += 256) smem[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[MT][NT];
  for (int a = 0; a < MT; ++a) for (int b = 0; b < NT; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float af[MT][8], bf[NT][8];
  for (int a = 0; a < MT; ++a) for (int s = 0; s < 8; ++s) af[a][s] = smem[(wm*64 + a*32 + li)*20 + lk*8 + s];
  for (int b = 0; b < NT; ++b) for (int s = 0; s < 8; ++s) bf[b][s] = smem[256*20*2 + (b*32 + li)*20 + lk*8 + s];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
      const float* a_s = smem + (it & 1) * 256 * 20 + (wm * MT * 32 + li) * 20 + lk * 8;
      const float* b_s = smem + 2*256*20 + (it & 1) * 192 * 20 + li * 20 + lk * 8;
#pragma unroll
      for (int a = 0; a < MT; ++a) { f32x4 lo = *(const f32x4*)(a_s + a*32*20), hi = *(const f32x4*)(a_s + a*32*20 + 4);
        af[a][0]=lo.x; af[a][1]=lo.y; af[a][2]=lo.z; af[a][3]=lo.w; af[a][4]=hi.x; af[a][5]=hi.y; af[a][6]=hi.z; af[a][7]=hi.w; }
#pragma unroll
      for (int b = 0; b < NT; ++b) { f32x4 lo = *(const f32x4*)(b_s + b*32*20), hi = *(const f32x4*)(b_s + b*32*20 + 4);
        bf[b][0]=lo.x; bf[b][1]=lo.y; bf[b][2]=lo.z; bf[b][3]=lo.w; bf[b][4]=hi.x; bf[b][5]=hi.y; bf[b][6]=hi.z; bf[b][7]=hi.w; }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
    if (MODE >= 2) __syncthreads();
  }
  float s = 0; for (int a = 0; a < MT; ++a) for (int b = 0; b < NT; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE, int MT, int NT> void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE,MT,NT><<<blocks, 256>>>(out, 100); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE,MT,NT><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 8 * MT * NT * 4096.0;
  printf("%-28s blocks %4d  MT%d NT%d  %.1f TFLOP/s  (%.2f ms)\n", name, blocks, MT, NT, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  run<0,2,2>("pure mfma", 256); run<0,2,2>("pure mfma", 512); run<0,2,2>("pure mfma", 768);
  run<1,2,2>("+lds frags", 256); run<1,2,2>("+lds frags", 512);
  run<2,2,2>("+barrier", 256); run<2,2,2>("+barrier", 512); run<2,2,2>("+barrier", 768);
  run<2,2,4>("+barrier", 256); run<2,2,6>("+barrier", 256); run<2,1,2>("+barrier", 512); run<2,1,2>("+barrier", 1024);
  return 0;
}
