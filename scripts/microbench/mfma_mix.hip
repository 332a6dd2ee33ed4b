// What a wave loses around its MFMAs: 1 wave per SIMD, 16 accumulators, 64 MFMAs per step (the Winograd K step),
// plus V VALU ops, L LDS reads, optional barrier per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VALU, int LDS, int BAR, int IL>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) smem[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 v[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = *(const f32x4 *)(smem + i * 256 + lane * 4); b[i] = *(const f32x4 *)(smem + 4096 + i * 256 + lane * 4); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (LDS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) b[g * 4 + q] = *(const f32x4 *)(smem + 4096 + ((it + g * 4 + q) & 15) * 256 + lane * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (IL == 2) {
#pragma unroll
            for (int pp = 0; pp < 4; pp += 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g * 4 + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + pp][j], b[g * 4 + pp][j], acc[g * 4 + pp], 0, 0, 0);
                    acc[g * 4 + pp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + pp + 1][j], b[g * 4 + pp + 1][j], acc[g * 4 + pp + 1], 0, 0, 0);
                }
            } else if (IL == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int px = 0; px < 4; ++px)
                        acc[g * 4 + px] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + px][j], b[g * 4 + px][j], acc[g * 4 + px], 0, 0, 0);
            } else {
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[g * 4 + px] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + px][j], b[g * 4 + px][j], acc[g * 4 + px], 0, 0, 0);
            }
            if (VALU) {      // 8 f32x4 ops = 32 VALU per group = 128 per step, on the operands of the NEXT group
                const int n = ((g + 1) & 3) * 4;
#pragma unroll
                for (int r = 0; r < VALU; ++r) {
                    v[n] = v[n] - v[n + 2]; v[n + 1] = v[n + 1] + v[n + 2]; v[n + 2] = v[n + 2] - v[n + 1]; v[n + 3] = v[n + 1] - v[n + 3];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + tid] = s;
}
template <int VALU, int LDS, int BAR, int IL> void run(const char *name) {
    float *out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<VALU, LDS, BAR, IL><<<256, 256>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<VALU, LDS, BAR, IL><<<256, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f us/step  %.1f TFLOP/s\n", name, ms * 1e3 / iters, 1024.0 * iters * 64 * 4096.0 / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
    run<0, 0, 0, 1>("64 MFMA, dependent chain of 4");
    run<0, 0, 0, 2>("64 MFMA, 2 accumulators alternating");
    run<0, 0, 0, 4>("64 MFMA, 4 accumulators round robin");
    run<2, 1, 1, 2>("IL2 + 128 VALU + 16 LDS + barrier");
    run<2, 1, 1, 4>("IL4 + 128 VALU + 16 LDS + barrier");
    run<0, 0, 1, 2>("IL2 + barrier");
    run<0, 0, 1, 4>("IL4 + barrier");
    }
    return 0;
}
