// Two waves per SIMD (512-thread workgroup, 128 accumulator registers each): does the second wave hide the first one's
// VALU / LDS work?  Per wave and step: 32 MFMAs (32x32x2 f32, 8 accumulators) + 32 packed VALU + 8 ds_read_b128 + barrier
// (= half a Winograd K step); the same with 16x16x4 MFMAs (64 per wave and step, 32 accumulators of 4 registers).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    const f32x2 alo = {a[0], a[1]}, ahi = {a[2], a[3]}, blo = {b[0], b[1]}, bhi = {b[2], b[3]};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <int WORK, int SMALL>      // WORK: 0 = MFMA only, 1 = + VALU + LDS + barrier;  SMALL: 16x16x4 MFMAs
__global__ __launch_bounds__(512, 1) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 512) smem[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[8];
    f32x4 acs[32];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v[8], b[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = *(const f32x4 *)(smem + i * 256 + lane * 4);
        b[i] = *(const f32x4 *)(smem + 4096 + i * 256 + lane * 4);
        t[i] = v[i] * 0.5f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int n = ((g + 1) & 1) * 4;
            f32x4 nb[4], nv[4];
            if (WORK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) nb[q] = *(const f32x4 *)(smem + 4096 + ((it + n + q) & 15) * 256 + lane * 4);
                nv[0] = sub4(t[n], t[n + 2]); nv[1] = t[n + 1] + t[n + 2]; nv[2] = sub4(t[n + 2], t[n + 1]); nv[3] = sub4(t[n + 1], t[n + 3]);
                t[n] = sub4(v[n], nv[2]); t[n + 1] = v[n + 1] + nv[0]; t[n + 2] = sub4(nv[1], v[n + 2]); t[n + 3] = sub4(v[n + 3], nv[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!SMALL) {
#pragma unroll
                for (int pp = 0; pp < 4; pp += 2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[g * 4 + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + pp][j], b[g * 4 + pp][j], acc[g * 4 + pp], 0, 0, 0);
                        acc[g * 4 + pp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + pp + 1][j], b[g * 4 + pp + 1][j], acc[g * 4 + pp + 1], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            const int a = (g * 4 + pp) * 4 + rb * 2;
                            acs[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[g * 4 + pp][j], b[g * 4 + pp][j], acs[a], 0, 0, 0);
                            acs[a + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[g * 4 + pp][(j + 1) & 3], b[g * 4 + pp][j], acs[a + 1], 0, 0, 0);
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WORK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { b[n + q] = nb[q]; v[n + q] = nv[q]; }
            }
        }
        if (WORK) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acs[i][0] + acs[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += t[i][0];
    out[blockIdx.x * 512 + tid] = s;
}
template <int WORK, int SMALL> void run(const char *name) {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<WORK, SMALL><<<256, 512>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<WORK, SMALL><<<256, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wave and step: 32 MFMAs of 4096 flop (32x32x2) or 64 of 2048 (16x16x4): 131072 flop either way; 2048 waves
    printf("%-72s %.3f us/step  %.1f TFLOP/s\n", name, ms * 1e3 / iters, 2048.0 * iters * 131072.0 / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("2 waves/SIMD, 32x32x2: MFMA only");
        run<1, 0>("2 waves/SIMD, 32x32x2: + 32 pk VALU + 8 LDS + barrier per wave");
        run<0, 1>("2 waves/SIMD, 16x16x4: MFMA only");
        run<1, 1>("2 waves/SIMD, 16x16x4: + 32 pk VALU + 8 LDS + barrier per wave");
    }
    return 0;
}
