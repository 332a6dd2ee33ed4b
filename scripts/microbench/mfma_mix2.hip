// Can a single wave (one per SIMD, 16 accumulators = the Winograd K step) hide its VALU / LDS work under its own MFMAs?
// Variants of the K step: 64 MFMAs (32x32x2 f32) + 64 packed VALU ops (the transform) + 16 ds_read_b128 + barrier,
//   mode 0: MFMAs only
//   mode 1: per 16-MFMA group, VALU and LDS reads AFTER the MFMAs, fenced with sched_barrier(0)   (round-2 structure)
//   mode 2: same instructions, no fences at all (compiler's own schedule)
//   mode 3: sched_group_barrier pipeline: {1 MFMA, 1 VALU} x 16 per group, LDS reads in the first slots
//   mode 4: as 3 but {1 MFMA, 2 VALU} x 8 then 8 MFMAs
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

template <int MODE>
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
    f32x4 v[16], b[16], t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = *(const f32x4 *)(smem + i * 256 + lane * 4);
        b[i] = *(const f32x4 *)(smem + 4096 + i * 256 + lane * 4);
        t[i] = v[i] * 0.5f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = ((g + 1) & 3) * 4;      // operands of the NEXT group are produced during this one
            f32x4 nb[4], nv[4];
            if (MODE != 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) nb[q] = *(const f32x4 *)(smem + 4096 + ((it + n + q) & 15) * 256 + lane * 4);
                // 16 packed ops: a column pass (8) + a quarter of a row pass (8)
                nv[0] = sub4(t[n], t[n + 2]); nv[1] = t[n + 1] + t[n + 2]; nv[2] = sub4(t[n + 2], t[n + 1]); nv[3] = sub4(t[n + 1], t[n + 3]);
                t[n] = sub4(v[n], nv[2]); t[n + 1] = v[n + 1] + nv[0]; t[n + 2] = sub4(nv[1], v[n + 2]); t[n + 3] = sub4(v[n + 3], nv[3]);
            }
            if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < 4; pp += 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g * 4 + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + pp][j], b[g * 4 + pp][j], acc[g * 4 + pp], 0, 0, 0);
                    acc[g * 4 + pp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g * 4 + pp + 1][j], b[g * 4 + pp + 1][j], acc[g * 4 + pp + 1], 0, 0, 0);
                }
            if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
            if (MODE != 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { b[n + q] = nb[q]; v[n + q] = nv[q]; }
            }
            if (MODE == 3) {
                // 4 LDS reads first, then 16 x {MFMA, VALU}
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                for (int r = 0; r < 16; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            }
            if (MODE == 4) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                for (int r = 0; r < 8; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
            }
        }
        if (MODE != 0) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += t[i][0];
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(const char *name) {
    float *out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.3f us/step  %.1f TFLOP/s\n", name, ms * 1e3 / iters, 1024.0 * iters * 64 * 4096.0 / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("64 MFMA only");
        run<1>("fenced groups: [LDS, VALU] | 16 MFMA |  (round-2 structure)");
        run<2>("same instructions, compiler's own schedule");
        run<3>("sched_group_barrier {MFMA, VALU} x 16");
        run<4>("sched_group_barrier MFMA, 4 LDS, {MFMA, 2 VALU} x 8, 7 MFMA");
    }
    return 0;
}
