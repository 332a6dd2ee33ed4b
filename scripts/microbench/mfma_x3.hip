// fp32 GEMM products through the bf16 matrix cores ("bf16x3"): a = ah + am + al (three bf16 pieces, 24 mantissa bits),
// a*b ~ ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm, fp32 accumulation.  Six v_mfma_f32_32x32x16_bf16 (8 passes each)
// replace eight v_mfma_f32_32x32x2_f32 (16 passes each) per 16 reduction channels and 32x32 tile: 2.67x fewer matrix
// cycles -- IF the splitting VALU work (per A fragment, shared by NB column blocks; B pre-split) does not eat it.
// Measures both loops (one wave per SIMD x 4, 256 workgroups, registers only) and checks the accuracy of one tile.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/mfma_x3 scripts/microbench/mfma_x3.hip && scripts/microbench/mfma_x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 cvt8(f32x4 lo, f32x4 hi) {
    const bf16x4 l = __builtin_convertvector(lo, bf16x4), h = __builtin_convertvector(hi, bf16x4);
    return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void widen8(bf16x8 v, f32x4 &lo, f32x4 &hi) {
    const bf16x4 l = __builtin_shufflevector(v, v, 0, 1, 2, 3), h = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    lo = __builtin_convertvector(l, f32x4);
    hi = __builtin_convertvector(h, f32x4);
}
// three bf16 pieces of eight fp32 values
__device__ __forceinline__ void split3(f32x4 lo, f32x4 hi, bf16x8 &p0, bf16x8 &p1, bf16x8 &p2) {
    p0 = cvt8(lo, hi);
    f32x4 wl, wh;
    widen8(p0, wl, wh);
    lo -= wl; hi -= wh;
    p1 = cvt8(lo, hi);
    widen8(p1, wl, wh);
    lo -= wl; hi -= wh;
    p2 = cvt8(lo, hi);
}

template <int MODE, int NB>
__global__ __launch_bounds__(256, 1) void loop_kernel(const float *a_in, const float *b_in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NB];
    for (int b = 0; b < NB; ++b) for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    f32x4 alo = *reinterpret_cast<const f32x4 *>(a_in + lane * 8), ahi = *reinterpret_cast<const f32x4 *>(a_in + lane * 8 + 4);
    f32x4 blo[NB], bhi[NB];
    bf16x8 b0[NB], b1[NB], b2[NB];
    for (int b = 0; b < NB; ++b) {
        blo[b] = *reinterpret_cast<const f32x4 *>(b_in + (b * 64 + lane) * 8);
        bhi[b] = *reinterpret_cast<const f32x4 *>(b_in + (b * 64 + lane) * 8 + 4);
        split3(blo[b], bhi[b], b0[b], b1[b], b2[b]);        // weights: split once, outside the loop
    }
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(alo), "+v"(ahi));             // a fresh A fragment per step as far as the compiler knows
        if (MODE == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(alo[j], blo[b][j], acc[b], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ahi[j], bhi[b][j], acc[b], 0, 0, 0);
            }
        } else {
            bf16x8 a0, a1, a2;
            split3(alo, ahi, a0, a1, a2);
#pragma unroll
            for (int b = 0; b < NB; ++b) {                   // small terms first
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[b], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0[b], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2[b], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0[b], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1[b], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[b], acc[b], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int b = 0; b < NB; ++b) for (int e = 0; e < 16; ++e) s += acc[b][e];
    if (out) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// accuracy: C[32][32] = A[32][K] B[K][32], K = 512, both ways, against double
template <int MODE>
__global__ void tile_kernel(const float *A, const float *B, float *C, int K) {
    const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f32x4 alo, ahi, blo, bhi;
        for (int j = 0; j < 4; ++j) {      // lane (i, kh): row / column i, k = k0 + 8 kh + j (+4)
            alo[j] = A[i * K + k0 + 8 * kh + j];  ahi[j] = A[i * K + k0 + 8 * kh + 4 + j];
            blo[j] = B[(k0 + 8 * kh + j) * 32 + i];  bhi[j] = B[(k0 + 8 * kh + 4 + j) * 32 + i];
        }
        if (MODE == 0) {
            // 32x32x2: lane (i, kh) supplies k = kh of each pair -> feed pairs (j of lo with kh) consistently for A and B
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(alo[j], blo[j], acc, 0, 0, 0);
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ahi[j], bhi[j], acc, 0, 0, 0);
        } else {
            bf16x8 a0, a1, a2, b0, b1, b2;
            split3(alo, ahi, a0, a1, a2);
            split3(blo, bhi, b0, b1, b2);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + i] = acc[e];
}

template <int MODE, int NB>
float time_loop(const float *a, const float *b, float *o, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<MODE, NB>), dim3(256), dim3(256), 0, 0, a, b, o, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((loop_kernel<MODE, NB>), dim3(256), dim3(256), 0, 0, a, b, o, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f / iters;          // us per 16-channel step
}

int main() {
    const int K = 512;
    std::vector<float> A(32 * K), B(K * 32);
    srand(7);
    for (auto &v : A) v = fmaxf(0.f, (rand() / (float)RAND_MAX - 0.3f) * 3.f);        // post-ReLU-like
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *dA, *dB, *dC, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&dO, 256 * 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j]; ref[i * 32 + j] = s; }
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
        else hipLaunchKernelGGL(tile_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
        std::vector<float> C(1024);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double e2 = 0, r2 = 0, emax = 0;
        for (int i = 0; i < 1024; ++i) { const double d = C[i] - ref[i]; e2 += d * d; r2 += ref[i] * ref[i]; emax = fmax(emax, fabs(d)); }
        printf("%s: relative rms error %.3e, max abs error %.3e (|ref| rms %.3f)\n", mode ? "bf16x3 (6 x 32x32x16 bf16)" : "fp32   (8 x 32x32x2 f32)  ", sqrt(e2 / r2), emax, sqrt(r2 / 1024));
    }
    const int iters = 4000;
    printf("us per 16-channel step of a wave (4 waves per CU, 256 CUs), registers only:\n");
    printf("  NB = 2: fp32 %.3f   bf16x3 %.3f\n", time_loop<0, 2>(dA, dB, dO, iters), time_loop<1, 2>(dA, dB, dO, iters));
    printf("  NB = 4: fp32 %.3f   bf16x3 %.3f\n", time_loop<0, 4>(dA, dB, dO, iters), time_loop<1, 4>(dA, dB, dO, iters));
    printf("  NB = 8: fp32 %.3f   bf16x3 %.3f\n", time_loop<0, 8>(dA, dB, dO, iters), time_loop<1, 8>(dA, dB, dO, iters));
    return 0;
}
