// Shared helpers for the gfx950 kernel library (internal; the public ABI is include/ds_kernels.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "ds_kernels.h"

namespace ds {

void set_error(const char *fmt, ...);

// Tuning knobs (DS_* environment variables read by the selection rules: tile pins, cost-model coefficients, A/B switches)
// exist in the -DDS_TUNING build only (libds_kernels_tuning.so, what scripts/ and the kernel tests load); the shipped
// library reads no environment variable and every knob has its default.
#ifdef DS_TUNING
inline const char *tune_env(const char *name) { return getenv(name); }
#else
inline const char *tune_env(const char *) { return nullptr; }
#endif

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DS_ERR_LAUNCH;
    }
    return DS_OK;
}

// MI355X: 256 CUs in 8 XCDs.  Memory-bound kernels cap their grid at 8 blocks per CU and
// grid-stride the rest (cdna_hip_programming.md Guideline 11).
constexpr int kCUs = 256;
constexpr int kMaxStreamBlocks = kCUs * 8;

// grid of the grid-stride streaming kernels (pools, copies, reductions): at most kMaxStreamBlocks workgroups
// (DS_STREAM_MAX = workgroups per CU, tuning library only)
inline int stream_grid(int64_t work_items, int per_block) {
    static int cap = 0;
    if (!cap) {
        const char *e = tune_env("DS_STREAM_MAX");
        cap = e && atoi(e) > 0 ? kCUs * atoi(e) : kMaxStreamBlocks;
    }
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// 16-byte load of data that is read once (streaming): non-temporal hint, so the lines do not displace what the
// neighbouring kernels keep in L2 / the Infinity Cache (scripts/microbench/stream_bw.hip)
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream4(const float *p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// *word = max(*word, v) for non-negative floats (they order like unsigned integers, so the result is exact and
// independent of the order).  Thousands of waves aim at one word: each first LOOKS at it (an agent-scope load, served by
// L2) and only the few whose value still exceeds it pay for the atomic -- with the plain atomic per wave the BatchNorm
// apply kernels ran 3-6x longer.
// A max|.| RECORD is DS_AMAX_FLOATS floats: 16 slots, one per 128-byte line, so that the waves of a launch spread over
// 16 L2 lines instead of queueing on one word; the reader takes the maximum of the 16 slots (amax_read).
__device__ __forceinline__ void atomic_max_nonneg(float *record, float v) {
    if (!(v > 0.f)) return;
    const unsigned bits = __float_as_uint(v);
    const unsigned slot = (blockIdx.x + (threadIdx.x >> 6)) & 15u;
    unsigned *w = reinterpret_cast<unsigned *>(record) + 32u * slot;
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= bits) return;
    atomicMax(w, bits);
}

// max over the 16 slots of a record (every lane gets it); a finished producer launch is assumed (stream order)
__device__ __forceinline__ float amax_read(const float *record) {
    float m = record[32 * (threadIdx.x & 15)];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ds_bn_finalize of ONE channel by ONE wave, for P <= 256 partials: bn_finalize_kernel's summation tree (thread t of 256 takes
// partial t, a butterfly per wave, then ((w0 + w1) + w2) + w3 in double) evaluated by 64 lanes, so the results have the bits of
// the separate launch.  The partials are read with agent-scope loads (they were published write-through by other workgroups
// of the SAME launch).  stats: [2][C][P].
__device__ __forceinline__ void bn_finalize_channel_by_wave(const float *stats, int P, int C, int c, double inv_count,
                                                           const ds_bn_finalize_in_launch &f, const float *pivot) {
    const int lane = threadIdx.x & 63;
    double sw[4], qw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int pidx = 64 * w + lane;
        double s = 0.0, q = 0.0;
        if (pidx < P) {
            const unsigned *sp = reinterpret_cast<const unsigned *>(stats + (int64_t)c * P + pidx);
            const unsigned *qp = reinterpret_cast<const unsigned *>(stats + ((int64_t)C + c) * P + pidx);
            s = (double)__uint_as_float(__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            q = (double)__uint_as_float(__hip_atomic_load(qp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        sw[w] = wave_sum_f64(s);
        qw[w] = wave_sum_f64(q);
    }
    if (lane == 0) {
        const double s = ((sw[0] + sw[1]) + sw[2]) + sw[3];
        const double q = ((qw[0] + qw[1]) + qw[2]) + qw[3];
        const double du = s * inv_count;
        const double mu = du + (pivot ? (double)pivot[c] : 0.0);      // pivot may alias mean: read before the write below
        double var = q * inv_count - du * du;          // biased variance (A3)
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)f.eps));
        f.mean[c] = (float)mu;
        f.rstd[c] = r;
        f.shift[c] = f.beta[c] - (float)mu * r;
        if (f.moving_mean) f.moving_mean[c] = f.decay * f.moving_mean[c] + (1.f - f.decay) * (float)mu;     // assign_moving_average
        if (f.moving_var) f.moving_var[c] = f.decay * f.moving_var[c] + (1.f - f.decay) * (float)var;
    }
}

// slim.batch_norm (scale = False) + ReLU backward for one element: dz from the layer's pre-BatchNorm z, the gradient dy of
// its activation and the per-channel rstd r, shift s, mean mu, column means a1 = mean(g), a2 = mean(g xhat).  ONE
// definition, explicit fused multiply-adds: ds_bn_bwd_apply and the wide dgrad's on-load form give the same bits.
__device__ __forceinline__ float bn_bwd_dz(float z, float dy, float r, float s, float mu, float a1, float a2) {
    const float g = __builtin_fmaf(z, r, s) > 0.f ? dy : 0.f;
    const float xh = (z - mu) * r;
    return r * __builtin_fmaf(-xh, a2, g - a1);
}

}  // namespace ds

#define DS_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ds::set_error(__VA_ARGS__);       \
            return DS_ERR_ARG;                \
        }                                     \
    } while (0)
