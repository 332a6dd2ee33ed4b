// BatchNorm (train mode, beta only) + ReLU, forward and backward.  HBM-bound streaming kernels.
//
// Restates slim.batch_norm(center=True, scale=False, decay=0.9997, epsilon=1e-3) as configured by
// slim/nets/inception_utils.py:48-70, applied after every slim.conv2d of
// image_model/inception_v1.py:63-250, followed by tf.nn.relu (inception_utils.py:68).
//   forward : mean/var come from the conv kernel's column-statistics partials (ds_bn_finalize),
//             y = relu(z*rstd + (beta - mean*rstd)) is scattered straight into the channel slices
//             of the Inception concat buffer (replaces tf.concat, inception_v1.py:96..248).
//   backward: g = dy*(y>0); dbeta = sum(g); dz = rstd*(g - mean(g) - xhat*mean(g*xhat)).
// All kernels move 16 B per lane per access and use a fixed (deterministic) reduction order.
#include <stdlib.h>
#include "ds_common.h"

namespace {

struct SegDev {
    int nseg;
    int c_begin[4], c_end[4], ld[4];
    float *ptr[4];
    int dtype[4];
    float *amax[4];
    const float *ptr2[4];      // bn_bwd_apply: second addend of the segment (ds_segments.ptr2)
};

SegDev to_dev(const ds_segments *s) {
    SegDev o;
    o.nseg = s->nseg;
    for (int i = 0; i < 4; ++i) {
        o.c_begin[i] = s->c_begin[i];
        o.c_end[i] = s->c_end[i];
        o.ld[i] = s->ld[i];
        o.ptr[i] = (float *)s->ptr[i];
        o.dtype[i] = s->dtype[i];
        o.amax[i] = s->amax[i];
        o.ptr2[i] = s->ptr2[i];
    }
    return o;
}

__device__ __forceinline__ float *seg_addr(const SegDev &sg, int64_t row, int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < sg.nseg && c >= sg.c_begin[i] && c < sg.c_end[i])
            return sg.ptr[i] + row * sg.ld[i] + (c - sg.c_begin[i]);
    return nullptr;
}

// z in fp32, or -- when ds_bn_bwd_reduce runs on a POOLED activation kept in 16-bit storage -- bf16
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ float4 ldz4(const T *p);
template <>
__device__ __forceinline__ float4 ldz4<float>(const float *p) { return ds::ld_stream4(p); }
template <>
__device__ __forceinline__ float4 ldz4<__bf16>(const __bf16 *p) {
    const f32x4_t v = __builtin_convertvector(*reinterpret_cast<const bf16x4_t *>(p), f32x4_t);
    return make_float4(v[0], v[1], v[2], v[3]);
}

using ds::wave_sum_f64;

#ifndef DS_BN_ROWS
#define DS_BN_ROWS 2          // rows per pass of the fixed-column streaming kernels (tuning builds: -DDS_BN_ROWS=4)
#endif

// Hand-off INSIDE a launch (the finalize + apply kernels below): the producer's few per-channel results go out as agent-scope
// relaxed atomic stores (write-through: visible at the device's coherence point once vmcnt drains), the consumers read them
// with agent-scope relaxed atomic loads (never a stale line of their XCD's L2).  No release / acquire fences: at agent scope
// those write back / invalidate a whole L2 per workgroup (measured: +80 us per launch).
template <bool COH>
__device__ __forceinline__ void st_res(float *p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH>
__device__ __forceinline__ float4 ld_res4(const float *p) {
    if constexpr (COH) {
        float4 v;
        v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    } else {
        return *reinterpret_cast<const float4 *>(p);
    }
}

// ---- forward ------------------------------------------------------------------------------------
// one workgroup per channel; partials are laid out [2][C][P] so the threads read contiguous floats
// (P is a few hundred for the persistent conv launches, one per row tile -- up to 2048 -- otherwise);
// combined in double in a fixed order: strided per thread, butterfly per wave, waves 0..3 (deterministic)
template <bool COH = false>
__device__ __forceinline__ void finalize_channel(const float *stats, int P, double inv_count, int C, int c, const float *beta,
                                                 const float *pivot, float eps, float decay, float *mean, float *rstd,
                                                 float *shift, float *mm, float *mv, float *mean_c = nullptr,
                                                 float *shift_c = nullptr) {
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) {
        s += (double)stats[(int64_t)c * P + p];
        q += (double)stats[((int64_t)C + c) * P + p];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        q = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        // the partials are sums of (z - pivot) and (z - pivot)^2: with the pivot near the mean the fp32 partial sums
        // carry the spread of z, not its offset, so E[u^2] - E[u]^2 does not cancel (|mean| >> std channels)
        const double du = s * inv_count;
        const double mu = du + (pivot ? (double)pivot[c] : 0.0);      // pivot may alias mean: read before the write below
        double var = q * inv_count - du * du;          // biased variance (A3)
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = (float)mu;
        st_res<COH>(rstd + c, r);
        st_res<COH>(shift + c, beta[c] - (float)mu * r);
        if (mean_c) {                 // z stored centred about the pivot (ds_bn_finalize_centered)
            mean_c[c] = (float)du;
            shift_c[c] = beta[c] - (float)du * r;
        }
        if (mm) mm[c] = decay * mm[c] + (1.f - decay) * (float)mu;     // assign_moving_average
        if (mv) mv[c] = decay * mv[c] + (1.f - decay) * (float)var;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *stats, int P, double inv_count, int C,
                                                          const float *beta, const float *pivot, float eps,
                                                          float decay, float *mean, float *rstd, float *shift,
                                                          float *mm, float *mv, float *mean_c, float *shift_c) {
    finalize_channel(stats, P, inv_count, C, (int)blockIdx.x, beta, pivot, eps, decay, mean, rstd, shift, mm, mv, mean_c, shift_c);
}

// ds_bn_finalize_multi: the channels of up to four layers in one grid (workgroup -> job by the running channel count)
struct FinJobsDev {
    int njobs;
    int first[4];              // first workgroup of the job
    int C[4], P[4];
    double inv_count[4];
    const float *stats[4], *beta[4], *pivot[4];
    float *mean[4], *rstd[4], *shift[4], *mm[4], *mv[4];
};

__global__ __launch_bounds__(256) void bn_finalize_multi_kernel(FinJobsDev jb, float eps, float decay) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < jb.njobs && (int)blockIdx.x >= jb.first[i]) j = i;
    finalize_channel(jb.stats[j], jb.P[j], jb.inv_count[j], jb.C[j], (int)blockIdx.x - jb.first[j], jb.beta[j], jb.pivot[j], eps,
                     decay, jb.mean[j], jb.rstd[j], jb.shift[j], jb.mm[j], jb.mv[j]);
}

// Streaming kernels (bn_apply_relu, bn_bwd_apply): the launch has gridDim.x * 256 = drow * (C / 4) threads, so a thread
// keeps ONE float4 column group for its whole walk down the rows (row += drow): the per-channel values and the
// segment lookup happen once, the loop has no division, and consecutive threads still read consecutive addresses.  Tensors
// that are read once come in with the non-temporal hint (scripts/microbench/stream_bw.hip: 5.3 -> 6.4 TB/s for "two in,
// one out"); two rows per pass with both rows' loads before either store (a load behind a store waits for it).
template <bool COH = false, bool Z16 = false>
__device__ __forceinline__ void apply_relu_body(int bid, const float *z, int64_t M, int C, const float *rstd,
                                                const float *shift, const SegDev &dst, int drow) {
    const int C4 = C >> 2;
    const int t0 = bid * 256 + threadIdx.x;
    const int row0 = t0 / C4, c = (t0 - row0 * C4) * 4;
    const float4 r = ld_res4<COH>(rstd + c), s = ld_res4<COH>(shift + c);
    int sgi = 0;                                 // this column group's destination segment
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < dst.nseg && c >= dst.c_begin[i] && c < dst.c_end[i]) sgi = i;
    const bool to16 = dst.dtype[sgi] == DS_DTYPE_BF16;
    float *const dptr = dst.ptr[sgi];
    const int64_t dld = dst.ld[sgi];
    const int dc = c - dst.c_begin[sgi];
    float smax = 0.f;                            // max(y) of this segment (y >= 0), if it asks
    constexpr int NR = DS_BN_ROWS;
    for (int64_t row = row0; row < M; row += NR * (int64_t)drow) {
        bool ok[NR];
        float4 vv[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            ok[u] = row + u * (int64_t)drow < M;
            const int64_t zo = (ok[u] ? row + u * (int64_t)drow : row) * C + c;
            if constexpr (Z16) vv[u] = ldz4<__bf16>(reinterpret_cast<const __bf16 *>(z) + zo);      // z stored as bf16 (ds_conv_desc.z_dtype)
            else vv[u] = ds::ld_stream4(z + zo);
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            if (u > 0 && !ok[u]) break;
            const float4 v = vv[u];
            float4 y;
            y.x = fmaxf(v.x * r.x + s.x, 0.f);
            y.y = fmaxf(v.y * r.y + s.y, 0.f);
            y.z = fmaxf(v.z * r.z + s.z, 0.f);
            y.w = fmaxf(v.w * r.w + s.w, 0.f);
            smax = fmaxf(smax, fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)));
            const int64_t e = (row + u * (int64_t)drow) * dld + dc;
            // destination segment: fp32, or bf16 (16-bit activation storage; round to nearest even)
            if (to16) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                const f32x4v yv = {y.x, y.y, y.z, y.w};
                *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(dptr) + e) = __builtin_convertvector(yv, bf16x4);
            } else {
                *reinterpret_cast<float4 *>(dptr + e) = y;
            }
        }
    }
    // non-negative floats order like unsigned integers.  (A wave may span two segments: every segment is reduced.)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= dst.nseg || !dst.amax[i]) continue;          // (uniform)
        float m = sgi == i ? smax : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) ds::atomic_max_nonneg(dst.amax[i], m);
    }
}

template <bool Z16 = false>
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float *z, int64_t M, int C, const float *rstd,
                                                            const float *shift, SegDev dst, int drow) {
    apply_relu_body<false, Z16>((int)blockIdx.x, z, M, C, rstd, shift, dst, drow);
}

// ---- finalize + apply as ONE launch --------------------------------------------------------------------------------------
// A finalize is one workgroup per channel and ~1 us of work; the apply pass that reads its result used to be the next launch
// on the same stream, i.e. a dependent-launch boundary (~4 us of idle GPU + the launch's own ramp) for every BatchNorm layer
// and direction -- at 32 samples per GPU a tenth of the step.  Here the first C workgroups of the launch finalize their
// channel and publish (release increment of ticket[0]); the others are the apply pass: thread 0 spins on ticket[0] == C
// (acquire), then the workgroup reads the per-channel vectors and streams as before.  No deadlock: a grid's workgroups are
// dispatched in id order, so whenever an apply workgroup occupies a slot every finalize workgroup has been placed already
// and runs to completion.  The last apply workgroup to pass its end (ticket[1]) zeroes both words for the next launch.
// Arithmetic and summation order are the separate launches': bit-identical.
__device__ __forceinline__ void fa_publish(unsigned *ticket) {
    if (threadIdx.x == 0) {          // (thread 0 wrote the channel's results: st_res<true>)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void fa_wait(unsigned *ticket, unsigned need) {
    if (threadIdx.x == 0) {
        unsigned n = 0;
        // (a thousand workgroups polling one word every few hundred ns queue up at its memory channel in front of the very
        // increments they wait for: first poll after ~3 us, then every ~1.5 us)
        __builtin_amdgcn_s_sleep(127);
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(60);
            if (++n > (1u << 21)) __builtin_trap();          // (seconds: cannot happen, see above -- fail loudly, never hang)
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void fa_leave(unsigned *ticket, unsigned napply) {
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == napply - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct FinApplyArgs {
    const float *stats; int P; double inv_count; int C;
    const float *beta, *pivot; float eps, decay;
    float *mean, *rstd, *shift, *mm, *mv;
    const float *z; int64_t M; SegDev dst; int drow;
    unsigned *ticket;
};

__global__ __launch_bounds__(256) void bn_finalize_apply_relu_kernel(const FinApplyArgs a) {
    if ((int)blockIdx.x < a.C) {
        finalize_channel<true>(a.stats, a.P, a.inv_count, a.C, (int)blockIdx.x, a.beta, a.pivot, a.eps, a.decay, a.mean, a.rstd,
                               a.shift, a.mm, a.mv);
        fa_publish(a.ticket);
        return;
    }
    fa_wait(a.ticket, (unsigned)a.C);
    apply_relu_body<true>((int)blockIdx.x - a.C, a.z, a.M, a.C, a.rstd, a.shift, a.dst, a.drow);
    fa_leave(a.ticket, gridDim.x - (unsigned)a.C);
}

// ---- backward -----------------------------------------------------------------------------------
// Each workgroup owns a contiguous block of rows; a thread owns one float4 column group and strides
// over the rows, so every wave reads whole contiguous rows (coalesced) and the per-channel sums stay
// in registers until one LDS combine at the end.
// rows per workgroup: 512 for the big maps, fewer on the 14x14 / 7x7 maps so that ~512 workgroups (two per CU)
// remain: with ~2000 the reduce of the small maps was 15-25 % slower (a thread then sums ~10 rows and the LDS
// combine and the partial stores weigh as much as the loads); DS_BN_BWD_BLOCKS overrides the target
int bwd_rows_per_block(int64_t M) {
    static int target = -1;
    if (target < 0) {
        const char *e = ds::tune_env("DS_BN_BWD_BLOCKS");
        target = e ? atoi(e) : 512;
    }
    int64_t r = (M + target - 1) / target;
    r = (r + 7) / 8 * 8;
    if (r < 16) r = 16;
    if (r > 512) r = 512;
    return (int)r;
}

template <typename TZ>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const TZ *z, int ldz, SegDev dy, int64_t M, int C,
                                                            const float *mean, const float *rstd,
                                                            const float *shift, float *partials, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [RG][C4][8]
    const int C4 = C >> 2;
    const int RG = 256 / C4 > 0 ? 256 / C4 : 1;
    const int tid = threadIdx.x;
    const int cg = tid % C4, rg = tid / C4;
    const bool active = tid < RG * C4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int c = cg * 4;
        const float4 r = *reinterpret_cast<const float4 *>(rstd + c);
        const float4 s = *reinterpret_cast<const float4 *>(shift + c);
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c);
        const float rr[4] = {r.x, r.y, r.z, r.w}, ss[4] = {s.x, s.y, s.z, s.w}, mm[4] = {mu.x, mu.y, mu.z, mu.w};
        // four rows per trip: eight independent 16-byte loads in flight per lane before the first use
        // (the sums stay in row order, so the result does not depend on the unrolling)
        int64_t row = r0 + rg;
        for (; row + 3 * RG < r1; row += 4 * RG) {
            float4 zv[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                zv[u] = ldz4<TZ>(z + (row + u * RG) * ldz + c);
                dv[u] = ds::ld_stream4(seg_addr(dy, row + u * RG, c));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = (zz[j] * rr[j] + ss[j] > 0.f) ? dd[j] : 0.f;
                    sg[j] += g;
                    sx[j] += g * ((zz[j] - mm[j]) * rr[j]);
                }
            }
        }
        for (; row < r1; row += RG) {
            const float4 zv = ldz4<TZ>(z + row * ldz + c);
            const float4 dv = ds::ld_stream4(seg_addr(dy, row, c));
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = (zz[j] * rr[j] + ss[j] > 0.f) ? dd[j] : 0.f;
                sg[j] += g;
                sx[j] += g * ((zz[j] - mm[j]) * rr[j]);
            }
        }
        float *o = sh + ((int64_t)rg * C4 + cg) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = sg[j];
            o[4 + j] = sx[j];
        }
    }
    __syncthreads();
    if (tid < C4) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += sh[((int64_t)g * C4 + tid) * 8 + j];
        const int P = gridDim.x;                       // partials laid out [2][C][P]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            partials[(int64_t)(tid * 4 + j) * P + blockIdx.x] = a[j];
            partials[((int64_t)C + tid * 4 + j) * P + blockIdx.x] = a[4 + j];
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *partials, int P, double inv_count, int C,
                                                              float *dbeta, float *coef) {
    // one workgroup per channel, same fixed combination order as bn_finalize_kernel
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) {
        s += (double)partials[(int64_t)c * P + p];
        q += (double)partials[((int64_t)C + c) * P + p];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        q = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        dbeta[c] = (float)s;
        if (coef) {
            coef[c] = (float)(s * inv_count);          // mean(g)
            coef[C + c] = (float)(q * inv_count);      // mean(g*xhat)
        }
    }
}

// Finalize from up to four column segments whose sums come from different producers:
//   kind 0: partials of ds_bn_bwd_reduce over that column range          (sum g, sum g*xhat)
//   kind 1: partials a dgrad kernel emitted with DS_EPI_BNSUMS            (sum g, sum g*y over y > 0)
// For y > 0, y = xhat + beta (y = z*rstd + beta - mean*rstd), so sum g*xhat = sum g*y - beta * sum g.
struct SumSegDev {
    int nseg;
    int c_begin[4], c_end[4], P[4], kind[4];
    const float *s[4], *q[4];
    int P2[4];                 // second source of the segment (ds_bn_sum_segments.P2 / s2 / q2), 0: none
    const float *s2[4], *q2[4];
    const float *beta[4];      // the segment's beta / dbeta, indexed from its first channel (ds_bn_bwd_finalize_segs: one
    float *dbeta[4];           // vector offset per segment; ds_bn_bwd_finalize_multi: the layers' own vectors); nullable
};

template <bool COH = false>
__device__ __forceinline__ void bwd_finalize_segs_channel(const SumSegDev &sg, double inv_count, int C, float *coef, int c) {
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int P = 0, kind = 0, P2 = 0;
    const float *sp = nullptr, *qp = nullptr, *beta = nullptr, *sp2 = nullptr, *qp2 = nullptr;
    float *dbeta = nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < sg.nseg && c >= sg.c_begin[i] && c < sg.c_end[i]) {
            P = sg.P[i];
            kind = sg.kind[i];
            sp = sg.s[i] + (int64_t)(c - sg.c_begin[i]) * P;
            qp = sg.q[i] + (int64_t)(c - sg.c_begin[i]) * P;
            P2 = sg.P2[i];
            if (P2 > 0) {
                sp2 = sg.s2[i] + (int64_t)(c - sg.c_begin[i]) * P2;
                qp2 = sg.q2[i] + (int64_t)(c - sg.c_begin[i]) * P2;
            }
            beta = sg.beta[i] ? sg.beta[i] + (c - sg.c_begin[i]) : nullptr;
            dbeta = sg.dbeta[i] ? sg.dbeta[i] + (c - sg.c_begin[i]) : nullptr;
        }
    double s = 0.0, q = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) {
        s += (double)sp[p];
        q += (double)qp[p];
    }
    for (int p = threadIdx.x; p < P2; p += 256) {          // (second source: the other addend's sums)
        s += (double)sp2[p];
        q += (double)qp2[p];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        q = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        if (kind == 1) q -= (double)*beta * s;
        if (dbeta) *dbeta = (float)s;
        st_res<COH>(coef + c, (float)(s * inv_count));
        st_res<COH>(coef + C + c, (float)(q * inv_count));
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_segs_kernel(SumSegDev sg, double inv_count, int C, float *coef) {
    bwd_finalize_segs_channel(sg, inv_count, C, coef, (int)blockIdx.x);
}

// OUT16: dz goes to a SEPARATE bf16 tensor (pixel stride lddz) instead of over z -- the 16-bit configurations' 1x1 input
// gradients read it with one 16-byte load per eight channels (conv_bf16d_kernel<.., XB = true>); the values are the ones that
// kernel would have rounded on load (RNE), so the dgrad's result has the same bits
template <bool OUT16, bool COH = false, bool ADD2 = false, bool Z16 = false>
__device__ __forceinline__ void bwd_apply_body(int bid, const float *z, int ldz, const SegDev &dy, int64_t M, int C,
                                               const float *mean, const float *rstd, const float *shift,
                                               const float *coef, float *dz, float *amax, int drow, int lddz) {
    // (thread = one float4 column group, rows row0, row0 + drow, ...: see bn_apply_relu_kernel)
    const int C4 = C >> 2;
    const int t0 = bid * 256 + threadIdx.x;
    const int row0 = t0 / C4, c = (t0 - row0 * C4) * 4;
    const float4 r4 = *reinterpret_cast<const float4 *>(rstd + c), s4 = *reinterpret_cast<const float4 *>(shift + c);
    const float4 m4 = *reinterpret_cast<const float4 *>(mean + c);
    const float4 k14 = ld_res4<COH>(coef + c), k24 = ld_res4<COH>(coef + C + c);
    const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
    const float a1[4] = {k14.x, k14.y, k14.z, k14.w}, a2[4] = {k24.x, k24.y, k24.z, k24.w};
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < dy.nseg && c >= dy.c_begin[i] && c < dy.c_end[i]) sgi = i;
    const float *const dyp = dy.ptr[sgi] + (c - dy.c_begin[sgi]);
    const int64_t dyld = dy.ld[sgi];
    // ADD2: the gradient is the sum of two tensors of the same layout (ds_segments.ptr2); a segment without one adds nothing
    const float *const dyp2 = (ADD2 && dy.ptr2[sgi]) ? dy.ptr2[sgi] + (c - dy.c_begin[sgi]) : nullptr;
    float am = 0.f;
    constexpr int NR = DS_BN_ROWS;          // rows per pass: every load of the pass before any of its stores
    for (int64_t row = row0; row < M; row += NR * (int64_t)drow) {
        int64_t rws[NR];
        bool ok[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            ok[u] = row + u * (int64_t)drow < M;
            rws[u] = ok[u] ? row + u * (int64_t)drow : row;
        }
        float4 zv[NR], dv[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            if constexpr (Z16) zv[u] = ldz4<__bf16>(reinterpret_cast<const __bf16 *>(z) + rws[u] * ldz + c);
            else zv[u] = ds::ld_stream4(z + rws[u] * ldz + c);
            dv[u] = ds::ld_stream4(dyp + rws[u] * dyld);
        }
        if (ADD2 && dyp2) {
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const float4 e = ds::ld_stream4(dyp2 + rws[u] * dyld);
                dv[u].x += e.x; dv[u].y += e.y; dv[u].z += e.z; dv[u].w += e.w;
            }
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            if (u > 0 && !ok[u]) break;
            const float zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ds::bn_bwd_dz(zz[j], dd[j], rr[j], ss[j], mm[j], a1[j], a2[j]);
            if (OUT16) {
                const f32x4_t ov = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<bf16x4_t *>(reinterpret_cast<__bf16 *>(dz) + rws[u] * lddz + c) = __builtin_convertvector(ov, bf16x4_t);
            } else {
                *reinterpret_cast<float4 *>(dz + rws[u] * ldz + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        if ((threadIdx.x & 63) == 0) ds::atomic_max_nonneg(amax, am);
    }
}

template <bool OUT16, bool ADD2 = false, bool Z16 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *z, int ldz, SegDev dy, int64_t M, int C,
                                                           const float *mean, const float *rstd, const float *shift,
                                                           const float *coef, float *dz, float *amax, int drow, int lddz) {
    bwd_apply_body<OUT16, false, ADD2, Z16>((int)blockIdx.x, z, ldz, dy, M, C, mean, rstd, shift, coef, dz, amax, drow, lddz);
}

// ds_bn_bwd_finalize_apply: the backward finalize (segments, per-segment beta / dbeta) and the apply pass as one launch
// (see bn_finalize_apply_relu_kernel)
struct BwdFinApplyArgs {
    SumSegDev sg; double inv_count; int C; float *coef;
    const float *z; int ldz; SegDev dy; int64_t M;
    const float *mean, *rstd, *shift;
    float *dz, *amax; int drow, lddz;
    unsigned *ticket;
};

template <bool OUT16>
__global__ __launch_bounds__(256) void bn_bwd_finalize_apply_kernel(const BwdFinApplyArgs a) {
    if ((int)blockIdx.x < a.C) {
        bwd_finalize_segs_channel<true>(a.sg, a.inv_count, a.C, a.coef, (int)blockIdx.x);
        fa_publish(a.ticket);
        return;
    }
    fa_wait(a.ticket, (unsigned)a.C);
    bwd_apply_body<OUT16, true>((int)blockIdx.x - a.C, a.z, a.ldz, a.dy, a.M, a.C, a.mean, a.rstd, a.shift, a.coef, a.dz, a.amax, a.drow,
                          a.lddz);
    fa_leave(a.ticket, gridDim.x - (unsigned)a.C);
}

// Launch shape of the fixed-column streaming kernels: gridDim.x * 256 threads = drow rows of C4 column groups each, i.e. the
// grid is a multiple of C4 / gcd(C4, 256); about 16 workgroups per CU (DS_STREAM_BPC; 4 until round 6: 13.28 -> 13.17 ms, bf16 9.60 -> 9.47), fewer for small tensors
// (two rows per thread and pass).
int column_grid(int64_t M, int C4, int *drow) {
    static int bpc = -1;
    if (bpc < 0) {
        const char *e = ds::tune_env("DS_STREAM_BPC");
        bpc = e && atoi(e) > 0 ? atoi(e) : 16;
    }
    int g = C4, b = 256;
    while (b) { const int t = g % b; g = b; b = t; }          // gcd(C4, 256)
    const int q = C4 / g;                                      // grid granule
    int64_t want = (M * C4 + 511) / 512;                       // workgroups if every thread took two rows
    if (want > (int64_t)ds::kCUs * bpc) want = (int64_t)ds::kCUs * bpc;
    int64_t k = (want + q - 1) / q;
    if (k < 1) k = 1;
    const int64_t grid = k * q;
    *drow = (int)(grid * 256 / C4);
    return (int)grid;
}

bool has_second_addend(const ds_segments *s) {
    for (int i = 0; i < s->nseg && i < 4; ++i)
        if (s->ptr2[i]) return true;
    return false;
}

int check_segments(const ds_segments *s, int C, const char *who, bool allow_bf16 = false) {
    DS_REQUIRE(s && s->nseg >= 1 && s->nseg <= 4, "%s: 1..4 segments required", who);
    int covered = 0;
    for (int i = 0; i < s->nseg; ++i) {
        DS_REQUIRE(s->dtype[i] == DS_DTYPE_F32 || (allow_bf16 && s->dtype[i] == DS_DTYPE_BF16),
                   "%s: segment %d has an unsupported storage type", who, i);
        DS_REQUIRE(s->c_begin[i] % 4 == 0 && s->c_end[i] % 4 == 0 && s->ld[i] % 4 == 0 && s->ptr[i] &&
                       (((uintptr_t)s->ptr[i]) & (s->dtype[i] == DS_DTYPE_BF16 ? 7 : 15)) == 0,
                   "%s: segment %d not 4-channel aligned", who, i);
        DS_REQUIRE((((uintptr_t)s->ptr2[i]) & 15) == 0, "%s: segment %d's second addend is not 16-byte aligned", who, i);
        covered += s->c_end[i] - s->c_begin[i];
    }
    DS_REQUIRE(covered == C, "%s: segments cover %d of %d channels", who, covered, C);
    return DS_OK;
}

}  // namespace

extern "C" int ds_bn_finalize(const float *stats, int32_t P, int64_t count, int32_t C, const float *beta,
                              const float *pivot, float eps, float decay, float *mean, float *rstd, float *shift,
                              float *moving_mean, float *moving_var, void *stream) {
    DS_REQUIRE(stats && beta && mean && rstd && shift && P > 0 && count > 0 && C > 0, "ds_bn_finalize: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, P,
                       1.0 / (double)count, C, beta, pivot, eps, decay, mean, rstd, shift, moving_mean, moving_var,
                       (float *)nullptr, (float *)nullptr);
    return ds::check_launch("ds_bn_finalize");
}

extern "C" int ds_bn_finalize_centered(const float *stats, int32_t P, int64_t count, int32_t C, const float *beta,
                                       const float *pivot, float eps, float decay, float *mean, float *rstd, float *shift,
                                       float *moving_mean, float *moving_var, float *mean_c, float *shift_c, void *stream) {
    DS_REQUIRE(stats && beta && mean && rstd && shift && mean_c && shift_c && P > 0 && count > 0 && C > 0,
               "ds_bn_finalize_centered: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, P,
                       1.0 / (double)count, C, beta, pivot, eps, decay, mean, rstd, shift, moving_mean, moving_var, mean_c,
                       shift_c);
    return ds::check_launch("ds_bn_finalize_centered");
}

extern "C" int ds_bn_finalize_multi(const ds_bn_finalize_job *jobs, int32_t njobs, float eps, float decay, void *stream) {
    DS_REQUIRE(jobs && njobs >= 1 && njobs <= 4, "ds_bn_finalize_multi: 1..4 jobs required");
    FinJobsDev jb = {};
    jb.njobs = njobs;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        const ds_bn_finalize_job &j = jobs[i];
        DS_REQUIRE(j.stats && j.beta && j.mean && j.rstd && j.shift && j.P > 0 && j.count > 0 && j.C > 0,
                   "ds_bn_finalize_multi: job %d is malformed", i);
        jb.first[i] = total;
        jb.C[i] = j.C; jb.P[i] = j.P;
        jb.inv_count[i] = 1.0 / (double)j.count;
        jb.stats[i] = j.stats; jb.beta[i] = j.beta; jb.pivot[i] = j.pivot;
        jb.mean[i] = j.mean; jb.rstd[i] = j.rstd; jb.shift[i] = j.shift; jb.mm[i] = j.moving_mean; jb.mv[i] = j.moving_var;
        total += j.C;
    }
    hipLaunchKernelGGL(bn_finalize_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, jb, eps, decay);
    return ds::check_launch("ds_bn_finalize_multi");
}

__global__ __launch_bounds__(256) void bn_infer_prepare_kernel(const float *beta, const float *mm, const float *mv,
                                                               float eps, int C, float *rstd, float *shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        const float r = 1.0f / sqrtf(mv[c] + eps);
        rstd[c] = r;
        shift[c] = beta[c] - mm[c] * r;
    }
}

extern "C" int ds_bn_infer_prepare(const float *beta, const float *moving_mean, const float *moving_var, float eps,
                                   int32_t C, float *rstd, float *shift, void *stream) {
    DS_REQUIRE(beta && moving_mean && moving_var && rstd && shift && C > 0, "ds_bn_infer_prepare: bad argument");
    hipLaunchKernelGGL(bn_infer_prepare_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, beta,
                       moving_mean, moving_var, eps, C, rstd, shift);
    return ds::check_launch("ds_bn_infer_prepare");
}

extern "C" int ds_bn_apply_relu(const float *z, int64_t M, int32_t C, const float *rstd, const float *shift,
                                const ds_segments *dst, void *stream) {
    DS_REQUIRE(z && rstd && shift && M > 0 && C > 0 && C % 4 == 0, "ds_bn_apply_relu: bad argument (C %% 4 != 0?)");
    if (int e = check_segments(dst, C, "ds_bn_apply_relu", true)) return e;
    int drow;
    const int grid = column_grid(M, C / 4, &drow);
    hipLaunchKernelGGL(bn_apply_relu_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, M, C, rstd, shift, to_dev(dst),
                       drow);
    return ds::check_launch("ds_bn_apply_relu");
}

extern "C" int ds_bn_finalize_apply_relu(const float *stats, int32_t P, int64_t count, int32_t C, const float *beta,
                                         const float *pivot, float eps, float decay, float *mean, float *rstd, float *shift,
                                         float *moving_mean, float *moving_var, const float *z, int64_t M,
                                         const ds_segments *dst, uint32_t *ticket, void *stream) {
    DS_REQUIRE(stats && beta && mean && rstd && shift && P > 0 && count > 0 && C > 0 && C % 4 == 0 && z && M > 0 && ticket,
               "ds_bn_finalize_apply_relu: bad argument (C %% 4 != 0?)");
    if (int e = check_segments(dst, C, "ds_bn_finalize_apply_relu", true)) return e;
    FinApplyArgs a;
    a.stats = stats; a.P = P; a.inv_count = 1.0 / (double)count; a.C = C;
    a.beta = beta; a.pivot = pivot; a.eps = eps; a.decay = decay;
    a.mean = mean; a.rstd = rstd; a.shift = shift; a.mm = moving_mean; a.mv = moving_var;
    a.z = z; a.M = M; a.dst = to_dev(dst);
    a.ticket = ticket;
    const int grid = column_grid(M, C / 4, &a.drow);
    hipLaunchKernelGGL(bn_finalize_apply_relu_kernel, dim3(C + grid), dim3(256), 0, (hipStream_t)stream, a);
    return ds::check_launch("ds_bn_finalize_apply_relu");
}

extern "C" int ds_bn_apply_relu_z16(const void *z16, int64_t M, int32_t C, const float *rstd, const float *shift,
                                    const ds_segments *dst, void *stream) {
    DS_REQUIRE(z16 && rstd && shift && M > 0 && C > 0 && C % 4 == 0 && (((uintptr_t)z16) & 7) == 0,
               "ds_bn_apply_relu_z16: bad argument (C %% 4 != 0, or z not 8-byte aligned?)");
    if (int e = check_segments(dst, C, "ds_bn_apply_relu_z16", true)) return e;
    int drow;
    const int grid = column_grid(M, C / 4, &drow);
    hipLaunchKernelGGL(bn_apply_relu_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float *>(z16), M, C, rstd, shift, to_dev(dst), drow);
    return ds::check_launch("ds_bn_apply_relu_z16");
}

extern "C" int ds_bn_bwd_partials(int64_t M, int32_t C) {
    (void)C;
    const int rpb = bwd_rows_per_block(M);
    return (int)((M + rpb - 1) / rpb);
}

extern "C" int ds_bn_bwd_reduce(const void *z, int32_t ldz, int32_t z_dtype, const ds_segments *dy, int64_t M, int32_t C,
                                const float *mean, const float *rstd, const float *shift, float *partials,
                                void *stream) {
    DS_REQUIRE(z && mean && rstd && shift && partials && M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldz >= C &&
                   ldz % 4 == 0 && (((uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)shift) & 15) == 0 &&
                   (z_dtype == DS_DTYPE_F32 || z_dtype == DS_DTYPE_BF16) &&
                   (((uintptr_t)z) & (z_dtype == DS_DTYPE_BF16 ? 7 : 15)) == 0,
               "ds_bn_bwd_reduce: bad argument (need C %% 4 == 0, C <= 1024, ldz %% 4 == 0, aligned pointers, z_dtype f32 / bf16)");
    if (int e = check_segments(dy, C, "ds_bn_bwd_reduce")) return e;
    DS_REQUIRE(!has_second_addend(dy), "ds_bn_bwd_reduce: a gradient in two addends takes its sums from the producers (ds_bn_sum_segments.P2)");
    const int C4 = C / 4;
    const int RG = 256 / C4 > 0 ? 256 / C4 : 1;
    const size_t shmem = (size_t)RG * C4 * 8 * sizeof(float);
    if (z_dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<__bf16>, dim3(ds_bn_bwd_partials(M, C)), dim3(256), shmem, (hipStream_t)stream,
                           (const __bf16 *)z, ldz, to_dev(dy), M, C, mean, rstd, shift, partials, bwd_rows_per_block(M));
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(ds_bn_bwd_partials(M, C)), dim3(256), shmem, (hipStream_t)stream,
                           (const float *)z, ldz, to_dev(dy), M, C, mean, rstd, shift, partials, bwd_rows_per_block(M));
    return ds::check_launch("ds_bn_bwd_reduce");
}

extern "C" int ds_bn_bwd_finalize(const float *partials, int32_t P, int64_t M, int32_t C, float *dbeta, float *coef,
                                  void *stream) {
    DS_REQUIRE(partials && dbeta && P > 0 && M > 0 && C > 0, "ds_bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, P,
                       1.0 / (double)M, C, dbeta, coef);
    return ds::check_launch("ds_bn_bwd_finalize");
}

namespace {
int build_sum_segs(const char *who, const ds_bn_sum_segments *sg, int64_t M, int32_t C, const float *beta, float *dbeta,
                   const float *const *beta_v, float *const *dbeta_v, float *coef, SumSegDev &d) {
    DS_REQUIRE(sg && coef && M > 0 && C > 0 && sg->nseg >= 1 && sg->nseg <= 4, "%s: bad argument", who);
    d.nseg = sg->nseg;
    int covered = 0;
    for (int i = 0; i < 4; ++i) {
        d.c_begin[i] = sg->c_begin[i]; d.c_end[i] = sg->c_end[i]; d.P[i] = sg->P[i]; d.kind[i] = sg->kind[i];
        d.s[i] = sg->s[i]; d.q[i] = sg->q[i];
        d.beta[i] = nullptr; d.dbeta[i] = nullptr;
        d.P2[i] = i < sg->nseg ? sg->P2[i] : 0; d.s2[i] = sg->s2[i]; d.q2[i] = sg->q2[i];
        if (i < sg->nseg) {
            DS_REQUIRE(sg->P2[i] >= 0 && (sg->P2[i] == 0 || (sg->s2[i] && sg->q2[i])), "%s: segment %d has a malformed second source", who, i);
            DS_REQUIRE(sg->s[i] && sg->q[i] && sg->P[i] > 0 && sg->c_end[i] > sg->c_begin[i] && (sg->kind[i] == 0 || sg->kind[i] == 1),
                       "%s: segment %d is malformed", who, i);
            d.beta[i] = beta_v ? beta_v[i] : (beta ? beta + sg->c_begin[i] : nullptr);
            d.dbeta[i] = dbeta_v ? dbeta_v[i] : (dbeta ? dbeta + sg->c_begin[i] : nullptr);
            DS_REQUIRE(sg->kind[i] == 0 || d.beta[i], "%s: DS_EPI_BNSUMS partials need beta", who);
            covered += sg->c_end[i] - sg->c_begin[i];
        }
    }
    DS_REQUIRE(covered == C, "%s: segments cover %d of %d channels", who, covered, C);
    return DS_OK;
}

int launch_finalize_segs(const char *who, const ds_bn_sum_segments *sg, int64_t M, int32_t C, const float *beta, float *dbeta,
                         const float *const *beta_v, float *const *dbeta_v, float *coef, void *stream) {
    SumSegDev d;
    if (int e = build_sum_segs(who, sg, M, C, beta, dbeta, beta_v, dbeta_v, coef, d)) return e;
    hipLaunchKernelGGL(bn_bwd_finalize_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, d, 1.0 / (double)M, C, coef);
    return ds::check_launch(who);
}
}  // namespace

extern "C" int ds_bn_bwd_finalize_segs(const ds_bn_sum_segments *sg, int64_t M, int32_t C, const float *beta,
                                       float *dbeta, float *coef, void *stream) {
    return launch_finalize_segs("ds_bn_bwd_finalize_segs", sg, M, C, beta, dbeta, nullptr, nullptr, coef, stream);
}

extern "C" int ds_bn_bwd_finalize_multi(const ds_bn_sum_segments *sg, int64_t M, int32_t C, const float *const *beta,
                                        float *const *dbeta, float *coef, void *stream) {
    DS_REQUIRE(beta, "ds_bn_bwd_finalize_multi: the segments' beta vectors are required");
    return launch_finalize_segs("ds_bn_bwd_finalize_multi", sg, M, C, nullptr, nullptr, beta, dbeta, coef, stream);
}

extern "C" int ds_bn_bwd_finalize_apply(const ds_bn_sum_segments *sg, const float *beta, float *dbeta, const float *const *beta_v,
                                        float *const *dbeta_v, float *coef, const float *z, int32_t ldz, const ds_segments *dy,
                                        int64_t M, int32_t C, const float *mean, const float *rstd, const float *shift, void *dz,
                                        int32_t dz_dtype, int32_t lddz, float *amax, uint32_t *ticket, void *stream) {
    const bool out16 = dz_dtype == DS_DTYPE_BF16;
    DS_REQUIRE(z && mean && rstd && shift && coef && dz && ticket && M > 0 && C > 0 && C % 4 == 0 && ldz >= C && ldz % 4 == 0 &&
                   (((uintptr_t)z) & 15) == 0 && (dz_dtype == DS_DTYPE_F32 || out16) &&
                   (((uintptr_t)dz) & (out16 ? 7 : 15)) == 0 && (!out16 || (lddz >= C && lddz % 4 == 0)),
               "ds_bn_bwd_finalize_apply: bad argument (see ds_bn_bwd_apply / ds_bn_bwd_apply_bf16)");
    if (int e = check_segments(dy, C, "ds_bn_bwd_finalize_apply")) return e;
    DS_REQUIRE(!has_second_addend(dy), "ds_bn_bwd_finalize_apply: no second addend (use ds_bn_bwd_finalize_segs + ds_bn_bwd_apply)");
    BwdFinApplyArgs a;
    if (int e = build_sum_segs("ds_bn_bwd_finalize_apply", sg, M, C, beta, dbeta, beta_v, dbeta_v, coef, a.sg)) return e;
    a.inv_count = 1.0 / (double)M; a.C = C; a.coef = coef;
    a.z = z; a.ldz = ldz; a.dy = to_dev(dy); a.M = M;
    a.mean = mean; a.rstd = rstd; a.shift = shift;
    a.dz = reinterpret_cast<float *>(dz); a.amax = amax; a.lddz = out16 ? lddz : ldz;
    a.ticket = ticket;
    const int grid = column_grid(M, C / 4, &a.drow);
    if (out16) hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<true>, dim3(C + grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<false>, dim3(C + grid), dim3(256), 0, (hipStream_t)stream, a);
    return ds::check_launch("ds_bn_bwd_finalize_apply");
}

extern "C" int ds_bn_bwd_apply(const float *z, int32_t ldz, const ds_segments *dy, int64_t M, int32_t C, const float *mean,
                               const float *rstd, const float *shift, const float *coef, float *dz, float *amax,
                               void *stream) {
    DS_REQUIRE(z && mean && rstd && shift && coef && dz && M > 0 && C > 0 && C % 4 == 0 && ldz >= C && ldz % 4 == 0 &&
                   ((((uintptr_t)z) | ((uintptr_t)dz)) & 15) == 0,
               "ds_bn_bwd_apply: bad argument (need C %% 4 == 0, ldz >= C, ldz %% 4 == 0, 16-byte aligned z / dz)");
    if (int e = check_segments(dy, C, "ds_bn_bwd_apply")) return e;
    int drow;
    const int grid = column_grid(M, C / 4, &drow);
    if (has_second_addend(dy))
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, z, ldz, to_dev(dy), M, C, mean,
                           rstd, shift, coef, dz, amax, drow, ldz);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, ldz, to_dev(dy), M, C, mean, rstd,
                           shift, coef, dz, amax, drow, ldz);
    return ds::check_launch("ds_bn_bwd_apply");
}

extern "C" int ds_bn_bwd_apply_z16(const void *z16, int32_t ldz, const ds_segments *dy, int64_t M, int32_t C, const float *mean,
                                   const float *rstd, const float *shift, const float *coef, void *dz16, int32_t lddz,
                                   float *amax, void *stream) {
    DS_REQUIRE(z16 && mean && rstd && shift && coef && dz16 && M > 0 && C > 0 && C % 4 == 0 && ldz >= C && ldz % 4 == 0 &&
                   lddz >= C && lddz % 4 == 0 && (((uintptr_t)z16) & 7) == 0 && (((uintptr_t)dz16) & 7) == 0 && z16 != dz16,
               "ds_bn_bwd_apply_z16: bad argument (need C %% 4 == 0, ldz / lddz >= C and %% 4 == 0, 8-byte aligned z / dz, dz != z)");
    if (int e = check_segments(dy, C, "ds_bn_bwd_apply_z16")) return e;
    int drow;
    const int grid = column_grid(M, C / 4, &drow);
    const float *zf = reinterpret_cast<const float *>(z16);
    if (has_second_addend(dy))
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, zf, ldz, to_dev(dy), M, C,
                           mean, rstd, shift, coef, reinterpret_cast<float *>(dz16), amax, drow, lddz);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, zf, ldz, to_dev(dy), M, C,
                           mean, rstd, shift, coef, reinterpret_cast<float *>(dz16), amax, drow, lddz);
    return ds::check_launch("ds_bn_bwd_apply_z16");
}

extern "C" int ds_bn_bwd_apply_bf16(const float *z, int32_t ldz, const ds_segments *dy, int64_t M, int32_t C, const float *mean,
                                    const float *rstd, const float *shift, const float *coef, void *dz16, int32_t lddz,
                                    float *amax, void *stream) {
    DS_REQUIRE(z && mean && rstd && shift && coef && dz16 && M > 0 && C > 0 && C % 4 == 0 && ldz >= C && ldz % 4 == 0 &&
                   lddz >= C && lddz % 4 == 0 && (((uintptr_t)z) & 15) == 0 && (((uintptr_t)dz16) & 7) == 0,
               "ds_bn_bwd_apply_bf16: bad argument (need C %% 4 == 0, ldz / lddz >= C and %% 4 == 0, aligned z / dz)");
    if (int e = check_segments(dy, C, "ds_bn_bwd_apply_bf16")) return e;
    int drow;
    const int grid = column_grid(M, C / 4, &drow);
    if (has_second_addend(dy))
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, z, ldz, to_dev(dy), M, C, mean,
                           rstd, shift, coef, reinterpret_cast<float *>(dz16), amax, drow, lddz);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, ldz, to_dev(dy), M, C, mean,
                           rstd, shift, coef, reinterpret_cast<float *>(dz16), amax, drow, lddz);
    return ds::check_launch("ds_bn_bwd_apply_bf16");
}
