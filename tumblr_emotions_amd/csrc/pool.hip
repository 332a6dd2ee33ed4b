// Pooling kernels (HBM-bound, 16 B per lane, NHWC).
//   max pool : slim.max_pool2d 3x3/2 SAME, 3x3/1 SAME, 2x2/2  image_model/inception_v1.py:67,79,94,118,208
//              TF SAME geometry: pad_before = pad_total/2, padded cells never win (A1).
//   MaxPoolGrad as a gather over the recorded arg-max (no atomics, deterministic).
//   avg pool 7x7 VALID + dropout(keep 0.8)   image_model/inception_v1.py:299-301
#include <type_traits>
#include "ds_common.h"

namespace {

// activation storage: fp32 or bf16 (16-bit activation storage of the bf16 / fp8 configurations); four channels at a time
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ float4 ld4(const T *p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <>
__device__ __forceinline__ float4 ld4<__bf16>(const __bf16 *p) {
    const f32x4v v = __builtin_convertvector(*reinterpret_cast<const bf16x4 *>(p), f32x4v);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(float *p, float a, float b, float c, float d) {
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void st4(__bf16 *p, float a, float b, float c, float d) {
    const f32x4v v = {a, b, c, d};
    *reinterpret_cast<bf16x4 *>(p) = __builtin_convertvector(v, bf16x4);
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const TI *x, TO *y, uint8_t *am, int N, int H, int W,
                                                          int C, int k, int stride, int pad_t, int pad_l, int OH,
                                                          int OW, const float *rstd = nullptr, const float *shift = nullptr) {
    // rstd / shift (nullable): x holds pre-BatchNorm values, y = relu(rstd * max(x) + shift) = max(relu(bn(x))) (rstd > 0)
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        int64_t p = i / C4;
        const int ow = (int)(p % OW);
        p /= OW;
        const int oh = (int)(p % OH);
        const int n = (int)(p / OH);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
        bool any = false;
        for (int kh = 0; kh < k; ++kh) {
            const int ih = oh * stride - pad_t + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = ow * stride - pad_l + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = ld4<TI>(x + (((int64_t)n * H + ih) * W + iw) * C + c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
                const int t = kh * k + kw;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!any || vv[j] > best[j]) {      // strict '>' keeps the first maximum (row-major)
                        best[j] = vv[j];
                        arg[j] = t;
                    }
                any = true;
            }
        }
        const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
        if (rstd) {
            const float4 r = *reinterpret_cast<const float4 *>(rstd + c), s = *reinterpret_cast<const float4 *>(shift + c);
            best[0] = fmaxf(best[0] * r.x + s.x, 0.f);
            best[1] = fmaxf(best[1] * r.y + s.y, 0.f);
            best[2] = fmaxf(best[2] * r.z + s.z, 0.f);
            best[3] = fmaxf(best[3] * r.w + s.w, 0.f);
        }
        st4(y + o, best[0], best[1], best[2], best[3]);
        if (am) *reinterpret_cast<uchar4 *>(am + o) = make_uchar4(arg[0], arg[1], arg[2], arg[3]);
    }
}

// KK/SS > 0 fix the window / stride at compile time (the divisions below become shifts)
template <int KK, int SS>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *dy, const uint8_t *am, float *dx,
                                                          int accumulate, int N, int H, int W, int C, int k_rt,
                                                          int stride_rt, int pad_t, int pad_l, int OH, int OW) {
    const int k = KK > 0 ? KK : k_rt, stride = SS > 0 ? SS : stride_rt;
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        int64_t p = i / C4;
        const int iw = (int)(p % W);
        p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        // windows (oh, ow) that contain (ih, iw): kh = ih + pad_t - oh*stride in [0, k)
        for (int kh = 0; kh < k; ++kh) {
            const int t_h = ih + pad_t - kh;
            if (t_h < 0 || t_h % stride) continue;
            const int oh = t_h / stride;
            if (oh >= OH) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int t_w = iw + pad_l - kw;
                if (t_w < 0 || t_w % stride) continue;
                const int ow = t_w / stride;
                if (ow >= OW) continue;
                const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
                const uchar4 a = *reinterpret_cast<const uchar4 *>(am + o);
                const float4 d = *reinterpret_cast<const float4 *>(dy + o);
                const int t = kh * k + kw;
                if (a.x == t) g[0] += d.x;
                if (a.y == t) g[1] += d.y;
                if (a.z == t) g[2] += d.z;
                if (a.w == t) g[3] += d.w;
            }
        }
        float4 *q = reinterpret_cast<float4 *>(dx + (((int64_t)n * H + ih) * W + iw) * C + c);
        float4 out = make_float4(g[0], g[1], g[2], g[3]);
        if (accumulate) {
            const float4 e = *q;
            out.x += e.x; out.y += e.y; out.z += e.z; out.w += e.w;
        }
        *q = out;
    }
}

// ---- 3x3 pools with a rolling window -----------------------------------------------------------
// A thread owns one (image, output column, 4-channel group) and walks down the rows keeping the
// per-row maxima of the last rows in registers: 3*stride neighbour loads per output instead of 9
// (the 3x3/1 pools of the nine Inception blocks read every input 9x otherwise).  The arg-max is
// separable too: first maximum inside a row (kw), then first row holding the maximum (kh), which
// is the row-major first maximum the reference semantics ask for.
struct RowMax {
    float v[4];
    int k[4];
};

template <typename TI>
__device__ __forceinline__ RowMax row_max3(const TI *x, int64_t row_base, int iw0, int W, int C, int c, bool row_ok) {
    RowMax r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.v[j] = -INFINITY;
        r.k[j] = 0;
    }
    if (!row_ok) return r;
    bool any = false;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int iw = iw0 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 v = ld4<TI>(x + (row_base + iw) * C + c);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!any || vv[j] > r.v[j]) {
                r.v[j] = vv[j];
                r.k[j] = kw;
            }
        any = true;
    }
    return r;
}

// The same in two halves, so that a row can be REQUESTED iterations before it is reduced: each thread's walk down the
// rows is one dependent chain (load -> max -> store, and a load behind a store waits for it: one in-order memory counter),
// i.e. the kernel's time was rows x memory latency -- 4.9 TB/s.  With the next rows' loads issued two rows ahead, before
// the stores of the current one, twice the bytes are in flight per thread.
#ifndef DS_POOL_PF16
#define DS_POOL_PF16 2          // rows in flight per thread of the 3x3/1 pools on 16-bit input (4 and 6 measured slower: bf16 step 9.52 -> 9.58 / 9.63)
#endif
struct RawRow3 {
    float4 v[3];
    bool ok[3];
};

template <typename TI>
__device__ __forceinline__ RawRow3 load_row3(const TI *x, int64_t row_base, int iw0, int W, int C, int c, bool row_ok) {
    RawRow3 q;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int iw = iw0 + kw;
        q.ok[kw] = row_ok && (unsigned)iw < (unsigned)W;
        q.v[kw] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q.ok[kw]) q.v[kw] = ld4<TI>(x + (row_base + iw) * C + c);
    }
    return q;
}

__device__ __forceinline__ RowMax reduce_row3(const RawRow3 &q) {      // = row_max3 on the loaded values
    RowMax r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.v[j] = -INFINITY;
        r.k[j] = 0;
    }
    bool any = false;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        if (!q.ok[kw]) continue;
        const float vv[4] = {q.v[kw].x, q.v[kw].y, q.v[kw].z, q.v[kw].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!any || vv[j] > r.v[j]) {
                r.v[j] = vv[j];
                r.k[j] = kw;
            }
        any = true;
    }
    return r;
}

// rstd/shift != nullptr: x is the PRE-BatchNorm conv output z and the pooled value is stored as
// relu(rstd*max(z) + shift).  With rstd > 0 the affine map and the ReLU are monotone, so this equals the max of
// relu(bn(z)) over the window (and ties only appear among positions whose ReLU gradient is zero anyway): the
// full-resolution activation of a conv that only feeds a pool is never written or re-read.
template <int STRIDE, typename TI, typename TO>
__global__ __launch_bounds__(256) void maxpool3_fwd_rolling(const TI *x, TO *y, uint8_t *am, int N, int H, int W,
                                                            int C, int pad_t, int pad_l, int OH, int OW,
                                                            const float *rstd, const float *shift, float *amax) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OW * C4;
    float ymax = 0.f;      // max of the (non-negative) outputs of the fused BatchNorm + ReLU + pool, when asked for
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const int ow = (int)((i / C4) % OW);
        const int n = (int)(i / ((int64_t)C4 * OW));
        const int iw0 = ow * STRIDE - pad_l;
        const int64_t img = (int64_t)n * H;
        RowMax r0, r1, r2;      // input rows ih0, ih0+1, ih0+2 of the current output row
        int ih0 = -pad_t;
        auto req = [&](int ih) { return load_row3(x, (img + ih) * W, iw0, W, C, c, (unsigned)ih < (unsigned)H); };
        r0 = row_max3(x, (img + ih0) * W, iw0, W, C, c, (unsigned)ih0 < (unsigned)H);
        r1 = row_max3(x, (img + ih0 + 1) * W, iw0, W, C, c, (unsigned)(ih0 + 1) < (unsigned)H);
        // STRIDE 1: rows requested ahead of their use -- q[0] = row ih0 + 2 (this output row's new row), q[1] = ih0 + 3 (the next
        // one's), ...  D rows in flight: two.  (16-bit input, whose rows are half the bytes and whose Branch_3 pools run at
        // 2.9 TB/s against 5.5 for the fp32 ones: four / six rows in flight measured SLOWER, DS_POOL_PF16 -- not the bytes in flight)
        constexpr int D = STRIDE == 1 ? (sizeof(TI) == 2 ? DS_POOL_PF16 : 2) : 1;
        RawRow3 q[D];
        if (STRIDE == 1) {
#pragma unroll
            for (int u = 0; u < D; ++u) q[u] = req(ih0 + 2 + u);
        }
        for (int oh = 0; oh < OH; ++oh) {
            RawRow3 na;
            if (STRIDE == 1) {
                na = req(ih0 + 2 + D);     // the load of a LATER output row goes out before this one's stores
                r2 = reduce_row3(q[0]);
            } else {                       // (3x3/2: measured mixed -- 112x112 and 28x28 gain, 56x56 loses: kept as it was)
                r2 = row_max3(x, (img + ih0 + 2) * W, iw0, W, C, c, (unsigned)(ih0 + 2) < (unsigned)H);
            }
            float best[4];
            int arg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // strict '>' keeps the first (lowest kh) maximum
                best[j] = r0.v[j];
                arg[j] = r0.k[j];
                if (r1.v[j] > best[j]) { best[j] = r1.v[j]; arg[j] = 3 + r1.k[j]; }
                if (r2.v[j] > best[j]) { best[j] = r2.v[j]; arg[j] = 6 + r2.k[j]; }
            }
            const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
            if (rstd) {
                const float4 r = *reinterpret_cast<const float4 *>(rstd + c);
                const float4 s = *reinterpret_cast<const float4 *>(shift + c);
                best[0] = fmaxf(best[0] * r.x + s.x, 0.f);
                best[1] = fmaxf(best[1] * r.y + s.y, 0.f);
                best[2] = fmaxf(best[2] * r.z + s.z, 0.f);
                best[3] = fmaxf(best[3] * r.w + s.w, 0.f);
                ymax = fmaxf(ymax, fmaxf(fmaxf(best[0], best[1]), fmaxf(best[2], best[3])));
            }
            st4(y + o, best[0], best[1], best[2], best[3]);
            if (am) *reinterpret_cast<uchar4 *>(am + o) = make_uchar4(arg[0], arg[1], arg[2], arg[3]);
            if (STRIDE == 1) {
                r0 = r1;
                r1 = r2;
#pragma unroll
                for (int u = 0; u + 1 < D; ++u) q[u] = q[u + 1];      // row ih0 + 3 becomes the next output row's new row ...
                q[D - 1] = na;                                        // ... and the row requested above the last in the queue
            } else {
                r0 = r2;
                r1 = row_max3(x, (img + ih0 + 3) * W, iw0, W, C, c, (unsigned)(ih0 + 3) < (unsigned)H);
            }
            ih0 += STRIDE;
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
        if ((threadIdx.x & 63) == 0) ds::atomic_max_nonneg(amax, ymax);
    }
}

// MaxPoolGrad of the 3x3 stride-1 SAME pool: input pixel (ih, iw) lies in the windows (oh, ow) with
// oh in [ih-1, ih+1], ow in [iw-1, iw+1]; window (oh, ow) names it iff argmax == (ih-oh+1)*3 + (iw-ow+1).
struct WinRow {
    float d[3][4];
    unsigned a[3];      // packed uchar4 arg-max of the three windows of one output row
};

template <typename TD>          // TD: storage of the pool's output gradient (float; __bf16: a dgrad wrote it rounded, ds_conv_desc.z_dtype)
__device__ __forceinline__ WinRow load_win_row(const TD *dy, const uint8_t *am, int64_t row_base, int iw, int W, int C,
                                               int c, bool row_ok) {
    WinRow r;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ow = iw - 1 + q;
        const bool ok = row_ok && (unsigned)ow < (unsigned)W;
        r.a[q] = 0xFFFFFFFFu;      // never matches a window index
#pragma unroll
        for (int j = 0; j < 4; ++j) r.d[q][j] = 0.f;
        if (ok) {
            const int64_t o = (row_base + ow) * C + c;
            r.a[q] = *reinterpret_cast<const unsigned *>(am + o);
            const float4 v = ld4<TD>(dy + o);
            r.d[q][0] = v.x; r.d[q][1] = v.y; r.d[q][2] = v.z; r.d[q][3] = v.w;
        }
    }
    return r;
}

__device__ __forceinline__ void win_row_grad(const WinRow &w, int p, float g[4]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {      // output col ow = iw-1+q  -> kw = 2-q
        const unsigned t = (unsigned)((2 - p) * 3 + (2 - q));
        const unsigned a = w.a[q];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((a >> (8 * j)) & 0xFFu) == t) g[j] += w.d[q][j];
    }
}

template <typename TD = float>
__global__ __launch_bounds__(256) void maxpool3s1_bwd_rolling(const TD *dy, const uint8_t *am, float *dx,
                                                              int accumulate, int N, int H, int W, int C) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const int iw = (int)((i / C4) % W);
        const int n = (int)(i / ((int64_t)C4 * W));
        const int64_t img = (int64_t)n * H;
        WinRow w0, w1, w2;      // output rows ih-1, ih, ih+1
        w0 = load_win_row(dy, am, (img - 1) * W, iw, W, C, c, false);
        w1 = load_win_row(dy, am, img * W, iw, W, C, c, true);
        for (int ih = 0; ih < H; ++ih) {
            w2 = load_win_row(dy, am, (img + ih + 1) * W, iw, W, C, c, ih + 1 < H);
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            win_row_grad(w0, 0, g);                // output row oh = ih-1+p  -> kh = 2-p
            win_row_grad(w1, 1, g);
            win_row_grad(w2, 2, g);
            float4 *dst = reinterpret_cast<float4 *>(dx + ((img + ih) * W + iw) * C + c);
            float4 out = make_float4(g[0], g[1], g[2], g[3]);
            if (accumulate) {
                const float4 e = *dst;
                out.x += e.x; out.y += e.y; out.z += e.z; out.w += e.w;
            }
            *dst = out;
            w0 = w1;
            w1 = w2;
        }
    }
}

// The same walk when the pool path is the LAST addend of an Inception block's input gradient (16-bit configurations: the fused
// 1x1 dgrad writes first, Branch_3's pool gradient is added here): the launch then holds the complete gradient of the previous
// block's concat output y, so it also emits that block's BatchNorm-backward sums -- per channel sum g and sum g*y with
// g = dx (y > 0), what DS_EPI_BNSUMS leaves behind (ds_bn_bwd_finalize_segs kind 1) -- and the previous block's four
// ds_bn_bwd_reduce passes over z and dy (8 B/element, two of them on the critical chain) disappear for 2 B/element of y here.
// Layout as bn_bwd_reduce_kernel: thread = (channel quad cg, unit group rg) keeps its channels for the whole launch, a unit =
// one image column (n, iw) walked down its H rows.  A workgroup owns a CHUNK of CW channel quads (CW divides C / 4: pool_sums_shape
// picks the divisor that fills the 256 threads -- 528 channels: 6 chunks of 22 quads x 11 unit groups instead of 132 of 256 threads,
// 832: 13 chunks of 16 x 16 instead of 208) and the units [ub * upb, (ub + 1) * upb), ub = blockIdx.x / chunks; partials [2][C][P] with
// P = gridDim.x / chunks, summed per thread in unit order and over the unit groups in a fixed order (deterministic).
template <typename TY, typename TD = float>
__global__ __launch_bounds__(256) void maxpool3s1_bwd_sums_kernel(const TD *dy, const uint8_t *am, float *dx, int accumulate,
                                                                  const TY *y, int N, int H, int W, int C, float *partials,
                                                                  int upb, int CW, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [RG][CW][8]
    const int RG = 256 / CW;
    const int tid = threadIdx.x;
    const int cg = tid % CW, rg = tid / CW;
    const bool active = tid < RG * CW;
    const int units = N * W;
    const int ub = blockIdx.x / chunks, q0 = (blockIdx.x - ub * chunks) * CW;      // first channel quad of the chunk
    const int u0 = ub * upb;
    const int u1 = u0 + upb < units ? u0 + upb : units;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sy[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int c = (q0 + cg) * 4;
        for (int u = u0 + rg; u < u1; u += RG) {
            const int n = u / W, iw = u - n * W;
            const int64_t img = (int64_t)n * H;
            WinRow w0, w1, w2;      // output rows ih-1, ih, ih+1
            w0 = load_win_row(dy, am, (img - 1) * W, iw, W, C, c, false);
            w1 = load_win_row(dy, am, img * W, iw, W, C, c, true);
            for (int ih = 0; ih < H; ++ih) {
                w2 = load_win_row(dy, am, (img + ih + 1) * W, iw, W, C, c, ih + 1 < H);
                const int64_t o = ((img + ih) * W + iw) * C + c;
                float yy[4];
                if constexpr (std::is_same<TY, float>::value) {
                    const float4 t = *reinterpret_cast<const float4 *>(y + o);
                    yy[0] = t.x; yy[1] = t.y; yy[2] = t.z; yy[3] = t.w;
                } else {
                    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    const f32x4_t t = __builtin_convertvector(*reinterpret_cast<const bf16x4_t *>(y + o), f32x4_t);
                    yy[0] = t[0]; yy[1] = t[1]; yy[2] = t[2]; yy[3] = t[3];
                }
                float g[4] = {0.f, 0.f, 0.f, 0.f};
                win_row_grad(w0, 0, g);
                win_row_grad(w1, 1, g);
                win_row_grad(w2, 2, g);
                float4 *dst = reinterpret_cast<float4 *>(dx + o);
                float4 out = make_float4(g[0], g[1], g[2], g[3]);
                if (accumulate) {
                    const float4 e = *dst;
                    out.x += e.x; out.y += e.y; out.z += e.z; out.w += e.w;
                }
                *dst = out;
                const float oo[4] = {out.x, out.y, out.z, out.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gm = yy[j] > 0.f ? oo[j] : 0.f;
                    sg[j] += gm;
                    sy[j] += gm * yy[j];
                }
                w0 = w1;
                w1 = w2;
            }
        }
        float *o = sh + ((int64_t)rg * CW + cg) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = sg[j];
            o[4 + j] = sy[j];
        }
    }
    __syncthreads();
    if (tid < CW) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += sh[((int64_t)g * CW + tid) * 8 + j];
        const int P = gridDim.x / chunks;               // partials laid out [2][C][P]
        const int ch = (q0 + tid) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            partials[(int64_t)(ch + j) * P + ub] = a[j];
            partials[((int64_t)C + ch + j) * P + ub] = a[4 + j];
        }
    }
}

// The launch shape.  CW = channel quads per workgroup: the divisor of C / 4 in [16, 64] (256-byte to 1 KB runs of dy / dx per
// pixel) that puts most of the 256 threads to work, the larger on ties; without one, all quads in one chunk.  Narrow chunks mean
// MORE unit groups per workgroup, so fewer unit blocks and fewer of the scattered 4-byte partial stores (2 C per block), which is
// what the launch paid for beside idle lanes (B = 256, 14x14x512: 92 -> 77 us with 64 quads x 4 groups instead of 128 x 2;
// 14x14x528: 128 -> 90 with 22 x 11 instead of 132 x 1; 7x7x832: 65 -> 32 with 16 x 16; r06_notes).  DS_POOL_SUMS_CHUNKS=0
// (tuning build): one chunk.  upb = units (image columns) per workgroup: one per unit group (every thread walks ONE column, the
// parallelism of maxpool3s1_bwd_rolling -- with four workgroups per CU and several columns per thread the launch was
// latency-bound: 119 us against 79) unless that makes more than 2048 unit blocks (partials).
struct PoolSumsShape { int CW, chunks, RG, upb, P; };
inline PoolSumsShape pool_sums_shape(int N, int W, int C) {
    const int units = N * W, C4 = C / 4;
    int cw = 0, act = 0;
    const char *e = ds::tune_env("DS_POOL_SUMS_CHUNKS");
    if (!(e && e[0] == '0'))
        for (int d = C4 < 64 ? C4 : 64; d >= 16; --d) {
            if (C4 % d) continue;
            const int a = d * (256 / d);
            if (a > act) { cw = d; act = a; }
        }
    if (cw == 0) cw = C4;      // C <= 1024: at most 256 quads
    if (const char *f = ds::tune_env("DS_POOL_SUMS_CW")) {      // tuning build: the largest divisor of C / 4 not above the value
        const int lim = atoi(f);
        for (int d = lim < C4 ? lim : C4; d >= 1; --d)
            if (C4 % d == 0) { cw = d; break; }
    }
    PoolSumsShape s;
    s.CW = cw;
    s.chunks = C4 / cw;
    s.RG = 256 / cw;
    const int cap = (units + 2047) / 2048;
    s.upb = s.RG > cap ? s.RG : cap;
    s.P = (units + s.upb - 1) / s.upb;
    return s;
}

// MaxPoolGrad of the 3x3 stride-2 pools: a thread owns the 2x2 input patch
//   rows {2p-pt+1, 2p-pt+2} x cols {2q-pl+1, 2q-pl+2},   p in [-1, OH-1], q in [-1, OW-1]
// which only the four windows (p..p+1, q..q+1) can name: one (arg-max, dy) load per input pixel instead of
// four.  Row 2p-pt+1 is the middle row (kh=1) of window p only; row 2p-pt+2 is kh=2 of window p and kh=0
// of window p+1; same for columns.
__global__ __launch_bounds__(256) void maxpool3s2_bwd_patch(const float *dy, const uint8_t *am, float *dx,
                                                            int accumulate, int N, int H, int W, int C, int pad_t,
                                                            int pad_l, int OH, int OW) {
    const int C4 = C >> 2;
    const int PH = OH + 1, PW = OW + 1;
    const int64_t total = (int64_t)N * PH * PW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int q = (int)(r % PW) - 1;
        r /= PW;
        const int p = (int)(r % PH) - 1;
        const int n = (int)(r / PH);
        float d[2][2][4];
        unsigned a[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int oh = p + u, ow = q + v;
                a[u][v] = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < 4; ++j) d[u][v][j] = 0.f;
                if ((unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW) {
                    const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
                    a[u][v] = *reinterpret_cast<const unsigned *>(am + o);
                    const float4 t = *reinterpret_cast<const float4 *>(dy + o);
                    d[u][v][0] = t.x; d[u][v][1] = t.y; d[u][v][2] = t.z; d[u][v][3] = t.w;
                }
            }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int ih = 2 * p - pad_t + 1 + y;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int iw = 2 * q - pad_l + 1 + x;
                if ((unsigned)iw >= (unsigned)W) continue;
                float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u <= y; ++u)          // window row p+u sees this input row as kh = (y+1) - 2u
#pragma unroll
                    for (int v = 0; v <= x; ++v) {    // window col q+v sees this input col as kw = (x+1) - 2v
                        const unsigned t = (unsigned)(((y + 1) - 2 * u) * 3 + ((x + 1) - 2 * v));
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (((a[u][v] >> (8 * j)) & 0xFFu) == t) g[j] += d[u][v][j];
                    }
                float4 *dst = reinterpret_cast<float4 *>(dx + (((int64_t)n * H + ih) * W + iw) * C + c);
                float4 out = make_float4(g[0], g[1], g[2], g[3]);
                if (accumulate) {
                    const float4 e = *dst;
                    out.x += e.x; out.y += e.y; out.z += e.z; out.w += e.w;
                }
                *dst = out;
            }
        }
    }
}

// counter-based uniform in [0,1): splitmix64 finaliser over (seed, element index)
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void avgpool_dropout_fwd_kernel(const float *x, int N, int HW, int C, float keep,
                                                                  uint64_t seed, const uint64_t *seed_dev,
                                                                  const float *mask_in, float *mask_out, float *out) {
    const int n = blockIdx.x;
    if (seed_dev) seed += seed_dev[0];        // per-step seed kept on device (hipGraph replay safe)
    const float inv = 1.f / (float)HW;
    for (int c = threadIdx.x * 4; c < C; c += 256 * 4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float *px = x + (int64_t)n * HW * C + c;
        for (int p = 0; p < HW; ++p) {
            const float4 v = *reinterpret_cast<const float4 *>(px + (int64_t)p * C);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t e = (int64_t)n * C + c + j;
            if (keep >= 1.f) m[j] = 1.f;
            else if (mask_in) m[j] = mask_in[e];
            else m[j] = hash_uniform(seed, (uint64_t)e) < keep ? 1.f : 0.f;
            s[j] = s[j] * inv * (keep >= 1.f ? 1.f : m[j] / keep);
        }
        *reinterpret_cast<float4 *>(out + (int64_t)n * C + c) = make_float4(s[0], s[1], s[2], s[3]);
        if (mask_out) *reinterpret_cast<float4 *>(mask_out + (int64_t)n * C + c) = make_float4(m[0], m[1], m[2], m[3]);
    }
}

__global__ __launch_bounds__(256) void avgpool_dropout_bwd_kernel(const float *dout, const float *mask, int N, int HW,
                                                                  int C, float scale, float *dx) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * HW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const int n = (int)(i / ((int64_t)HW * C4));
        float4 d = *reinterpret_cast<const float4 *>(dout + (int64_t)n * C + c);
        if (mask) {
            const float4 m = *reinterpret_cast<const float4 *>(mask + (int64_t)n * C + c);
            d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
        }
        d.x *= scale; d.y *= scale; d.z *= scale; d.w *= scale;
        *reinterpret_cast<float4 *>(dx + i * 4) = d;
    }
}

// ---- BatchNorm backward of a conv that feeds only a 3x3/2 max pool, straight from the POOLED gradient -------
// The pool's gradient w.r.t. its full-resolution input is never written: these two kernels rebuild it per 2x2
// input patch from the four windows that can name it (same walk as maxpool3s2_bwd_patch) and feed it into the
// BatchNorm(+ReLU) backward formulas,
//     g = dy_full * (z*rstd + shift > 0),   dbeta = sum g,   dz = rstd * (g - mean(g) - xhat * mean(g*xhat)).
// Saves the full-resolution write of MaxPoolGrad and its two re-reads (Conv2d_1a_7x7, Conv2d_2c_3x3).
struct PatchGrad {
    float g[2][2][4];      // [row y][col x][channel]: pooled gradient routed to input pixel (2p-pt+1+y, 2q-pl+1+x)
};

__device__ __forceinline__ PatchGrad patch_grad(const float *dpool, const uint8_t *am, int n, int p, int q, int c, int C,
                                                int OH, int OW) {
    float d[2][2][4];
    unsigned a[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int oh = p + u, ow = q + v;
            a[u][v] = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[u][v][j] = 0.f;
            if ((unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW) {
                const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
                a[u][v] = *reinterpret_cast<const unsigned *>(am + o);
                const float4 t = *reinterpret_cast<const float4 *>(dpool + o);
                d[u][v][0] = t.x; d[u][v][1] = t.y; d[u][v][2] = t.z; d[u][v][3] = t.w;
            }
        }
    PatchGrad r;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.g[y][x][j] = 0.f;
#pragma unroll
            for (int u = 0; u <= y; ++u)
#pragma unroll
                for (int v = 0; v <= x; ++v) {
                    const unsigned t = (unsigned)(((y + 1) - 2 * u) * 3 + ((x + 1) - 2 * v));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (((a[u][v] >> (8 * j)) & 0xFFu) == t) r.g[y][x][j] += d[u][v][j];
                }
        }
    return r;
}

// partials laid out [2][C][P], P = gridDim.x; the launch makes 256*gridDim.x a multiple of C/4 so that a thread
// keeps its channel group over its whole grid-stride walk
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const float *z, const float *dpool, const uint8_t *am,
                                                                 int N, int H, int W, int C, int pad_t, int pad_l, int OH,
                                                                 int OW, const float *mean, const float *rstd,
                                                                 const float *shift, float *partials) {
    __shared__ float sh[256][8];
    const int C4 = C >> 2;
    const int PH = OH + 1, PW = OW + 1;
    const int64_t total = (int64_t)N * PH * PW * C4;
    const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(first % C4) * 4;
    const float4 r4 = *reinterpret_cast<const float4 *>(rstd + c);
    const float4 s4 = *reinterpret_cast<const float4 *>(shift + c);
    const float4 m4 = *reinterpret_cast<const float4 *>(mean + c);
    const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = first; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i / C4;
        const int q = (int)(r % PW) - 1;
        r /= PW;
        const int p = (int)(r % PH) - 1;
        const int n = (int)(r / PH);
        const PatchGrad pg = patch_grad(dpool, am, n, p, q, c, C, OH, OW);
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int ih = 2 * p - pad_t + 1 + y;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int iw = 2 * q - pad_l + 1 + x;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 zv = *reinterpret_cast<const float4 *>(z + (((int64_t)n * H + ih) * W + iw) * C + c);
                const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = (zz[j] * rr[j] + ss[j] > 0.f) ? pg.g[y][x][j] : 0.f;
                    sg[j] += g;
                    sx[j] += g * ((zz[j] - mm[j]) * rr[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh[threadIdx.x][j] = sg[j];
        sh[threadIdx.x][4 + j] = sx[j];
    }
    __syncthreads();
    if (threadIdx.x < C4) {      // owner of channel group threadIdx.x: add its threads of this block in index order
        const int cg = threadIdx.x;
        int t0 = (int)(((int64_t)cg - (int64_t)blockIdx.x * 256) % C4);
        if (t0 < 0) t0 += C4;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = t0; t < 256; t += C4)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += sh[t][j];
        const int P = gridDim.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            partials[(int64_t)(cg * 4 + j) * P + blockIdx.x] = a[j];
            partials[((int64_t)C + cg * 4 + j) * P + blockIdx.x] = a[4 + j];
        }
    }
}

__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float *z, const float *dpool, const uint8_t *am,
                                                                int N, int H, int W, int C, int pad_t, int pad_l, int OH,
                                                                int OW, const float *mean, const float *rstd,
                                                                const float *shift, const float *coef, float *dz) {
    const int C4 = C >> 2;
    const int PH = OH + 1, PW = OW + 1;
    const int64_t total = (int64_t)N * PH * PW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int q = (int)(r % PW) - 1;
        r /= PW;
        const int p = (int)(r % PH) - 1;
        const int n = (int)(r / PH);
        const PatchGrad pg = patch_grad(dpool, am, n, p, q, c, C, OH, OW);
        const float4 r4 = *reinterpret_cast<const float4 *>(rstd + c);
        const float4 s4 = *reinterpret_cast<const float4 *>(shift + c);
        const float4 m4 = *reinterpret_cast<const float4 *>(mean + c);
        const float4 k1 = *reinterpret_cast<const float4 *>(coef + c);
        const float4 k2 = *reinterpret_cast<const float4 *>(coef + C + c);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
        const float a1[4] = {k1.x, k1.y, k1.z, k1.w}, a2[4] = {k2.x, k2.y, k2.z, k2.w};
        // the patch's four z values first, then the four stores: a load behind a store waits for it (one in-order memory
        // counter), which made the patch a chain of four round trips; z is read once (dz overwrites it): non-temporal
        float4 zv[2][2];
        bool ok[2][2];
        int64_t offs[2][2];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int ih = 2 * p - pad_t + 1 + y, iw = 2 * q - pad_l + 1 + x;
                ok[y][x] = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                offs[y][x] = (((int64_t)n * H + ih) * W + iw) * C + c;
                zv[y][x] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[y][x]) zv[y][x] = ds::ld_stream4(z + offs[y][x]);
            }
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                if (!ok[y][x]) continue;
                const float zz[4] = {zv[y][x].x, zv[y][x].y, zv[y][x].z, zv[y][x].w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = (zz[j] * rr[j] + ss[j] > 0.f) ? pg.g[y][x][j] : 0.f;
                    const float xhat = (zz[j] - mm[j]) * rr[j];
                    o[j] = rr[j] * (g - a1[j] - xhat * a2[j]);
                }
                *reinterpret_cast<float4 *>(dz + offs[y][x]) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

}  // namespace

namespace {
template <typename TI, typename TO>
void launch_pool3(int stride, hipStream_t st, const TI *x, TO *y, uint8_t *argmax, int N, int H, int W, int C, int pad_t,
                  int pad_l, int OH, int OW, const float *rstd, const float *shift, float *amax = nullptr) {
    const int64_t cols = (int64_t)N * OW * (C / 4);
    if (stride == 1)
        hipLaunchKernelGGL((maxpool3_fwd_rolling<1, TI, TO>), dim3(ds::stream_grid(cols, 256)), dim3(256), 0, st, x, y, argmax,
                           N, H, W, C, pad_t, pad_l, OH, OW, rstd, shift, amax);
    else
        hipLaunchKernelGGL((maxpool3_fwd_rolling<2, TI, TO>), dim3(ds::stream_grid(cols, 256)), dim3(256), 0, st, x, y, argmax,
                           N, H, W, C, pad_t, pad_l, OH, OW, rstd, shift, amax);
}
}  // namespace

extern "C" int ds_maxpool_fwd(const void *x, void *y, uint8_t *argmax, int32_t N, int32_t H, int32_t W, int32_t C,
                              int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW,
                              int32_t act_dtype, void *stream) {
    DS_REQUIRE(x && y && C % 4 == 0 && k >= 1 && k <= 15 && stride >= 1, "ds_maxpool_fwd: bad argument (C %% 4?)");
    DS_REQUIRE(act_dtype == DS_DTYPE_F32 || act_dtype == DS_DTYPE_BF16, "ds_maxpool_fwd: act_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16");
    const bool b16 = act_dtype == DS_DTYPE_BF16;
    hipStream_t st = (hipStream_t)stream;
    if (k == 3 && (stride == 1 || stride == 2)) {      // every 3x3 pool of Inception-v1: rolling window
        if (b16) launch_pool3(stride, st, (const __bf16 *)x, (__bf16 *)y, argmax, N, H, W, C, pad_t, pad_l, OH, OW, nullptr, nullptr);
        else launch_pool3(stride, st, (const float *)x, (float *)y, argmax, N, H, W, C, pad_t, pad_l, OH, OW, nullptr, nullptr);
        return ds::check_launch("ds_maxpool_fwd");
    }
    const int64_t total = (int64_t)N * OH * OW * (C / 4);
    if (b16)
        hipLaunchKernelGGL((maxpool_fwd_kernel<__bf16, __bf16>), dim3(ds::stream_grid(total, 256)), dim3(256), 0, st,
                           (const __bf16 *)x, (__bf16 *)y, argmax, N, H, W, C, k, stride, pad_t, pad_l, OH, OW);
    else
        hipLaunchKernelGGL((maxpool_fwd_kernel<float, float>), dim3(ds::stream_grid(total, 256)), dim3(256), 0, st,
                           (const float *)x, (float *)y, argmax, N, H, W, C, k, stride, pad_t, pad_l, OH, OW);
    return ds::check_launch("ds_maxpool_fwd");
}

extern "C" int ds_maxpool_bn_relu_fwd(const float *z, const float *rstd, const float *shift, void *y, uint8_t *argmax,
                                      int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                                      int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW, int32_t y_dtype, float *amax,
                                      void *stream) {
    DS_REQUIRE(z && rstd && shift && y && C % 4 == 0 && k >= 1 && stride >= 1, "ds_maxpool_bn_relu_fwd: bad argument (C %% 4 == 0)");
    DS_REQUIRE(y_dtype == DS_DTYPE_F32 || y_dtype == DS_DTYPE_BF16, "ds_maxpool_bn_relu_fwd: y_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16");
    if (!(k == 3 && (stride == 1 || stride == 2))) {       // other windows (MaxPool_5a 2x2/2): the generic kernel, fp32 output
        DS_REQUIRE(y_dtype == DS_DTYPE_F32 && !amax, "ds_maxpool_bn_relu_fwd: windows other than 3x3 write fp32 and track no maximum");
        const int64_t total = (int64_t)N * OH * OW * (C / 4);
        hipLaunchKernelGGL((maxpool_fwd_kernel<float, float>), dim3(ds::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           z, (float *)y, argmax, N, H, W, C, k, stride, pad_t, pad_l, OH, OW, rstd, shift);
        return ds::check_launch("ds_maxpool_bn_relu_fwd");
    }
    if (y_dtype == DS_DTYPE_BF16) launch_pool3(stride, (hipStream_t)stream, z, (__bf16 *)y, argmax, N, H, W, C, pad_t, pad_l, OH, OW, rstd, shift, amax);
    else launch_pool3(stride, (hipStream_t)stream, z, (float *)y, argmax, N, H, W, C, pad_t, pad_l, OH, OW, rstd, shift, amax);
    return ds::check_launch("ds_maxpool_bn_relu_fwd");
}

extern "C" int ds_maxpool_bwd(const float *dy, const uint8_t *argmax, float *dx, int32_t accumulate, int32_t N,
                              int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                              int32_t OH, int32_t OW, void *stream) {
    DS_REQUIRE(dy && argmax && dx && C % 4 == 0, "ds_maxpool_bwd: bad argument");
    if (k == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && OH == H && OW == W) {
        hipLaunchKernelGGL(maxpool3s1_bwd_rolling<float>, dim3(ds::stream_grid((int64_t)N * W * (C / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, dy, argmax, dx, accumulate, N, H, W, C);
        return ds::check_launch("ds_maxpool_bwd");
    }
    const int64_t total = (int64_t)N * H * W * (C / 4);
    const dim3 grid(ds::stream_grid(total, 256));
    if (k == 3 && stride == 2 && H <= 2 * OH && W <= 2 * OW)
        hipLaunchKernelGGL(maxpool3s2_bwd_patch, dim3(ds::stream_grid((int64_t)N * (OH + 1) * (OW + 1) * (C / 4), 256)),
                           dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, accumulate, N, H, W, C, pad_t, pad_l, OH,
                           OW);
    else if (k == 3 && stride == 2)
        hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2>), grid, dim3(256), 0, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, N, H, W, C, k, stride, pad_t, pad_l, OH, OW);
    else if (k == 2 && stride == 2)
        hipLaunchKernelGGL((maxpool_bwd_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, N, H, W, C, k, stride, pad_t, pad_l, OH, OW);
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, N, H, W, C, k, stride, pad_t, pad_l, OH, OW);
    return ds::check_launch("ds_maxpool_bwd");
}

extern "C" int ds_maxpool3_bwd_sums_partials(int32_t N, int32_t W, int32_t C) {
    if (N <= 0 || W <= 0 || C < 4) return 0;
    return pool_sums_shape(N, W, C).P;
}

// The 3x3 / 1 MaxPoolGrad reading its output gradient from bf16 storage (a Conv2DBackpropInput wrote it rounded:
// ds_conv_desc.z_dtype on the dgrad): partials == nullptr: the plain pass (ds_maxpool_bwd), else with the sums (ds_maxpool3_bwd_sums)
extern "C" int ds_maxpool3_bwd_dy16(const void *dy16, const uint8_t *argmax, float *dx, int32_t accumulate, const void *y,
                                    int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t C, float *partials,
                                    void *stream) {
    DS_REQUIRE(dy16 && argmax && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (((uintptr_t)dy16) & 7) == 0,
               "ds_maxpool3_bwd_dy16: bad argument (C %% 4 == 0, 8-byte aligned dy)");
    const __bf16 *dy = reinterpret_cast<const __bf16 *>(dy16);
    if (!partials) {
        hipLaunchKernelGGL(maxpool3s1_bwd_rolling<__bf16>, dim3(ds::stream_grid((int64_t)N * W * (C / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, dy, argmax, dx, accumulate, N, H, W, C);
        return ds::check_launch("ds_maxpool3_bwd_dy16");
    }
    DS_REQUIRE(y && C <= 1024 && (y_dtype == DS_DTYPE_F32 || y_dtype == DS_DTYPE_BF16), "ds_maxpool3_bwd_dy16: the sums need y (fp32 / bf16), C <= 1024");
    const PoolSumsShape sh = pool_sums_shape(N, W, C);
    const size_t shmem = (size_t)sh.RG * sh.CW * 8 * sizeof(float);
    if (y_dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL((maxpool3s1_bwd_sums_kernel<__bf16, __bf16>), dim3(sh.P * sh.chunks), dim3(256), shmem, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, (const __bf16 *)y, N, H, W, C, partials, sh.upb, sh.CW, sh.chunks);
    else
        hipLaunchKernelGGL((maxpool3s1_bwd_sums_kernel<float, __bf16>), dim3(sh.P * sh.chunks), dim3(256), shmem, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, (const float *)y, N, H, W, C, partials, sh.upb, sh.CW, sh.chunks);
    return ds::check_launch("ds_maxpool3_bwd_dy16");
}

extern "C" int ds_maxpool3_bwd_sums(const float *dy, const uint8_t *argmax, float *dx, int32_t accumulate, const void *y,
                                    int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t C, float *partials,
                                    void *stream) {
    DS_REQUIRE(dy && argmax && dx && y && partials && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 1024,
               "ds_maxpool3_bwd_sums: bad argument (C %% 4 == 0, C <= 1024)");
    DS_REQUIRE(y_dtype == DS_DTYPE_F32 || y_dtype == DS_DTYPE_BF16, "ds_maxpool3_bwd_sums: y_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16");
    DS_REQUIRE((int64_t)N * H * W * C < (1ll << 40), "ds_maxpool3_bwd_sums: tensor too large");
    const PoolSumsShape sh = pool_sums_shape(N, W, C);
    const size_t shmem = (size_t)sh.RG * sh.CW * 8 * sizeof(float);
    if (y_dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL((maxpool3s1_bwd_sums_kernel<__bf16, float>), dim3(sh.P * sh.chunks), dim3(256), shmem, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, (const __bf16 *)y, N, H, W, C, partials, sh.upb, sh.CW, sh.chunks);
    else
        hipLaunchKernelGGL((maxpool3s1_bwd_sums_kernel<float, float>), dim3(sh.P * sh.chunks), dim3(256), shmem, (hipStream_t)stream, dy, argmax, dx,
                           accumulate, (const float *)y, N, H, W, C, partials, sh.upb, sh.CW, sh.chunks);
    return ds::check_launch("ds_maxpool3_bwd_sums");
}

extern "C" int ds_avgpool_dropout_fwd(const float *x, int32_t N, int32_t HW, int32_t C, float keep, uint64_t seed,
                                      const uint64_t *seed_dev, const float *mask_in, float *mask_out, float *out,
                                      void *stream) {
    DS_REQUIRE(x && out && C % 4 == 0 && N > 0 && HW > 0 && keep > 0.f, "ds_avgpool_dropout_fwd: bad argument");
    hipLaunchKernelGGL(avgpool_dropout_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, N, HW, C, keep, seed,
                       seed_dev, mask_in, mask_out, out);
    return ds::check_launch("ds_avgpool_dropout_fwd");
}

extern "C" int ds_avgpool_dropout_bwd(const float *dout, const float *mask, int32_t N, int32_t HW, int32_t C,
                                      float keep, float *dx, void *stream) {
    DS_REQUIRE(dout && dx && C % 4 == 0 && keep > 0.f, "ds_avgpool_dropout_bwd: bad argument");
    const float scale = (keep >= 1.f ? 1.f : 1.f / keep) / (float)HW;
    const int64_t total = (int64_t)N * HW * (C / 4);
    hipLaunchKernelGGL(avgpool_dropout_bwd_kernel, dim3(ds::stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dout, keep >= 1.f ? nullptr : mask, N, HW, C, scale, dx);
    return ds::check_launch("ds_avgpool_dropout_bwd");
}

namespace {
int pool_bwd_grid(int N, int OH, int OW, int C) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * (OH + 1) * (OW + 1) * C4;
    int g = ds::stream_grid(total, 256 * 4);
    // 256*g must be a multiple of C4 (a thread keeps its channel group): g multiple of C4 / gcd(256, C4)
    int a = 256, b = C4;
    while (b) { const int t = a % b; a = b; b = t; }
    const int step = C4 / a;
    g = (g + step - 1) / step * step;
    return g;
}
}  // namespace

extern "C" int ds_bn_pool_bwd_partials(int32_t N, int32_t OH, int32_t OW, int32_t C) { return pool_bwd_grid(N, OH, OW, C); }

extern "C" int ds_bn_pool_bwd_reduce(const float *z, const float *dpool, const uint8_t *argmax, int32_t N, int32_t H,
                                     int32_t W, int32_t C, int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW,
                                     const float *mean, const float *rstd, const float *shift, float *partials,
                                     void *stream) {
    DS_REQUIRE(z && dpool && argmax && mean && rstd && shift && partials && C % 4 == 0 && C <= 1024 && H <= 2 * OH &&
                   W <= 2 * OW,
               "ds_bn_pool_bwd_reduce: bad argument (3x3 stride-2 SAME pools, C %% 4 == 0, C <= 1024)");
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(pool_bwd_grid(N, OH, OW, C)), dim3(256), 0, (hipStream_t)stream, z,
                       dpool, argmax, N, H, W, C, pad_t, pad_l, OH, OW, mean, rstd, shift, partials);
    return ds::check_launch("ds_bn_pool_bwd_reduce");
}

extern "C" int ds_bn_pool_bwd_apply(const float *z, const float *dpool, const uint8_t *argmax, int32_t N, int32_t H,
                                    int32_t W, int32_t C, int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW,
                                    const float *mean, const float *rstd, const float *shift, const float *coef,
                                    float *dz, void *stream) {
    DS_REQUIRE(z && dpool && argmax && mean && rstd && shift && coef && dz && C % 4 == 0 && H <= 2 * OH && W <= 2 * OW,
               "ds_bn_pool_bwd_apply: bad argument (3x3 stride-2 SAME pools, C %% 4 == 0)");
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(pool_bwd_grid(N, OH, OW, C)), dim3(256), 0, (hipStream_t)stream, z,
                       dpool, argmax, N, H, W, C, pad_t, pad_l, OH, OW, mean, rstd, shift, coef, dz);
    return ds::check_launch("ds_bn_pool_bwd_apply");
}
