// Text-tower, loss, optimiser and small plumbing kernels.  All HBM / latency bound.
//   embedding gather  tf.nn.embedding_lookup            image_text_model/im_text_rnn_model.py:85
//   LSTM cell         BasicLSTMCell + dynamic_rnn mask  im_text_rnn_model.py:89-90 (A7/A8)
//   softmax CE        slim.losses.softmax_cross_entropy im_text_rnn_model.py:124-125 (A9)
//   Adam              tf.train.AdamOptimizer            im_text_rnn_model.py:134-135 (A10)
#include <stdlib.h>
#include "ds_common.h"

namespace {

// ---- embedding gather: one wavefront per row, rows written in the LSTM's time-major order ----------
// A wave owns RPW consecutive (b, t) positions.  Per position: the id is a wave-uniform scalar load, the row is
// VEC-wide vectors handed to lanes 0, 1, ... (D = 300 -> 75 float4: one full and one 11-lane instruction), no
// per-element integer division.  Four rows are fetched before the first is stored (memory-level parallelism:
// the 12 MB table is cache resident, the stores are the HBM stream) and the stores are non-temporal -- the
// rows are next read by the projection GEMM, long after they left the caches at 2^20 tokens.
template <int VEC, bool NT = true, bool OUTORDER = false>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *table, const int64_t *ids, float *out, int B,
                                                          int T, int D, int64_t rows, int time_major, int rpw) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t total = (int64_t)B * T;
    int64_t r = wave * rpw;
    const int64_t r1 = r + rpw < total ? r + rpw : total;
    if (r >= r1) return;
    const int DV = D / VEC;
    // OUTORDER (time-major output): r walks the OUTPUT rows t * B + b, so a wave's stores -- and the stores of
    // consecutive waves -- form one linear stream (the ids are then read with stride T: 8 MB of scattered 8-byte reads
    // against 1.26 GB of row stores); otherwise r walks the id list and a time-major row lands B * D floats from the next
    int b, t;
    if (OUTORDER) { t = (int)(r / B); b = (int)(r - (int64_t)t * B); }
    else { b = (int)(r / T); t = (int)(r - (int64_t)b * T); }     // once per wave; then incremented
    constexpr int U = 4;                                           // rows in flight
    if (VEC == 4 && DV <= 128) {
        // Rows of at most 128 float4 (two chunks per lane), software pipelined: the id reads and the eight row loads of
        // the NEXT four rows go out before the stores of the current four.  In the plain loop below every load sits
        // behind the previous chunk's stores and waits for them (one in-order memory counter): two round trips per group.
        const float *srcA[U], *srcB[U];
        float *dstA[U], *dstB[U];
        bool okA[U], okB[U];
        vec_t a0[U], a1[U], b0[U], b1[U];
        auto fetch = [&](int64_t rr, const float *(&src)[U], float *(&dst)[U], bool (&ok)[U], vec_t (&v0)[U], vec_t (&v1)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = rr + u < r1;
                const int64_t id = in ? ids[OUTORDER ? (int64_t)b * T + t : rr + u] : -1;               // wave-uniform
                ok[u] = id >= 0 && id < rows;                          // out-of-range id -> zero row
                src[u] = table + (ok[u] ? id : 0) * D;
                dst[u] = in ? out + (OUTORDER ? rr + u : (time_major ? (int64_t)t * B + b : rr + u)) * D : nullptr;
                if (OUTORDER) { if (++b == B) { b = 0; ++t; } }
                else if (++t == T) { t = 0; ++b; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v0[u] = vec_t(0.f);
                v1[u] = vec_t(0.f);
                if (lane < DV && ok[u]) v0[u] = *reinterpret_cast<const vec_t *>(src[u] + lane * VEC);
                if (64 + lane < DV && ok[u]) v1[u] = *reinterpret_cast<const vec_t *>(src[u] + (64 + lane) * VEC);
            }
        };
        auto put = [&](float *(&dst)[U], vec_t (&v0)[U], vec_t (&v1)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!dst[u]) continue;
                if (lane < DV) {
                    if (NT) __builtin_nontemporal_store(v0[u], reinterpret_cast<vec_t *>(dst[u] + lane * VEC));
                    else *reinterpret_cast<vec_t *>(dst[u] + lane * VEC) = v0[u];
                }
                if (64 + lane < DV) {
                    if (NT) __builtin_nontemporal_store(v1[u], reinterpret_cast<vec_t *>(dst[u] + (64 + lane) * VEC));
                    else *reinterpret_cast<vec_t *>(dst[u] + (64 + lane) * VEC) = v1[u];
                }
            }
        };
        fetch(r, srcA, dstA, okA, a0, a1);
        for (; r < r1; r += 2 * U) {
            if (r + U < r1) fetch(r + U, srcB, dstB, okB, b0, b1);
            put(dstA, a0, a1);
            if (r + U >= r1) break;
            if (r + 2 * U < r1) fetch(r + 2 * U, srcA, dstA, okA, a0, a1);
            put(dstB, b0, b1);
        }
        return;
    }
    for (; r < r1; r += U) {
        const float *src[U];
        float *dst[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = r + u < r1;
            const int64_t id = in ? ids[OUTORDER ? (int64_t)b * T + t : r + u] : -1;               // wave-uniform
            ok[u] = id >= 0 && id < rows;                          // out-of-range id -> zero row
            src[u] = table + (ok[u] ? id : 0) * D;
            dst[u] = in ? out + (OUTORDER ? r + u : (time_major ? (int64_t)t * B + b : r + u)) * D : nullptr;
            if (OUTORDER) { if (++b == B) { b = 0; ++t; } }
            else if (++t == T) { t = 0; ++b; }
        }
        for (int v0 = 0; v0 < DV; v0 += 64) {
            const int v = v0 + lane;
            vec_t val[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                val[u] = vec_t(0.f);
                if (v < DV && ok[u]) val[u] = *reinterpret_cast<const vec_t *>(src[u] + v * VEC);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v < DV && dst[u]) {
                    if (NT) __builtin_nontemporal_store(val[u], reinterpret_cast<vec_t *>(dst[u] + v * VEC));
                    else *reinterpret_cast<vec_t *>(dst[u] + v * VEC) = val[u];
                }
        }
    }
}

// ---- embedding gradient: deterministic wavefront-reduced scatter-add ------------------------------
// dtable[v, :] = sum over the positions p = b*T + t with ids[p] == v of dx[row(p), :], one wave per
// vocabulary row.  The wave sweeps the id list 64 positions at a time, ballots the matches and adds the
// matching rows in ascending position order -- no atomics, no sort, bit-reproducible -- with lane l owning
// columns l, l+64, ... of the row.  Rows nobody referenced come out as zeros (the whole table is written).
// Not reachable from the reference graph (its embedding is trainable=False, im_text_rnn_model.py:82); it
// backs the optional full fine-tuning switch.
template <int NQ>
__global__ __launch_bounds__(256) void embedding_grad_kernel(const float *dx, const int64_t *ids, float *dtable, int B,
                                                             int T, int D, int64_t rows, int time_major) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= rows) return;
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
    const int BT = B * T;
    for (int base = 0; base < BT; base += 64) {
        const int pos = base + lane;
        const int64_t id = pos < BT ? ids[pos] : -1;
        unsigned long long mask = __ballot(id == v);
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int p = base + j;
            const int b = p / T, t = p - b * T;
            const float *src = dx + (time_major ? (int64_t)t * B + b : (int64_t)p) * D;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = lane + 64 * q;
                if (c < D) acc[q] += src[c];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = lane + 64 * q;
        if (c < D) dtable[v * D + c] = acc[q];
    }
}

// ---- LSTM cell -----------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float *gates, const float *rec, int nslabs,
                                                            int64_t slab_stride, const float *c_prev,
                                                            const float *h_prev, const int64_t *seq_len, int t, int B,
                                                            int H, float forget_bias, float *c_out, float *h_out) {
    const int64_t total = (int64_t)B * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / H), h = (int)(i - (int64_t)b * H);
        float *g = gates + (int64_t)b * 4 * H + h;
        float pi = g[0], pj = g[H], pf = g[2 * H], po = g[3 * H];     // x_t*Wx + bias (hoisted GEMM)
        for (int s = 0; s < nslabs; ++s) {                             // + h_{t-1}*Wh, split-K slabs
            const float *r = rec + s * slab_stride + (int64_t)b * 4 * H + h;
            pi += r[0]; pj += r[H]; pf += r[2 * H]; po += r[3 * H];
        }
        const float si = sigmoidf_(pi);
        const float tj = tanhf(pj);
        const float sf = sigmoidf_(pf + forget_bias);
        const float so = sigmoidf_(po);
        const float cp = c_prev[i], hp = h_prev[i];
        const float cn = cp * sf + si * tj;
        const float hn = tanhf(cn) * so;
        const bool live = (int64_t)t < seq_len[b];
        g[0] = si;
        g[H] = tj;
        g[2 * H] = sf;
        g[3 * H] = so;
        c_out[i] = live ? cn : cp;      // dynamic_rnn copies the state through past seq_len
        h_out[i] = live ? hn : hp;
    }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float *acts, const float *c_t, const float *c_prev,
                                                            const float *dh, const float *dh_slabs, int nslabs,
                                                            int64_t slab_stride, const float *dc,
                                                            const int64_t *seq_len, int t, int B, int H,
                                                            float *dgates, float *dc_prev, float *dh_carry) {
    const int64_t total = (int64_t)B * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / H), h = (int)(i - (int64_t)b * H);
        const float *a = acts + (int64_t)b * 4 * H + h;
        float *dg = dgates + (int64_t)b * 4 * H + h;
        const bool live = (int64_t)t < seq_len[b];
        float dhv = dh[i];                                     // carried / injected part of d(loss)/d(h_t)
        for (int s = 0; s < nslabs; ++s) dhv += dh_slabs[s * slab_stride + i];   // + dgates_{t+1}*Wh^T (split-K slabs)
        const float dcv = dc[i];
        if (live) {
            const float si = a[0], tj = a[H], sf = a[2 * H], so = a[3 * H];
            const float tc = tanhf(c_t[i]);
            const float dct = dcv + dhv * so * (1.f - tc * tc);
            dg[0] = dct * tj * si * (1.f - si);
            dg[H] = dct * si * (1.f - tj * tj);
            dg[2 * H] = dct * c_prev[i] * sf * (1.f - sf);
            dg[3 * H] = dhv * tc * so * (1.f - so);
            dc_prev[i] = dct * sf;
            dh_carry[i] = 0.f;
        } else {
            dg[0] = 0.f;
            dg[H] = 0.f;
            dg[2 * H] = 0.f;
            dg[3 * H] = 0.f;
            dc_prev[i] = dcv;
            dh_carry[i] = dhv;
        }
    }
}

// ---- softmax cross entropy (+ gradient), one workgroup, fixed summation order -----------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float *logits, const int64_t *labels, int B, int C,
                                                         float grad_scale, const float *grad_scale_dev,
                                                         float *loss, float *dlogits) {
    __shared__ float red[256];
    if (grad_scale_dev) grad_scale *= grad_scale_dev[0];   // upstream d(loss) handed over on device
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float *z = logits + (int64_t)b * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(z[c] - m);
        // a label outside [0, C) (dataset converted for another class list) must not index out of bounds:
        // it poisons the loss with NaN so the run stops visibly; one_hot() of such a label is all zeros in TF
        const int64_t yl = labels[b];
        const bool yok = yl >= 0 && yl < C;
        const int y = yok ? (int)yl : -1;
        acc += yok ? (m + logf(s)) - z[y] : NAN;
        if (dlogits) {
            const float k = grad_scale / (float)B, inv = 1.f / s;
            for (int c = 0; c < C; ++c)
                dlogits[(int64_t)b * C + c] = (expf(z[c] - m) * inv - (c == y ? 1.f : 0.f)) * k;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = red[0] / (float)B;
}

// ---- TF Adam over a flat buffer ------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_tf_kernel(float *theta, const float *g, float *m, float *v, int64_t n,
                                                      int64_t n_wd, float wd, float grad_scale, float lr_t,
                                                      const float *lr_t_dev, float b1, float b2, float eps) {
    if (lr_t_dev) lr_t = lr_t_dev[0];         // step size kept on device (hipGraph replay safe)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float w = theta[i];
        float gi = g[i] * grad_scale;
        if (i < n_wd) gi += wd * w;                       // d/dw of wd * sum(w^2)/2
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        theta[i] = w - lr_t * mi / (sqrtf(vi) + eps);     // epsilon outside the bias correction (A10)
    }
}

// ---- reductions / plumbing ---------------------------------------------------------------------------
constexpr int kSumsqBlocks = 256;

__global__ __launch_bounds__(256) void sumsq_stage1(const float *x, int64_t n, float *partials) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i] * x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sum_stage2(const float *partials, int n, float *out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)partials[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// column sums of x[M, C] (BiasAddGrad): grid (C/64, RS); thread = (column, one of 4 row lanes)
__global__ __launch_bounds__(256) void colsum_stage1(const float *x, int64_t M, int C, int ld, int rows_per_split,
                                                     float *partials) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t r1 = r0 + rows_per_split;
    if (r1 > M) r1 = M;
    float acc = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) acc += x[r * ld + c];
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) partials[(int64_t)blockIdx.y * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

// few row splits (M <= 512: the head biases at B <= 512): both stages in one launch, split by split in the same
// arithmetic and order (fp32 over 4 row lanes per split, double over the splits) -- same bits, one dispatch less
__global__ __launch_bounds__(256) void colsum_small_kernel(const float *x, int64_t M, int C, int ld, int rows_per_split,
                                                           int RS, float *out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double total = 0.0;
    for (int s = 0; s < RS; ++s) {
        const int64_t r0 = (int64_t)s * rows_per_split;
        int64_t r1 = r0 + rows_per_split;
        if (r1 > M) r1 = M;
        float acc = 0.f;
        if (c < C)
            for (int64_t r = r0 + rl; r < r1; r += 4) acc += x[r * ld + c];
        __syncthreads();
        red[rl][cl] = acc;
        __syncthreads();
        if (rl == 0) total += (double)(red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
    }
    if (rl == 0 && c < C) out[c] = (float)total;
}

__global__ __launch_bounds__(256) void colsum_stage2(const float *partials, int RS, int C, float *out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    for (int s = 0; s < RS; ++s) acc += (double)partials[(int64_t)s * C + c];
    out[c] = (float)acc;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float *src, int lds, float *dst, int ldd, int64_t rows,
                                                     int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}

__global__ __launch_bounds__(256) void pad_channels_kernel(const float *src, int cs, float *dst, int cd,
                                                           int64_t pixels) {
    const int64_t total = pixels * cd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / cd;
        const int c = (int)(i - p * cd);
        dst[i] = c < cs ? src[p * cs + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float *dst, int64_t n, float value) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = value;
}

}  // namespace

extern "C" int ds_gather_rows(const float *table, const int64_t *ids, float *out, int32_t B, int32_t T, int32_t D,
                              int64_t table_rows, int32_t time_major, void *stream) {
    DS_REQUIRE(table && ids && out && B > 0 && T > 0 && D > 0 && table_rows > 0, "ds_gather_rows: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const bool a16 = (((uintptr_t)table | (uintptr_t)out) & 15) == 0, a8 = (((uintptr_t)table | (uintptr_t)out) & 7) == 0;
    const int64_t total = (int64_t)B * T;
    // rows per wave: enough waves to fill the chip several times over, at most 16 rows each
    int rpw = (int)((total + (int64_t)ds::kCUs * 32 * 4 - 1) / ((int64_t)ds::kCUs * 32 * 4));
    rpw = rpw < 4 ? 4 : (rpw > 16 ? 16 : rpw);
    const int64_t waves = (total + rpw - 1) / rpw;
    const dim3 grid((unsigned)((waves + 3) / 4));
    static int plain = -1;
    if (plain < 0) {
        const char *e = ds::tune_env("DS_GATHER_PLAIN_STORES");      // A/B aid
        plain = e ? atoi(e) : 0;
    }
    static int outorder = -1;
    if (outorder < 0) {
        const char *e = ds::tune_env("DS_GATHER_OUTORDER");          // A/B aid
        outorder = e ? atoi(e) : 1;
    }
    if (D % 4 == 0 && a16 && !plain && time_major && outorder == 1) {
        hipLaunchKernelGGL((gather_rows_kernel<4, true, true>), grid, dim3(256), 0, s, table, ids, out, B, T, D, table_rows, time_major, rpw);
    } else if (D % 4 == 0 && a16 && time_major && outorder == 2) {
        hipLaunchKernelGGL((gather_rows_kernel<4, false, true>), grid, dim3(256), 0, s, table, ids, out, B, T, D, table_rows, time_major, rpw);
    } else if (D % 4 == 0 && a16 && plain) {
        hipLaunchKernelGGL((gather_rows_kernel<4, false>), grid, dim3(256), 0, s, table, ids, out, B, T, D, table_rows, time_major, rpw);
    } else if (D % 4 == 0 && a16) {
        hipLaunchKernelGGL(gather_rows_kernel<4>, grid, dim3(256), 0, s, table, ids, out, B, T, D, table_rows, time_major, rpw);
    } else if (D % 2 == 0 && a8) {
        hipLaunchKernelGGL(gather_rows_kernel<2>, grid, dim3(256), 0, s, table, ids, out, B, T, D, table_rows, time_major, rpw);
    } else {
        hipLaunchKernelGGL(gather_rows_kernel<1>, grid, dim3(256), 0, s, table, ids, out, B, T, D, table_rows, time_major, rpw);
    }
    return ds::check_launch("ds_gather_rows");
}

extern "C" int ds_embedding_grad(const float *dx, const int64_t *ids, float *dtable, int32_t B, int32_t T, int32_t D,
                                 int64_t table_rows, int32_t time_major, void *stream) {
    DS_REQUIRE(dx && ids && dtable && B > 0 && T > 0 && D > 0 && D <= 512 && table_rows > 0,
               "ds_embedding_grad: bad argument (D <= 512)");
    const dim3 grid((unsigned)((table_rows + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (D <= 64) hipLaunchKernelGGL(embedding_grad_kernel<1>, grid, block, 0, s, dx, ids, dtable, B, T, D, table_rows, time_major);
    else if (D <= 128) hipLaunchKernelGGL(embedding_grad_kernel<2>, grid, block, 0, s, dx, ids, dtable, B, T, D, table_rows, time_major);
    else if (D <= 320) hipLaunchKernelGGL(embedding_grad_kernel<5>, grid, block, 0, s, dx, ids, dtable, B, T, D, table_rows, time_major);
    else hipLaunchKernelGGL(embedding_grad_kernel<8>, grid, block, 0, s, dx, ids, dtable, B, T, D, table_rows, time_major);
    return ds::check_launch("ds_embedding_grad");
}

extern "C" int ds_lstm_cell_fwd(float *gates, const float *rec_slabs, int32_t nslabs, int64_t slab_stride,
                                const float *c_prev, const float *h_prev, const int64_t *seq_len, int32_t t, int32_t B,
                                int32_t H, float forget_bias, float *c_out, float *h_out, void *stream) {
    DS_REQUIRE(gates && c_prev && h_prev && seq_len && c_out && h_out && B > 0 && H > 0 && nslabs >= 0 &&
                   (nslabs == 0 || rec_slabs),
               "ds_lstm_cell_fwd: bad argument");
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ds::stream_grid((int64_t)B * H, 256)), dim3(256), 0,
                       (hipStream_t)stream, gates, rec_slabs, nslabs, slab_stride, c_prev, h_prev, seq_len, t, B, H,
                       forget_bias, c_out, h_out);
    return ds::check_launch("ds_lstm_cell_fwd");
}

extern "C" int ds_lstm_cell_bwd(const float *acts, const float *c_t, const float *c_prev, const float *dh,
                                const float *dh_slabs, int32_t nslabs, int64_t slab_stride, const float *dc,
                                const int64_t *seq_len, int32_t t, int32_t B, int32_t H, float *dgates,
                                float *dc_prev, float *dh_carry, void *stream) {
    DS_REQUIRE(acts && c_t && c_prev && dh && dc && seq_len && dgates && dc_prev && dh_carry && nslabs >= 0 &&
                   (nslabs == 0 || dh_slabs),
               "ds_lstm_cell_bwd: bad argument");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ds::stream_grid((int64_t)B * H, 256)), dim3(256), 0,
                       (hipStream_t)stream, acts, c_t, c_prev, dh, dh_slabs, nslabs, slab_stride, dc, seq_len, t, B,
                       H, dgates, dc_prev, dh_carry);
    return ds::check_launch("ds_lstm_cell_bwd");
}

extern "C" int ds_softmax_ce(const float *logits, const int64_t *labels, int32_t B, int32_t C, float grad_scale,
                             const float *grad_scale_dev, float *loss, float *dlogits, void *stream) {
    DS_REQUIRE(logits && labels && B > 0 && C > 0, "ds_softmax_ce: bad argument");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, B, C, grad_scale,
                       grad_scale_dev, loss, dlogits);
    return ds::check_launch("ds_softmax_ce");
}

extern "C" int ds_adam_tf(float *theta, const float *g, float *m, float *v, int64_t n, int64_t n_wd, float wd,
                          float grad_scale, float lr_t, const float *lr_t_dev, float beta1, float beta2, float eps,
                          void *stream) {
    DS_REQUIRE(theta && g && m && v && n > 0 && n_wd >= 0 && n_wd <= n, "ds_adam_tf: bad argument");
    hipLaunchKernelGGL(adam_tf_kernel, dim3(ds::stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, theta, g, m, v,
                       n, n_wd, wd, grad_scale, lr_t, lr_t_dev, beta1, beta2, eps);
    return ds::check_launch("ds_adam_tf");
}

extern "C" int ds_sumsq(const float *x, int64_t n, float *scratch, float *out, void *stream) {
    DS_REQUIRE(x && scratch && out && n > 0, "ds_sumsq: bad argument");
    int blocks = ds::stream_grid(n, 256 * 8);
    if (blocks > kSumsqBlocks) blocks = kSumsqBlocks;
    hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
    hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, out);
    return ds::check_launch("ds_sumsq");
}

// ---- length-sorted batches (ds_seq_sort_desc, ds_permute_rows) ------------------------------------------------------------
// rank of sample i = #{j : len_j > len_i, or len_j == len_i and j < i}: descending length, stable.  One workgroup, B^2 / 256
// comparisons per thread (B = 256: 256) -- the batch is tiny, the point is that nothing leaves the device.
__global__ __launch_bounds__(256) void seq_sort_desc_kernel(const int64_t *seq_len, int B, int T, int *perm, int64_t *len_sorted) {
    extern __shared__ int lens[];
    for (int i = threadIdx.x; i < B; i += 256) {
        const int64_t l = seq_len[i];
        lens[i] = (int)(l < 0 ? 0 : (l > T ? T : l));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const int li = lens[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int lj = lens[j];
            rank += (lj > li || (lj == li && j < i)) ? 1 : 0;
        }
        perm[rank] = i;
        len_sorted[rank] = li;
    }
}

template <typename E>
__global__ __launch_bounds__(256) void permute_rows_kernel(const E *src, int64_t lds, E *dst, int64_t ldd, const int *perm,
                                                           int rows, int cols, int gather) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        const int pr = perm[r];
        if (gather) dst[(int64_t)r * ldd + c] = src[(int64_t)pr * lds + c];
        else dst[(int64_t)pr * ldd + c] = src[(int64_t)r * lds + c];
    }
}

extern "C" int ds_seq_sort_desc(const int64_t *seq_len, int32_t B, int32_t T, int32_t *perm, int64_t *len_sorted, void *stream) {
    DS_REQUIRE(seq_len && perm && len_sorted && B > 0 && B <= 4096 && T > 0, "ds_seq_sort_desc: bad argument (1 <= B <= 4096)");
    hipLaunchKernelGGL(seq_sort_desc_kernel, dim3(1), dim3(256), (size_t)B * sizeof(int), (hipStream_t)stream, seq_len, B, T, perm,
                       len_sorted);
    return ds::check_launch("ds_seq_sort_desc");
}

extern "C" int ds_permute_rows(const void *src, int64_t lds, void *dst, int64_t ldd, const int32_t *perm, int32_t rows, int32_t cols,
                               int32_t elem_bytes, int32_t gather, void *stream) {
    DS_REQUIRE(src && dst && perm && rows > 0 && cols > 0 && lds >= cols && ldd >= cols && (elem_bytes == 4 || elem_bytes == 8),
               "ds_permute_rows: bad argument (elem_bytes 4 or 8)");
    const int grid = ds::stream_grid((int64_t)rows * cols, 256);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(permute_rows_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)src, lds,
                           (float *)dst, ldd, perm, rows, cols, gather);
    else
        hipLaunchKernelGGL(permute_rows_kernel<int64_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const int64_t *)src, lds,
                           (int64_t *)dst, ldd, perm, rows, cols, gather);
    return ds::check_launch("ds_permute_rows");
}

extern "C" int ds_colsum(const float *x, int64_t M, int32_t C, int32_t ld, float *scratch, float *out, void *stream) {
    DS_REQUIRE(x && scratch && out && M > 0 && C > 0 && ld >= C, "ds_colsum: bad argument");
    int RS = (int)((M + 63) / 64);
    if (RS > 64) RS = 64;
    const int rps = (int)((M + RS - 1) / RS);
    if (RS <= 8) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, rps, RS, out);
        return ds::check_launch("ds_colsum");
    }
    hipLaunchKernelGGL(colsum_stage1, dim3((C + 63) / 64, RS), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, rps,
                       scratch);
    hipLaunchKernelGGL(colsum_stage2, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, RS, C, out);
    return ds::check_launch("ds_colsum");
}

// out[m][n] = epilogue(sum_s slabs[s][m][n]): the second half of a split-K GEMM (ds_conv_igemm with splits > 1 writes slab s at
// z + s * z_split_stride).  The M <= 512 GEMMs of the heads are ONE row tile or two: a single launch walks K = 512 ... 1024
// serially in a handful of workgroups (40-57 us each, on the chain between the towers' forward and backward passes);
// split eight ways and combined here in a fixed order (deterministic) they take a third of that.
__global__ __launch_bounds__(256) void slab_epilogue_kernel(const float *slabs, int splits, int64_t slab_stride, int lds,
                                                            int64_t M, int N, float *out, int ldo, const float *bias,
                                                            const float *mask, int ldmask, int flags) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        float v = 0.f;
        for (int k = 0; k < splits; ++k) v += slabs[(int64_t)k * slab_stride + m * lds + n];
        if (flags & DS_EPI_BIAS) v += bias[n];
        const int64_t o = m * ldo + n;
        if (flags & DS_EPI_ACCUM) v += out[o];
        if (flags & DS_EPI_MASK) v = mask[m * ldmask + n] > 0.f ? v : 0.f;
        if (flags & DS_EPI_RELU) v = fmaxf(v, 0.f);
        out[o] = v;
    }
}

extern "C" int ds_slab_epilogue(const float *slabs, int32_t splits, int64_t slab_stride, int32_t lds, int64_t M, int32_t N,
                                float *out, int32_t ldo, const float *bias, const float *mask, int32_t ldmask, int32_t flags,
                                void *stream) {
    DS_REQUIRE(slabs && out && splits >= 1 && M > 0 && N > 0 && lds >= N && ldo >= N, "ds_slab_epilogue: bad argument");
    DS_REQUIRE((flags & ~(DS_EPI_BIAS | DS_EPI_ACCUM | DS_EPI_MASK | DS_EPI_RELU)) == 0 && (!(flags & DS_EPI_BIAS) || bias) &&
                   (!(flags & DS_EPI_MASK) || (mask && ldmask >= N)),
               "ds_slab_epilogue: flags are BIAS / ACCUM / MASK / RELU (with their operands)");
    hipLaunchKernelGGL(slab_epilogue_kernel, dim3(ds::stream_grid(M * N, 256)), dim3(256), 0, (hipStream_t)stream, slabs, splits,
                       slab_stride, lds, M, N, out, ldo, bias, mask, ldmask, flags);
    return ds::check_launch("ds_slab_epilogue");
}

extern "C" int ds_copy2d(const float *src, int32_t lds, float *dst, int32_t ldd, int64_t rows, int32_t cols,
                         void *stream) {
    DS_REQUIRE(src && dst && rows > 0 && cols > 0, "ds_copy2d: bad argument");
    hipLaunchKernelGGL(copy2d_kernel, dim3(ds::stream_grid(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       lds, dst, ldd, rows, cols);
    return ds::check_launch("ds_copy2d");
}

extern "C" int ds_pad_channels(const float *src, int32_t cs, float *dst, int32_t cd, int64_t pixels, void *stream) {
    DS_REQUIRE(src && dst && cs > 0 && cd >= cs && pixels > 0, "ds_pad_channels: bad argument");
    hipLaunchKernelGGL(pad_channels_kernel, dim3(ds::stream_grid(pixels * cd, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, cs, dst, cd, pixels);
    return ds::check_launch("ds_pad_channels");
}

extern "C" int ds_fill(float *dst, int64_t n, float value, void *stream) {
    DS_REQUIRE(dst && n > 0, "ds_fill: bad argument");
    hipLaunchKernelGGL(fill_kernel, dim3(ds::stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, n, value);
    return ds::check_launch("ds_fill");
}
