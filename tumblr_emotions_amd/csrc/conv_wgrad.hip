// Conv2DBackpropFilter / transposed MatMul (wgrad) on fp32 MFMA, split over the pixel dimension.
//
//   dw[tap, ci, co] = sum_m x[pixel(m) + tap, ci] * dz[m, co]
//
// Only the trainable scope needs it (image_model/inception_v1.py:229-250 Mixed_5c, :302-303
// Logits, and the tf.get_variable matrices of im_text_rnn_model.py:89,98-104), i.e. ~4 % of the
// backward FLOPs, so the kernel favours simplicity: both operands are staged pixel-major in LDS
// exactly as they lie in HBM (NHWC rows), fragments are read with ds_read_b32 (32 consecutive
// channels per half-wave: conflict-free), and the reduction over pixels is cut into `splits`
// slabs that a second kernel sums in a fixed order (deterministic, no atomics).
#include <stdio.h>
#include <stdlib.h>
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TJ = 128;   // co tile
constexpr int TP = 16;    // pixels per K-tile

struct WgradParams {
    ds_conv_desc d;
    const float *x;
    const float *dz;
    float *out;         // dw (splits==1) or workspace [splits][taps*Cin*Cout]
    int lddz;
    int M;
    int ci_tiles;
    int pix_per_split;  // multiple of TP
    int x_vec, z_vec;
    unsigned x_bytes, z_bytes;   // extents covered by the two buffer descriptors
    float inv_ohw, inv_ow;
    int slabs;          // wgrad_direct_kernel: split-K slabs (see the id -> slab mapping there)
};

constexpr unsigned kOOB = 0x80000000u;   // byte offset beyond any descriptor: the load returns 0

// SRD buffer loads as in conv_igemm.hip: rows past the split, padding pixels and channels past the
// tensor get an out-of-range offset and come back as zeros, so the loader has no branches and the
// loads of a K-tile stay in flight under the MFMAs of the previous one.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 load4(__amdgpu_buffer_rsrc_t r, unsigned off, bool ok, bool vec, int valid) {
    if (vec) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : kOOB, 0, 0));
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (ok && j < valid) ? off + 4 * j : kOOB, 0, 0));
    return v;
}

// floor(a / b) for 0 <= a < 2^24 with inv = 1.0f / b (one correction step each way)
__device__ __forceinline__ int fdiv(int a, int b, float inv) {
    int q = (int)((float)a * inv);
    int r = a - q * b;
    if (r < 0) --q;
    if (r >= b) ++q;
    return q;
}

// TIT = ci tile: 128 (4 waves as 2x2, 64x64 each) or 64 (4 waves side by side, 64x32 each) for layers whose Cin
// a 128-wide tile would pad by a quarter or more (Cin <= 64, 144, 160, 192 ...: every conv below Mixed_5c under
// train_all, Branch_2 of Mixed_5c in the reference freeze).
template <int TIT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int LDX = TIT + 4, LDZ = TJ + 4;
    constexpr int WNW = TIT == 128 ? 2 : 4;              // waves along co
    constexpr int CW = TJ / WNW;                          // co columns per wave: 64 or 32
    constexpr int BS = CW / 32;                           // 32-wide co sub-tiles per wave
    constexpr int XQ = TIT / 4;                           // float4 per X row
    constexpr int XROWS = 256 / XQ;                       // X rows fetched per pass: 8 or 16
    constexpr int XP = TP / XROWS;                        // X passes per K-tile: 2 or 1
    __shared__ __attribute__((aligned(16))) float smem[2 * TP * LDX + 2 * TP * LDZ];
    float *Xs = smem;                    // [2][TP][LDX]
    float *Zs = smem + 2 * TP * LDX;     // [2][TP][LDZ]
    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = TIT == 128 ? wave >> 1 : 0, wn = TIT == 128 ? wave & 1 : wave;
    const int li = lane & 31, lk = lane >> 5;
    const int tap = blockIdx.x / p.ci_tiles;
    const int i0 = (blockIdx.x % p.ci_tiles) * TIT;
    const int j0 = blockIdx.y * TJ;
    const int split = blockIdx.z;
    const int dh = tap / d.KW, dw = tap % d.KW;
    const int ohw = d.OH * d.OW;
    const int m_begin = split * p.pix_per_split;
    int m_end = m_begin + p.pix_per_split;
    if (m_end > p.M) m_end = p.M;

    int it_valid = (d.Cin - (i0 + wm * 64) + 31) / 32;
    it_valid = it_valid < 0 ? 0 : (it_valid > 2 ? 2 : it_valid);
    int jt_valid = (d.Cout - (j0 + wn * CW) + 31) / 32;
    jt_valid = jt_valid < 0 ? 0 : (jt_valid > BS ? BS : jt_valid);

    f32x16 acc[2][BS];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < BS; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int xrow = tid / XQ, xc4 = (tid % XQ) * 4;     // this thread's float4 of the X tile
    const int zrow = tid >> 5, zc4 = (tid & 31) * 4;     // ... and of the dz tile (two passes of 8 rows)
    f32x4 rx[XP], rz[2];
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_z = make_srd(p.dz, p.z_bytes);
    const int cx = i0 + xc4, cz = j0 + zc4;
    const bool small = p.M < (1 << 24);

    auto load_tile = [&](int mt0) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int m = mt0 + xrow + XROWS * i;
            const bool rv = m < m_end;
            const int mm = rv ? m : 0;
            const int n = small ? fdiv(mm, ohw, p.inv_ohw) : mm / ohw;
            const int r = mm - n * ohw;
            const int oh = small ? fdiv(r, d.OW, p.inv_ow) : r / d.OW;
            const int ow = r - oh * d.OW;
            const int ih = oh * d.stride - d.pad_t + dh, iw = ow * d.stride - d.pad_l + dw;
            // fold_cin > 0 (stem): KW is folded into the channel axis, channel cx lives in pixel iw + cx/fold_cin
            const int iwc = d.fold_cin > 0 ? iw + cx / d.fold_cin : iw;
            const bool okx = rv && (unsigned)ih < (unsigned)d.H && (unsigned)iwc < (unsigned)d.W && cx < d.Cin;
            const unsigned offx = (unsigned)(((n * d.H + ih) * d.W + iw) * d.ldx + cx);
            rx[i] = load4(srd_x, offx * 4u, okx, p.x_vec, d.Cin - cx);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mt0 + zrow + 8 * i;
            const bool okz = m < m_end && cz < d.Cout;
            const unsigned offz = (unsigned)(m < m_end ? m : 0) * (unsigned)p.lddz + (unsigned)cz;
            rz[i] = load4(srd_z, offz * 4u, okz, p.z_vec, d.Cout - cz);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XP; ++i)
            *reinterpret_cast<f32x4 *>(Xs + (buf * TP + xrow + XROWS * i) * LDX + xc4) = rx[i];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<f32x4 *>(Zs + (buf * TP + zrow + 8 * i) * LDZ + zc4) = rz[i];
    };
    auto compute = [&](int buf) {
        const float *xs = Xs + buf * TP * LDX + wm * 64 + li;
        const float *zs = Zs + buf * TP * LDZ + wn * CW + li;
#pragma unroll
        for (int s = 0; s < TP / 2; ++s) {
            const int row = 2 * s + lk;
            float af[2], bf[BS];
            af[0] = xs[row * LDX];
            af[1] = xs[row * LDX + 32];
#pragma unroll
            for (int b = 0; b < BS; ++b) bf[b] = zs[row * LDZ + 32 * b];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < BS; ++b)
                    if (a < it_valid && b < jt_valid)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    };

    const int KT = (m_end - m_begin + TP - 1) / TP;
    if (KT > 0) {
        load_tile(m_begin);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_tile(m_begin + (kt + 1) * TP);
            compute(cur);
            if (kt + 1 < KT) store_tile(cur ^ 1);
            __syncthreads();
        }
    }

    float *out = p.out + (int64_t)split * (d.KH * d.KW) * d.Cin * d.Cout + (int64_t)tap * d.Cin * d.Cout;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            const int col = j0 + wn * CW + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < d.Cin && col < d.Cout) out[(int64_t)row * d.Cout + col] = acc[a][b][r];
            }
        }
}


// ================================================================================================
// Register-direct wgrad (round 2): no LDS in the main loop, one wave = one (32 AI) x (32 BJ) block of dw.
//
// For C[ci, co] = sum_m x[m, ci] dz[m, co] both MFMA operands are "k-major": lane (i, kh) of v_mfma_f32_32x32x2_f32
// needs x[m0 + kh][channel i] and dz[m0 + kh][column i] -- 32 consecutive lanes read 32 consecutive channels of ONE
// pixel row, i.e. the operands are coalesced exactly as they lie in HBM.  Which physical channel plays "row i" of a
// block is free, so lane i takes AI consecutive channels (AI*i .. AI*i + AI - 1) with ONE 4*AI-byte load and uses
// component a as the A operand of block a (block a = channels {AI*i + a}); the same for dz with BJ.  A K step (two
// pixels) is therefore two loads (at AI = BJ = 4: two coalesced 512-byte rows per half-wave) for AI*BJ MFMAs (16
// at 4x4 = 1024 matrix cycles) -- the load path is idle and four steps of prefetch hide the latency.
// Taps: the lane walks its own pixel (n, oh, ow) incrementally, the tap shift and the SAME padding are folded into
// the byte offset (out-of-range offset -> zeros).  The four waves of a workgroup take four quarters of the
// workgroup's pixel range and are summed through LDS at the end; workgroups along gridDim.y are the split-K slabs
// summed by splitk_reduce_kernel in a fixed order (deterministic, no atomics).
// The transposed stores put component b back next to its neighbours: out[ci][co0 + BJ*li + 0..BJ-1] is one store.
// ================================================================================================
#ifndef DS_WG_U
#define DS_WG_U 4          // K steps of operands in flight per wave (tuning builds: -DDS_WG_U=8)
#endif
template <int W>
__device__ __forceinline__ f32x4 loadw(__amdgpu_buffer_rsrc_t r, unsigned off) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if constexpr (W == 4) {
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    } else if constexpr (W == 2) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
        v[0] = t[0]; v[1] = t[1];
    } else {
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
    return v;
}

template <int AI, int BJ>
__global__ __launch_bounds__(256, (AI * BJ > 8) ? 1 : 2) void wgrad_direct_kernel(const WgradParams p) {
    constexpr int NACC = AI * BJ, U = DS_WG_U;
    __shared__ __attribute__((aligned(16))) float red[2 * NACC * 16 * 64];
    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int co_tiles = (d.Cout + 32 * BJ - 1) / (32 * BJ);
    // slab = id % slabs with slabs in {1, 2, 4} or a multiple of 8: the hardware deals workgroup ids round-robin to the
    // 8 XCDs, so every workgroup of a slab runs on the same XCD(s) and the tiles resident there at any time (a run of
    // consecutive tile ids, co fastest) walk the slab's pixel rows in step -- x and dz cross the fabric about once
    const int split = (int)(blockIdx.x % (unsigned)p.slabs);
    int bid = (int)(blockIdx.x / (unsigned)p.slabs);
    const int jt = bid % co_tiles; bid /= co_tiles;
    const int itile = bid % p.ci_tiles;
    const int tap = bid / p.ci_tiles;
    const int dh = tap / d.KW, dw = tap % d.KW;
    const int ci0 = itile * 32 * AI, co0 = jt * 32 * BJ;
    const int ci = ci0 + AI * li, co = co0 + BJ * li;
    const bool ci_ok = ci < d.Cin, co_ok = co < d.Cout;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_z = make_srd(p.dz, p.z_bytes);

    // pixel range: workgroup's slab, then this wave's quarter (even length so the two k lanes stay paired)
    const int m_begin = split * p.pix_per_split;
    int m_end = m_begin + p.pix_per_split;
    if (m_end > p.M) m_end = p.M;
    int q = (m_end - m_begin + 3) / 4;
    q = (q + 1) & ~1;
    const int wb = m_begin + wave * q;
    int we = wb + q;
    if (we > m_end) we = m_end;
    const int steps = we > wb ? (we - wb + 1) / 2 : 0;

    // The lane walks pixels m, m + 2, m + 4, ... of its wave's quarter.  Everything per step is an add or a select:
    // the input pixel of (n, oh, ow) at this tap is a byte offset that advances by a constant per step, plus a
    // constant when ow wraps to the next row and another when oh wraps to the next image (unsigned arithmetic, the
    // intermediate values of padding pixels may wrap around); the padding test is one unsigned compare per axis
    // against the tap's valid [lo, hi] output range.
    int m = wb + lk;
    const int ohw = d.OH * d.OW;
    int oh = 0, ow = 0;
    unsigned offx, offz;
    {
        const int mm = m < p.M ? m : 0;
        const int n = mm / ohw;
        const int r = mm - n * ohw;
        oh = r / d.OW;
        ow = r - oh * d.OW;
        const int ih = oh * d.stride - d.pad_t + dh, iw = ow * d.stride - d.pad_l + dw;
        offx = ((unsigned)((n * d.H + ih) * d.W + iw) * (unsigned)d.ldx + (unsigned)ci) * 4u;
        offz = ((unsigned)mm * (unsigned)p.lddz + (unsigned)co) * 4u;
    }
    const unsigned ldx4 = (unsigned)d.ldx * 4u;
    const unsigned step_x = 2u * (unsigned)d.stride * ldx4;
    const unsigned row_x = (unsigned)((d.W - d.OW) * d.stride) * ldx4;                // ow -= OW, oh += 1
    const unsigned img_x = (unsigned)((d.H - d.OH * d.stride) * d.W) * ldx4;          // oh -= OH, n += 1
    const unsigned step_z = 2u * (unsigned)p.lddz * 4u;
    // valid output coordinates for this tap: 0 <= o * stride - pad + tap < extent
    auto lo_of = [](int pad, int tap, int stride) { const int v = pad - tap; return v <= 0 ? 0 : (v + stride - 1) / stride; };
    auto hi_of = [](int pad, int tap, int stride, int ext, int oext) {
        const int v = ext - 1 + pad - tap;          // o * stride <= v
        const int h = v < 0 ? -1 : v / stride;
        return h > oext - 1 ? oext - 1 : h;
    };
    const int lo_h = lo_of(d.pad_t, dh, d.stride), lo_w = lo_of(d.pad_l, dw, d.stride);
    const unsigned span_h = (unsigned)(hi_of(d.pad_t, dh, d.stride, d.H, d.OH) - lo_h);
    const unsigned span_w = (unsigned)(hi_of(d.pad_l, dw, d.stride, d.W, d.OW) - lo_w);
    const int OWc = d.OW, OHc = d.OH;
    auto offsets = [&](unsigned &ox, unsigned &oz) {
        const bool mv = m < we;
        const bool in = (unsigned)(oh - lo_h) <= span_h && (unsigned)(ow - lo_w) <= span_w;
        oz = (mv && co_ok) ? offz : kOOB;
        ox = (mv && ci_ok && in) ? offx : kOOB;
        m += 2;
        offz += step_z;
        offx += step_x;
        ow += 2;
        // up to two row wraps per step when OW == 1 (the GEMM case H = W = 1 has OH = 1 too: both wrap every step)
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1 && OWc > 1) break;
            const bool wrap = ow >= OWc;
            ow = wrap ? ow - OWc : ow;
            oh += wrap ? 1 : 0;
            offx += wrap ? row_x : 0u;
            const bool wrap2 = oh >= OHc;
            oh = wrap2 ? oh - OHc : oh;
            offx += wrap2 ? img_x : 0u;
        }
    };

    f32x16 acc[AI][BJ];
#pragma unroll
    for (int a = 0; a < AI; ++a)
#pragma unroll
        for (int b = 0; b < BJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 xa[U], zb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        unsigned ox, oz;
        offsets(ox, oz);
        xa[u] = loadw<AI>(srd_x, ox);
        zb[u] = loadw<BJ>(srd_z, oz);
    }
    for (int s = 0; s < steps; s += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 a4 = xa[u], b4 = zb[u];
            unsigned ox, oz;
            offsets(ox, oz);                  // step s + u + U (past the range: zeros)
            xa[u] = loadw<AI>(srd_x, ox);
            zb[u] = loadw<BJ>(srd_z, oz);
#pragma unroll
            for (int a = 0; a < AI; ++a)
#pragma unroll
                for (int b = 0; b < BJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[a], b4[b], acc[a][b], 0, 0, 0);
        }
    }

    // ---- sum the four waves: 2,3 -> 0,1 then 1 -> 0 ------------------------------------------------------------
    auto put = [&](int slot) {
#pragma unroll
        for (int a = 0; a < AI; ++a)
#pragma unroll
            for (int b = 0; b < BJ; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v = {acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                    *reinterpret_cast<f32x4 *>(red + ((slot * NACC + a * BJ + b) * 4 + r4) * 256 + lane * 4) = v;
                }
    };
    auto get = [&](int slot) {
#pragma unroll
        for (int a = 0; a < AI; ++a)
#pragma unroll
            for (int b = 0; b < BJ; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(red + ((slot * NACC + a * BJ + b) * 4 + r4) * 256 + lane * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[a][b][4 * r4 + e] += v[e];
                }
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) get(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave != 0) return;
    get(0);

    float *out = p.out + (int64_t)split * (d.KH * d.KW) * d.Cin * d.Cout + (int64_t)tap * d.Cin * d.Cout;
    if (!co_ok) return;
#pragma unroll
    for (int a = 0; a < AI; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ci0 + AI * ((r & 3) + 8 * (r >> 2) + 4 * lk) + a;
            if (row < d.Cin) {
                float *o = out + (int64_t)row * d.Cout + co;
                if constexpr (BJ == 4) {
                    *reinterpret_cast<f32x4 *>(o) = f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                } else if constexpr (BJ == 2) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<f32x2 *>(o) = f32x2{acc[a][0][r], acc[a][1][r]};
                } else {
                    *o = acc[a][0][r];
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *ws, float *dw, int64_t n, int splits) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * n + i];
        dw[i] = s;
    }
}

// 64-wide ci tiles when they cut the padded reduction width by a fifth or more
int pick_ti(const ds_conv_desc *d) {
    const int p128 = (d->Cin + 127) / 128 * 128, p64 = (d->Cin + 63) / 64 * 64;
    return (p128 - p64) * 5 >= p128 ? 64 : 128;
}

int pick_splits(const ds_conv_desc *d, int64_t M) {
    const int ti = pick_ti(d);
    const int tiles = d->KH * d->KW * ((d->Cin + ti - 1) / ti) * ((d->Cout + TJ - 1) / TJ);
    static int occ = 0, minpix = 0;
    if (!occ) {
        const char *e = ds::tune_env("DS_WGRAD_OCC");
        occ = e ? atoi(e) : 3;          // measured best of 2/3/4/6 workgroups per CU (MI355X, joint step)
        e = ds::tune_env("DS_WGRAD_MINPIX");
        minpix = e ? atoi(e) : 64;
    }
    int splits = (occ * ds::kCUs + tiles - 1) / tiles;
    const int max_splits = (int)((M + minpix - 1) / minpix);    // keep >= minpix pixels per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}

// ---- which kernel, which tile, how many slabs ---------------------------------------------------------------
struct WgradPlan {
    int direct;      // 1: wgrad_direct_kernel<ai, bj>
    int ai, bj;
    int splits;
};

int max_width(int C, int ld, uintptr_t ptr) {
    for (int w = 4; w > 1; w >>= 1)
        if (C % w == 0 && ld % w == 0 && (ptr & (4 * w - 1)) == 0) return w;
    return 1;
}

bool direct_ok(const ds_conv_desc *d, int64_t M) {
    static int mode = -1;
    if (mode < 0) {
        const char *e = ds::tune_env("DS_WGRAD_DIRECT");
        mode = e ? atoi(e) : 1;
    }
    return mode && d->fold_cin == 0 && M >= 512;
}

// Tile shape and slab count by a cycle model of the launch: rounds of resident workgroups x (K steps x MFMA cycles
// + epilogue), plus the split-K reduction's traffic.  Slabs are 1, 2, 4 or a multiple of 8 (XCD-aligned, see the
// kernel); a slab keeps >= 64 pixels per wave.
WgradPlan plan_direct(const ds_conv_desc *d, int64_t M, int wx, int wz, int only_ai = 0, int only_bj = 0) {
    WgradPlan best = {};
    double best_cost = 1e30;
    static const int kSlabs[] = {1, 2, 4, 8, 16, 24, 32, 40, 48, 56, 64};
    const double wbytes = (double)d->KH * d->KW * d->Cin * d->Cout * 4.0;
    for (int ai = 1; ai <= wx; ai *= 2)
        for (int bj = 1; bj <= wz; bj *= 2) {
            if ((only_ai && ai != only_ai) || (only_bj && bj != only_bj)) continue;
            const int nacc = ai * bj;
            const int occ = nacc > 8 ? 1 : (nacc > 4 ? 2 : 3);
            const int tiles = d->KH * d->KW * ((d->Cin + 32 * ai - 1) / (32 * ai)) * ((d->Cout + 32 * bj - 1) / (32 * bj));
            for (int slabs : kSlabs) {
                if (slabs > 1 && M / slabs < 256) break;
                const int xcds = slabs >= 8 ? 8 : slabs;                    // XCDs' worth of slots a slab group spans
                const double per_xcd = (double)tiles * slabs / 8.0;        // workgroups per XCD
                const double slots = 32.0 * occ;
                double rounds = (double)(int64_t)(per_xcd / slots);
                if (rounds * slots < per_xcd) rounds += 1.0;
                const double steps = (double)M / slabs / 8.0;              // K steps per wave
                // narrow blocks issue a load pair per few MFMAs: charge the address path too
                const double step_cycles = nacc * 64.0 > 160.0 ? nacc * 64.0 : 160.0;
                const double t_wg = steps * step_cycles + 1500.0 + nacc * 350.0;
                const double t_red = slabs > 1 ? 6000.0 + slabs * wbytes / 1500.0 : 0.0;
                const double cost = rounds * t_wg + t_red;
                (void)xcds;
                if (cost < best_cost) {
                    best_cost = cost;
                    best.direct = 1; best.ai = ai; best.bj = bj; best.splits = slabs;
                }
            }
        }
    return best;
}

// (Round 6, measured and NOT kept: the two LSTM matrices -- 2048 output columns, every x row block the A operand of sixteen
// column tiles -- run 168 -> 137 us and 222 -> 176 us on 64 x 128 / 128 x 128 tiles, the text-only step 1.06 -> 1.01 ms; but the
// joint step, where these launches sit beside the image tower's backward chain, went 13.12 -> 13.15 ms and 7.80 -> 7.85 at B = 128:
// one or two fat workgroups per CU hold the chain's launches off longer than three thin ones.  profiles/r06_notes.md.)
WgradPlan plan_wgrad(const ds_conv_desc *d, int64_t M, const float *x, const float *dz, int lddz, const float *dw,
                     const void *ws) {
    if (direct_ok(d, M)) {
        // BJ is also the width of the kernel's vector stores into dw (or the split-K workspace): their alignment counts
        const uintptr_t zalign = (uintptr_t)dz | (uintptr_t)dw | (uintptr_t)ws;
        const int wx = max_width(d->Cin, d->ldx, (uintptr_t)x), wz = max_width(d->Cout, lddz, zalign);
        WgradPlan pl = plan_direct(d, M, wx, wz);
        if (const char *e = ds::tune_env("DS_WGRAD_FORCE")) {          // "ai,bj,slabs" (tuning aid; the caller sizes the workspace)
            int a = 0, b = 0, sl = 0;
            if (sscanf(e, "%d,%d,%d", &a, &b, &sl) == 3) {
                if ((a && a <= wx) || (b && b <= wz)) {          // the slab count the model gives THAT tile shape
                    const WgradPlan f = plan_direct(d, M, wx, wz, a <= wx ? a : 0, b <= wz ? b : 0);
                    if (f.direct) pl = f;
                }
                if (sl) pl.splits = sl;
            }
        }
        return pl;
    }
    WgradPlan pl = {};
    pl.splits = pick_splits(d, M);
    return pl;
}

}  // namespace

extern "C" size_t ds_conv_wgrad_workspace(const ds_conv_desc *d) {
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    // the vector widths depend on the run-time pointers and leading dimensions: size for the worst of them
    int splits = pick_splits(d, M);
    if (direct_ok(d, M)) {
        splits = 1;
        for (int wx = 1; wx <= 4; wx *= 2)
            for (int wz = 1; wz <= 4; wz *= 2) {
                const int sp = plan_direct(d, M, wx, wz).splits;
                splits = sp > splits ? sp : splits;
            }
    }
    if (splits == 1) return 0;
    return (size_t)splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
}

extern "C" int ds_conv_wgrad(const ds_conv_desc *d, const float *x, const float *dz, int32_t lddz, float *dw,
                             void *ws, size_t ws_bytes, void *stream) {
    DS_REQUIRE(d && x && dz && dw, "ds_conv_wgrad: null argument");
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    DS_REQUIRE(M < (1ll << 31), "ds_conv_wgrad: M too large");
    DS_REQUIRE(d->fold_cin == 0 || (d->KW == 1 && d->ldx == d->fold_cin && d->fold_cin % 4 == 0 && d->Cin % d->fold_cin == 0),
               "ds_conv_wgrad: fold_cin needs KW=1, ldx==fold_cin, fold_cin %% 4 == 0");
    const WgradPlan pl = plan_wgrad(d, M, x, dz, lddz, dw, ws);
    if (ds::tune_env("DS_WGRAD_DEBUG"))
        fprintf(stderr, "wgrad M=%lld Cin=%d Cout=%d k=%d: direct=%d ai=%d bj=%d slabs=%d\n", (long long)M, d->Cin, d->Cout, d->KH,
                pl.direct, pl.ai, pl.bj, pl.splits);
    const int splits = pl.splits;
    const size_t need = splits == 1 ? 0 : (size_t)splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
    if (need > 0 && (ws == nullptr || ws_bytes < need)) {
        ds::set_error("ds_conv_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
        return DS_ERR_WORKSPACE;
    }
    WgradParams p;
    p.d = *d;
    p.x = x; p.dz = dz; p.lddz = lddz;
    p.out = splits == 1 ? dw : (float *)ws;
    p.M = (int)M;
    const int64_t x_elems = ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin;
    const int64_t z_elems = (M - 1) * lddz + d->Cout;
    DS_REQUIRE(x_elems * 4 < (1ll << 31) && z_elems * 4 < (1ll << 31), "ds_conv_wgrad: operand larger than 2 GiB (split the batch)");
    p.x_bytes = (unsigned)(x_elems * 4);
    p.z_bytes = (unsigned)(z_elems * 4);
    p.inv_ohw = 1.0f / (float)(d->OH * d->OW);
    p.inv_ow = 1.0f / (float)d->OW;
    p.x_vec = (d->ldx % 4 == 0) && (d->Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    p.z_vec = (lddz % 4 == 0) && (d->Cout % 4 == 0) && (((uintptr_t)dz & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    int pps = (int)((M + splits - 1) / splits);
    if (pl.direct) {
        p.ci_tiles = (d->Cin + 32 * pl.ai - 1) / (32 * pl.ai);
        p.pix_per_split = (pps + 7) / 8 * 8;
        const int co_tiles = (d->Cout + 32 * pl.bj - 1) / (32 * pl.bj);
        p.slabs = splits;
        const dim3 grid((unsigned)(d->KH * d->KW * p.ci_tiles * co_tiles * splits));
#define DS_WG(A, B) if (pl.ai == A && pl.bj == B) hipLaunchKernelGGL((wgrad_direct_kernel<A, B>), grid, dim3(256), 0, s, p);
        DS_WG(4, 4) DS_WG(4, 2) DS_WG(4, 1) DS_WG(2, 4) DS_WG(2, 2) DS_WG(2, 1) DS_WG(1, 4) DS_WG(1, 2) DS_WG(1, 1)
#undef DS_WG
    } else {
        const int ti = pick_ti(d);
        p.ci_tiles = (d->Cin + ti - 1) / ti;
        p.pix_per_split = ((pps + TP - 1) / TP) * TP;
        dim3 grid(d->KH * d->KW * p.ci_tiles, (d->Cout + TJ - 1) / TJ, splits);
        if (ti == 64) hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(256), 0, s, p);
    }
    if (splits > 1) {
        const int64_t n = (int64_t)d->KH * d->KW * d->Cin * d->Cout;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ds::stream_grid(n, 256)), dim3(256), 0, s,
                           (const float *)ws, dw, n, splits);
    }
    return ds::check_launch("ds_conv_wgrad");
}
