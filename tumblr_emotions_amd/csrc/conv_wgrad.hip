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
        const char *e = getenv("DS_WGRAD_OCC");
        occ = e ? atoi(e) : 3;          // measured best of 2/3/4/6 workgroups per CU (MI355X, joint step)
        e = getenv("DS_WGRAD_MINPIX");
        minpix = e ? atoi(e) : 64;
    }
    int splits = (occ * ds::kCUs + tiles - 1) / tiles;
    const int max_splits = (int)((M + minpix - 1) / minpix);    // keep >= minpix pixels per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}

}  // namespace

extern "C" size_t ds_conv_wgrad_workspace(const ds_conv_desc *d) {
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    const int splits = pick_splits(d, M);
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
    const int splits = pick_splits(d, M);
    const size_t need = ds_conv_wgrad_workspace(d);
    if (need > 0 && (ws == nullptr || ws_bytes < need)) {
        ds::set_error("ds_conv_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
        return DS_ERR_WORKSPACE;
    }
    WgradParams p;
    p.d = *d;
    p.x = x; p.dz = dz; p.lddz = lddz;
    p.out = splits == 1 ? dw : (float *)ws;
    p.M = (int)M;
    const int ti = pick_ti(d);
    p.ci_tiles = (d->Cin + ti - 1) / ti;
    int pps = (int)((M + splits - 1) / splits);
    p.pix_per_split = ((pps + TP - 1) / TP) * TP;
    p.x_vec = (d->ldx % 4 == 0) && (d->Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    p.z_vec = (lddz % 4 == 0) && (d->Cout % 4 == 0) && (((uintptr_t)dz & 15) == 0);
    const int64_t x_elems = ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin;
    const int64_t z_elems = (M - 1) * lddz + d->Cout;
    DS_REQUIRE(x_elems * 4 < (1ll << 31) && z_elems * 4 < (1ll << 31), "ds_conv_wgrad: operand larger than 2 GiB (split the batch)");
    p.x_bytes = (unsigned)(x_elems * 4);
    p.z_bytes = (unsigned)(z_elems * 4);
    p.inv_ohw = 1.0f / (float)(d->OH * d->OW);
    p.inv_ow = 1.0f / (float)d->OW;
    dim3 grid(d->KH * d->KW * p.ci_tiles, (d->Cout + TJ - 1) / TJ, splits);
    hipStream_t s = (hipStream_t)stream;
    if (ti == 64) hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(256), 0, s, p);
    if (splits > 1) {
        const int64_t n = (int64_t)d->KH * d->KW * d->Cin * d->Cout;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ds::stream_grid(n, 256)), dim3(256), 0, s,
                           (const float *)ws, dw, n, splits);
    }
    return ds::check_launch("ds_conv_wgrad");
}
