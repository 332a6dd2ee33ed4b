// Implicit-GEMM convolution / GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces slim.conv2d's Conv2D, its Conv2DBackpropInput, and every tf.matmul on the Deep
// Sentiment path (reference call sites: see include/ds_kernels.h).  Design notes:
//   * A operand = NHWC activations, gathered tap by tap (no im2col buffer); K-tile = 16
//     channels of one tap, so a row of the A tile is 64 contiguous bytes of one input pixel.
//   * B operand = the TF HWIO weight tensor read in place through (tap, n, k) strides: forward
//     convs see it n-contiguous (transposed into LDS on the fly), dgrad sees it k-contiguous
//     with the tap order flipped.  Nothing is ever re-packed.
//   * 4 waves per workgroup, each owning an (MT*32)x(NT*32) accumulator in AGPR/VGPRs;
//     LDS rows are padded to 20 floats so the ds_read_b128 fragment reads are conflict-free.
//   * workgroups are persistent over row tiles (grid.x is a multiple of 8 so that all column
//     tiles of a row tile, which share the A rows, land on the same XCD/L2), which also lets the
//     BatchNorm column statistics be accumulated in registers and emitted once per workgroup.
//   * fp32 MFMA is an exact k-ordered fmaf chain (cdna_hip_programming.md section 3), so results
//     match a scalar fp32 reference to rounding.
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;    // K-tile (floats)
constexpr int LDK = 20;   // padded LDS row stride (floats): 80 B, keeps b128 reads conflict-free

struct ConvParams {
    ds_conv_desc d;
    const float *x;
    const float *w;
    float *z;
    const float *bias;
    const float *mask;
    float *stats;
    int M;          // N*OH*OW
    int taps;       // KH*KW
    int chunks;     // ceil(Cin/BK)
    int row_tiles;  // ceil(M/BM)
    int a_vec;      // 16-byte vector loads legal on A
    int b_vec;      // ... on B
};

template <int MT, int NT, int WM, int WN, bool BNMAJOR, bool FOLD>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int AR = BM / 64;                       // float4 A loads per thread per K-tile
    constexpr int BR = (BN * 4 + 255) / 256;          // float4 B loads per thread per K-tile
    static_assert(WM * WN == 4, "four waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
    float *As = smem;
    float *Bs = smem + 2 * BM * LDK;

    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lk = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int ohw = d.OH * d.OW;
    const int KT = p.taps * p.chunks;
    const int flags = d.flags;

    int nt_valid = (d.Cout - (n0 + wn * NT * 32) + 31) / 32;
    nt_valid = nt_valid < 0 ? 0 : (nt_valid > NT ? NT : nt_valid);

    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;

    const int arow = tid >> 2, ak4 = (tid & 3) * 4;

    for (int tile = blockIdx.x; tile < p.row_tiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        int ih0[AR], iw0[AR];
        const float *xb[AR];
        bool rv[AR];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int m = m0 + arow + 64 * i;
            rv[i] = m < p.M;
            int mm = rv[i] ? m : 0;
            int n = mm / ohw;
            int r = mm - n * ohw;
            int oh = r / d.OW;
            int ow = r - oh * d.OW;
            ih0[i] = oh * d.stride - d.pad_t;
            iw0[i] = ow * d.stride - d.pad_l;
            xb[i] = p.x + (int64_t)n * d.H * d.W * d.ldx;
        }

        f32x16 acc[MT][NT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        float4 ra[AR], rb[BR];
        int tap = 0, c0 = 0, dh = 0, dw = 0;   // K-tile that the next load_tile() fetches

        auto load_tile = [&]() {
            // ---- A: activations ------------------------------------------------------------
            const int k = c0 + ak4;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int ih = ih0[i] + dh;
                const int iw = iw0[i] + dw;
                const int iwc = FOLD ? iw + k / d.fold_cin : iw;
                const bool ok = rv[i] && (unsigned)ih < (unsigned)d.H && (unsigned)iwc < (unsigned)d.W;
                const float *ptr = xb[i] + ((int64_t)ih * d.W + iw) * d.ldx + k;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    if (p.a_vec) {
                        if (k < d.Cin) v = *reinterpret_cast<const float4 *>(ptr);
                    } else {
                        if (k + 0 < d.Cin) v.x = ptr[0];
                        if (k + 1 < d.Cin) v.y = ptr[1];
                        if (k + 2 < d.Cin) v.z = ptr[2];
                        if (k + 3 < d.Cin) v.w = ptr[3];
                    }
                }
                ra[i] = v;
            }
            // ---- B: weights, read in place from the HWIO tensor --------------------------------
            const int tap_eff = d.flip ? p.taps - 1 - tap : tap;
            const float *wt = p.w + (int64_t)tap_eff * d.w_tap_stride;
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int idx = tid + 256 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < BN * 4) {
                    if (BNMAJOR) {   // n contiguous (forward): float4 spans 4 output channels
                        const int n4 = idx % (BN / 4), kk = c0 + idx / (BN / 4);
                        const int nn = n0 + n4 * 4;
                        if (kk < d.Cin) {
                            const float *ptr = wt + (int64_t)kk * d.w_k_stride + nn;
                            if (p.b_vec) {
                                if (nn < d.Cout) v = *reinterpret_cast<const float4 *>(ptr);
                            } else {
                                if (nn + 0 < d.Cout) v.x = ptr[0];
                                if (nn + 1 < d.Cout) v.y = ptr[1];
                                if (nn + 2 < d.Cout) v.z = ptr[2];
                                if (nn + 3 < d.Cout) v.w = ptr[3];
                            }
                        }
                    } else {         // k contiguous (dgrad / transposed matmul)
                        const int nn = n0 + (idx >> 2), kk = c0 + (idx & 3) * 4;
                        if (nn < d.Cout) {
                            const float *ptr = wt + (int64_t)nn * d.w_n_stride + kk;
                            if (p.b_vec) {
                                if (kk < d.Cin) v = *reinterpret_cast<const float4 *>(ptr);
                            } else {
                                if (kk + 0 < d.Cin) v.x = ptr[0];
                                if (kk + 1 < d.Cin) v.y = ptr[1];
                                if (kk + 2 < d.Cin) v.z = ptr[2];
                                if (kk + 3 < d.Cin) v.w = ptr[3];
                            }
                        }
                    }
                }
                rb[i] = v;
            }
            // ---- advance to the next K-tile ----------------------------------------------------
            c0 += BK;
            if (c0 >= d.Cin) {
                c0 = 0;
                ++tap;
                if (++dw == d.KW) { dw = 0; ++dh; }
            }
        };

        auto store_tile = [&](int buf) {
            float *a_s = As + buf * BM * LDK;
            float *b_s = Bs + buf * BN * LDK;
#pragma unroll
            for (int i = 0; i < AR; ++i)
                *reinterpret_cast<float4 *>(a_s + (arow + 64 * i) * LDK + ak4) = ra[i];
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int idx = tid + 256 * i;
                if (idx < BN * 4) {
                    if (BNMAJOR) {
                        const int n4 = idx % (BN / 4), kk = idx / (BN / 4);
                        float *q = b_s + (n4 * 4) * LDK + kk;
                        q[0] = rb[i].x;
                        q[LDK] = rb[i].y;
                        q[2 * LDK] = rb[i].z;
                        q[3 * LDK] = rb[i].w;
                    } else {
                        *reinterpret_cast<float4 *>(b_s + (idx >> 2) * LDK + (idx & 3) * 4) = rb[i];
                    }
                }
            }
        };

        auto compute = [&](int buf) {
            const float *a_s = As + buf * BM * LDK + (wm * MT * 32 + li) * LDK + lk * 8;
            const float *b_s = Bs + buf * BN * LDK + (wn * NT * 32 + li) * LDK + lk * 8;
            float af[MT][8], bf[NT][8];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const float4 lo = *reinterpret_cast<const float4 *>(a_s + a * 32 * LDK);
                const float4 hi = *reinterpret_cast<const float4 *>(a_s + a * 32 * LDK + 4);
                af[a][0] = lo.x; af[a][1] = lo.y; af[a][2] = lo.z; af[a][3] = lo.w;
                af[a][4] = hi.x; af[a][5] = hi.y; af[a][6] = hi.z; af[a][7] = hi.w;
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const float4 lo = *reinterpret_cast<const float4 *>(b_s + b * 32 * LDK);
                const float4 hi = *reinterpret_cast<const float4 *>(b_s + b * 32 * LDK + 4);
                bf[b][0] = lo.x; bf[b][1] = lo.y; bf[b][2] = lo.z; bf[b][3] = lo.w;
                bf[b][4] = hi.x; bf[b][5] = hi.y; bf[b][6] = hi.z; bf[b][7] = hi.w;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        if (b < nt_valid)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
        };

        load_tile();
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_tile();
            compute(cur);
            if (kt + 1 < KT) store_tile(cur ^ 1);
            __syncthreads();
        }

        // ---- epilogue: bias / accumulate / mask / relu, store, BatchNorm column statistics -------
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            if (b < nt_valid) {
                const int col = n0 + wn * NT * 32 + b * 32 + li;
                const bool colok = col < d.Cout;
                const float bv = ((flags & DS_EPI_BIAS) && colok) ? p.bias[col] : 0.f;
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int a = 0; a < MT; ++a) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (row < p.M && colok) {
                            float v = acc[a][b][r] + bv;
                            const int64_t off = (int64_t)row * d.ldz + col;
                            if (flags & DS_EPI_ACCUM) v += p.z[off];
                            if (flags & DS_EPI_MASK) v = p.mask[(int64_t)row * d.ldmask + col] > 0.f ? v : 0.f;
                            if (flags & DS_EPI_RELU) v = fmaxf(v, 0.f);
                            p.z[off] = v;
                            s += v;
                            q += v * v;
                        }
                    }
                }
                csum[b] += s;
                csq[b] += q;
            }
        }
    }

    if (flags & DS_EPI_STATS) {
        // rows of a column live in lanes l and l^32, and in the WM waves stacked along M
        float *red = smem;   // [WM][BN][2]; safe: every wave passed the last K-loop barrier
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            float s = csum[b] + __shfl_xor(csum[b], 32);
            float q = csq[b] + __shfl_xor(csq[b], 32);
            if (lk == 0) {
                const int c = wn * NT * 32 + b * 32 + li;
                red[(wm * BN + c) * 2 + 0] = s;
                red[(wm * BN + c) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.Cout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s += red[(w * BN + tid) * 2 + 0];
                q += red[(w * BN + tid) * 2 + 1];
            }
            float *o = p.stats + (int64_t)blockIdx.x * 2 * d.Cout;
            o[n0 + tid] = s;
            o[d.Cout + n0 + tid] = q;
        }
    }
}

// ---- host-side dispatch -----------------------------------------------------------------------
struct TileCfg {
    int bm, bn;
};

enum CfgId { CFG_128x128 = 0, CFG_256x64, CFG_256x32, CFG_256x96, CFG_64x128, CFG_64x64, CFG_COUNT };
const TileCfg kCfg[CFG_COUNT] = {{128, 128}, {256, 64}, {256, 32}, {256, 96}, {64, 128}, {64, 64}};

int64_t conv_M(const ds_conv_desc *d) { return (int64_t)d->N * d->OH * d->OW; }

CfgId pick_cfg(const ds_conv_desc *d) {
    const int64_t M = conv_M(d);
    CfgId c;
    if (d->Cout <= 32) c = CFG_256x32;
    else if (d->Cout <= 64) c = CFG_256x64;
    else if (d->Cout <= 96) c = CFG_256x96;
    else c = CFG_128x128;
    // small problems: prefer more, smaller workgroups so that all 256 CUs get work
    auto blocks = [&](CfgId k) { return ((M + kCfg[k].bm - 1) / kCfg[k].bm) * ((d->Cout + kCfg[k].bn - 1) / kCfg[k].bn); };
    if (blocks(c) < 2 * ds::kCUs) {
        if (d->Cout > 64 && blocks(CFG_64x128) >= blocks(c)) c = CFG_64x128;
        if (blocks(c) < ds::kCUs) c = CFG_64x64;
    }
    return c;
}

void grid_for(const ds_conv_desc *d, CfgId c, int *gx, int *gy, int *row_tiles) {
    const int64_t M = conv_M(d);
    *row_tiles = (int)((M + kCfg[c].bm - 1) / kCfg[c].bm);
    *gy = (d->Cout + kCfg[c].bn - 1) / kCfg[c].bn;
    int target = (3 * ds::kCUs) / *gy;       // ~3 resident workgroups per CU in total
    if (target < 8) target = 8;
    int x = *row_tiles < target ? *row_tiles : target;
    if (x >= 8) x &= ~7;                      // multiple of 8: column tiles of a row tile share an XCD
    *gx = x;
}

template <int MT, int NT, int WM, int WN>
void launch_cfg(const ConvParams &p, dim3 grid, hipStream_t s, bool bnmajor, bool fold) {
    if (bnmajor) {
        if (fold) hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, true, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, true, false>), grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, false, false>), grid, dim3(256), 0, s, p);
    }
}

}  // namespace

extern "C" int ds_conv_igemm_partials(const ds_conv_desc *d) {
    int gx, gy, rt;
    grid_for(d, pick_cfg(d), &gx, &gy, &rt);
    return gx;
}

extern "C" int ds_conv_igemm(const ds_conv_desc *d, const float *x, const float *w, float *z, const float *bias,
                             const float *mask, float *stats, void *stream) {
    DS_REQUIRE(d && x && w && z, "ds_conv_igemm: null argument");
    DS_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->N > 0 && d->OH > 0 && d->OW > 0, "ds_conv_igemm: bad dims");
    DS_REQUIRE(d->w_n_stride == 1 || d->w_k_stride == 1, "ds_conv_igemm: one weight stride must be 1");
    DS_REQUIRE(!(d->flags & DS_EPI_BIAS) || bias, "ds_conv_igemm: DS_EPI_BIAS without bias");
    DS_REQUIRE(!(d->flags & DS_EPI_MASK) || mask, "ds_conv_igemm: DS_EPI_MASK without mask");
    DS_REQUIRE(!(d->flags & DS_EPI_STATS) || stats, "ds_conv_igemm: DS_EPI_STATS without stats buffer");
    DS_REQUIRE(conv_M(d) < (1ll << 31), "ds_conv_igemm: M too large");
    const bool bnmajor = d->w_n_stride == 1 && d->w_k_stride != 1;
    const bool fold = d->fold_cin > 0;
    DS_REQUIRE(!fold || (bnmajor && d->KW == 1 && d->fold_cin % 4 == 0 && d->ldx == d->fold_cin),
               "ds_conv_igemm: fold_cin needs KW=1, n-contiguous weights, ldx==fold_cin");

    ConvParams p;
    p.d = *d;
    p.x = x; p.w = w; p.z = z; p.bias = bias; p.mask = mask; p.stats = stats;
    p.M = (int)conv_M(d);
    p.taps = d->KH * d->KW;
    p.chunks = (d->Cin + BK - 1) / BK;
    p.a_vec = (d->ldx % 4 == 0) && (d->Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    if (bnmajor)
        p.b_vec = (d->Cout % 4 == 0) && (d->w_k_stride % 4 == 0) && (d->w_tap_stride % 4 == 0) && (((uintptr_t)w & 15) == 0);
    else
        p.b_vec = (d->Cin % 4 == 0) && (d->w_n_stride % 4 == 0) && (d->w_tap_stride % 4 == 0) && (((uintptr_t)w & 15) == 0);

    const CfgId c = pick_cfg(d);
    int gx, gy, rt;
    grid_for(d, c, &gx, &gy, &rt);
    p.row_tiles = rt;
    dim3 grid(gx, gy);
    hipStream_t s = (hipStream_t)stream;
    switch (c) {
        case CFG_128x128: launch_cfg<2, 2, 2, 2>(p, grid, s, bnmajor, fold); break;
        case CFG_256x64: launch_cfg<2, 2, 4, 1>(p, grid, s, bnmajor, fold); break;
        case CFG_256x32: launch_cfg<2, 1, 4, 1>(p, grid, s, bnmajor, fold); break;
        case CFG_256x96: launch_cfg<2, 3, 4, 1>(p, grid, s, bnmajor, fold); break;
        case CFG_64x128: launch_cfg<1, 2, 2, 2>(p, grid, s, bnmajor, fold); break;
        default: launch_cfg<1, 1, 2, 2>(p, grid, s, bnmajor, fold); break;
    }
    return ds::check_launch("ds_conv_igemm");
}
