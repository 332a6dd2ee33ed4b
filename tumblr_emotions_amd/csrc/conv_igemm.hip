// Implicit-GEMM convolution / GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces slim.conv2d's Conv2D, its Conv2DBackpropInput, and every tf.matmul on the Deep
// Sentiment path (reference call sites: see include/ds_kernels.h).  Design notes:
//   * A operand = NHWC activations, gathered tap by tap (no im2col buffer); K-tile = 16
//     channels of one tap, so a row of the A tile is 64 contiguous bytes of one input pixel.
//   * B operand = the TF HWIO weight tensor read in place through (tap, n, k) strides: forward
//     convs see it n-contiguous (staged [k][n] in LDS, fragments read with conflict-free
//     ds_read_b32), dgrad sees it k-contiguous with the tap order flipped (staged [n][k], fragments
//     read with ds_read_b128 like A).  Nothing is ever re-packed.
//   * all global reads are SRD buffer loads (buffer_load_dwordx4 ... offen): padding pixels, the K
//     tail and rows/columns past the tensor get an out-of-range offset and the hardware returns
//     zeros -- no branches and no selects in the main loop, so a K-tile's loads issue back to back,
//     stay in flight under the MFMAs of the current tile and are waited for only at the LDS store.
//   * 4 waves per workgroup stacked along M; each wave owns (MT*32) rows x all BN = NT*32 columns,
//     so the tile width is chosen per layer from NT = 1..6 to fit Cout with little padding.
//   * launch geometry (grid_for): one workgroup per row tile up to 2048 row tiles; above that the
//     workgroups are persistent over row tiles (grid.x a multiple of 8 so that all column tiles of
//     a row tile, which share the A rows, land on the same XCD/L2) with the K-loop pipeline running
//     across tiles.  BatchNorm column statistics are accumulated in registers and emitted once per
//     workgroup either way.
//   * fp32 MFMA is an exact k-ordered fmaf chain (cdna_hip_programming.md section 3), so results
//     match a scalar fp32 reference to rounding.
#include <stdlib.h>
#include <type_traits>
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector (HIP's float4 is a struct)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

namespace {

constexpr int BK = 16;    // K-tile (floats)
constexpr int LDK = 20;   // padded LDS row stride for [row][k] tiles: 80 B, conflict-free b128 reads
constexpr unsigned kOOB = 0x80000000u;   // byte offset beyond any descriptor: the load returns 0

struct ConvParams {
    ds_conv_desc d;
    const float *x;
    const float *w;
    float *z;
    const float *bias;
    const float *mask;
    float *stats;
    const float *pivot;   // per-column shift of the statistics (nullable)
    int M;          // N*OH*OW
    int taps;       // KH*KW
    int row_tiles;  // ceil(M/BM)
    unsigned x_bytes, w_bytes;   // extents covered by the two buffer descriptors
    int prio_mode;               // 1: staggered static wave priorities (see kernel)
    int col_tiles;               // > 0: 1-D XCD-aware launch (see TileId); 0: (row, column) = (blockIdx.x, blockIdx.y)
    int col_total;               // conv_bf16d_kernel: columns of the converted weight tensor (Cout rounded up to 32)
    ds_bn_bwd_on_load bnb;       // gemm_wide_kernel<.., BNB = true>: copy of *d.bnb (the descriptor's pointer is a host pointer)
    int pool_rpb, pool_bpi;      // gemm_wide_kernel<.., POOL = true>: image rows per 32-pixel block, blocks per image
    ds_bn_finalize_in_launch fin;      // gemm_wide_kernel: copy of *d.fin (fin.ticket == nullptr: the caller finalizes)
    double fin_inv_count;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// Which tile does this workgroup own?  Two launch shapes:
//   * col_tiles == 0: 2-D grid, blockIdx.x = first row tile (persistent stride gridDim.x), blockIdx.y = column
//     tile; grid.x is a multiple of 8, so the column tiles of a row tile share an XCD.
//   * col_tiles > 0 (one workgroup per tile): 1-D grid, XCD-aware.  Workgroup id lands on XCD id % 8
//     (MI355X_MICROARCH.md, observed placement -- a different one only costs speed).  The row-major list of
//     (row tile, column tile) pairs is cut into 8 equal contiguous ranges, one per XCD, and consecutive ids of
//     ONE XCD walk its range: the workgroups that read the same A rows run on the same XCD back to back, so the
//     rows cross the fabric once per row tile instead of once per column tile (the 1x1 dgrad read its operand
//     4.6x, profiles/r01h_pmc_traffic.txt), and every XCD gets the same number of tiles whatever the tile
//     counts are (2 row tiles x 64 column tiles for an LSTM step included).  Up to 7 surplus workgroups find
//     row >= row_tiles and do nothing.
struct TileId {
    int row, col, stride;      // first row tile, column tile, row-tile stride of the persistent loop
};
__device__ __forceinline__ TileId tile_id(const ConvParams &p) {
    TileId t;
    if (p.col_tiles > 0) {
        const int id = blockIdx.x;
        const int lin = (id & 7) * (int)(gridDim.x >> 3) + (id >> 3);      // position in the row-major tile list
        t.row = lin / p.col_tiles;
        t.col = lin - t.row * p.col_tiles;
        t.stride = p.row_tiles;            // exactly one tile per workgroup
    } else {
        t.row = blockIdx.x;
        t.col = blockIdx.y;
        t.stride = gridDim.x;
    }
    return t;
}

template <bool VEC>
__device__ __forceinline__ f32x4 load4(__amdgpu_buffer_rsrc_t r, unsigned off, bool ok, int valid) {
    // off: byte offset of 4 consecutive floats; `valid` (<= 4) of them exist (scalar path only)
    if (VEC) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : kOOB, 0, 0));
    } else {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (ok && j < valid) ? off + 4 * j : kOOB, 0, 0));
        return v;
    }
}

// second launch-bound = waves per SIMD the register allocator must leave room for: the small tiles
// are the workhorses and measured fastest at 3-4 resident workgroups per CU
template <int MT, int NT, bool BNMAJOR, bool FOLD, bool VEC>
__global__ __launch_bounds__(256, (MT == 1 && NT == 1) ? 4 : (MT == 1 && NT == 2) ? 3 : 1) void conv_igemm_kernel(
    const ConvParams p) {
    constexpr int WM = 4;
    constexpr int BM = WM * MT * 32, BN = NT * 32;
    constexpr int AR = BM / 64;                       // float4 A loads per thread per K-tile
    constexpr int BR = (BN * 4 + 255) / 256;          // float4 B loads per thread per K-tile
    constexpr int LDB = BNMAJOR ? (BN + 4) : LDK;     // B tile: [k][n] (n-contiguous weights) or [n][k]
    constexpr int BSZ = BNMAJOR ? BK * LDB : BN * LDK;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK + 2 * BSZ];
    float *As = smem;
    float *Bs = smem + 2 * BM * LDK;

    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int li = lane & 31, lk = lane >> 5;
    const TileId tid0 = tile_id(p);
    const int n0 = tid0.col * BN;
    const int ohw = d.OH * d.OW;
    const int chunks = (d.Cin + BK - 1) / BK;
    // split-K (small-M GEMMs): blockIdx.z owns K-tiles [kt0, kt1) and writes its own output slab
    const int KT_all = p.taps * chunks;
    const int kts = (KT_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int kt0 = (int)blockIdx.z * kts;
    const int KT = (kt0 + kts < KT_all ? kt0 + kts : KT_all) - kt0;     // K-tiles of this split (may be <= 0)
    float *const zout = p.z + (int64_t)blockIdx.z * p.d.z_split_stride;
    const int flags = d.flags;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);

    // De-convoy the waves that share a SIMD.  With equal priority the matrix pipe is arbitrated fairly
    // among the 3-4 resident waves (one per co-resident workgroup), so they all finish their MFMA burst
    // together, all do their address/LDS/barrier phase together, and the pipe idles for that whole phase
    // (PMC: MFMA busy 72 %, idle share = non-MFMA phase / K-step).  Distinct static priorities make the
    // arbitration unfair: the phases stagger and one wave's non-MFMA work hides under another's MFMAs.
    // Co-resident workgroups are ~256 apart in dispatch order (id % 8 -> XCD, (id / 8) % 32 -> CU); a
    // different placement only costs speed.  s_setprio is scalar: the switch is wave-uniform.
    switch (p.prio_mode ? ((blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) >> 8) & 3 : 0) {
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
    }

    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;

    const int arow = tid >> 2, ak4 = (tid & 3) * 4;

    // B-tile coordinates of this thread's loads are the same for every row tile and K-tile
    int b_n[BR], b_k[BR];
    unsigned b_off[BR];      // element offset inside one tap, without the K-tile base
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int idx = tid + 256 * i;
        if (BNMAJOR) {       // n contiguous (forward): a float4 spans 4 output channels
            b_n[i] = n0 + (idx % (BN / 4)) * 4;
            b_k[i] = idx / (BN / 4);
            b_off[i] = (unsigned)b_k[i] * (unsigned)d.w_k_stride + (unsigned)b_n[i];
        } else {             // k contiguous (dgrad / transposed matmul): a float4 spans 4 reduction channels
            b_n[i] = n0 + (idx >> 2);
            b_k[i] = (idx & 3) * 4;
            b_off[i] = (unsigned)b_n[i] * (unsigned)d.w_n_stride + (unsigned)b_k[i];
        }
        b_ok[i] = idx < BN * 4 && b_n[i] < d.Cout;
    }

    // ---- loader state: the row tile and K-tile that the next load_tile() fetches ------------------
    int ih0[AR], iw0[AR];
    unsigned xb[AR];         // element offset of the image that row i belongs to
    int tap, c0, dh, dw;
    f32x4 ra[AR], rb[BR];
    f32x16 acc[MT][NT];

    auto setup_tile = [&](int tile) {      // point the loader at the first K-tile of row tile `tile`
        const int m0 = tile * BM;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + arow + 64 * i;
            const bool rv = m < p.M;
            const int mm = rv ? m : 0;
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oh = r / d.OW;
            const int ow = r - oh * d.OW;
            ih0[i] = rv ? oh * d.stride - d.pad_t : -(1 << 20);   // invalid rows fail the bounds test below
            iw0[i] = ow * d.stride - d.pad_l;
            xb[i] = (unsigned)n * (unsigned)(d.H * d.W) * (unsigned)d.ldx;
        }
        // K-tile order: channel chunk OUTER, tap INNER -- the KH*KW taps of one 16-channel chunk touch the
        // same (tile + halo) pixels' 64-byte segments, ~15 KB per workgroup, which stays in the XCD's L2
        // (tap-outer order streamed 9 x the whole tile through L2: measured 8x over-fetch on the 3x3 layers)
        const int chunk = kt0 / p.taps;
        tap = kt0 - chunk * p.taps;
        c0 = chunk * BK;
        dh = tap / d.KW;
        dw = tap - dh * d.KW;
    };

    {

        auto load_tile = [&]() {
            // ---- A: activations ----------------------------------------------------------------
            const int k = c0 + ak4;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int ih = ih0[i] + dh;
                const int iw = iw0[i] + dw;
                const int iwc = FOLD ? iw + k / d.fold_cin : iw;
                const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iwc < (unsigned)d.W && k < d.Cin;
                const unsigned off = xb[i] + (unsigned)(ih * d.W + iw) * (unsigned)d.ldx + (unsigned)k;
                ra[i] = load4<VEC>(srd_x, off * 4u, ok, d.Cin - k);
            }
            // ---- B: weights, read in place from the HWIO tensor --------------------------------
            const int tap_eff = d.flip ? p.taps - 1 - tap : tap;
            const unsigned wt = (unsigned)tap_eff * (unsigned)d.w_tap_stride +
                                (unsigned)c0 * (BNMAJOR ? (unsigned)d.w_k_stride : 1u);
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const bool ok = b_ok[i] && c0 + b_k[i] < d.Cin;
                rb[i] = load4<VEC>(srd_w, (wt + b_off[i]) * 4u, ok, BNMAJOR ? d.Cout - b_n[i] : d.Cin - c0 - b_k[i]);
            }
            // ---- advance to the next K-tile: next tap of the same channel chunk, then the next chunk ----
            ++tap;
            if (++dw == d.KW) {
                dw = 0;
                if (++dh == d.KH) { dh = 0; tap = 0; c0 += BK; }
            }
        };

        auto store_tile = [&](int buf) {
            float *a_s = As + buf * BM * LDK;
            float *b_s = Bs + buf * BSZ;
#pragma unroll
            for (int i = 0; i < AR; ++i)
                *reinterpret_cast<f32x4 *>(a_s + (arow + 64 * i) * LDK + ak4) = ra[i];
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int idx = tid + 256 * i;
                if (idx < BN * 4) {
                    if (BNMAJOR)
                        *reinterpret_cast<f32x4 *>(b_s + (idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4) = rb[i];
                    else
                        *reinterpret_cast<f32x4 *>(b_s + (idx >> 2) * LDK + (idx & 3) * 4) = rb[i];
                }
            }
        };

        auto compute = [&](int buf) {
            const float *a_s = As + buf * BM * LDK + (wm * MT * 32 + li) * LDK + lk * 8;
            float af[MT][8], bf[NT][8];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(a_s + a * 32 * LDK);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(a_s + a * 32 * LDK + 4);
                af[a][0] = lo.x; af[a][1] = lo.y; af[a][2] = lo.z; af[a][3] = lo.w;
                af[a][4] = hi.x; af[a][5] = hi.y; af[a][6] = hi.z; af[a][7] = hi.w;
            }
            if (BNMAJOR) {      // B staged [k][n]: lane (li, lk) needs B[k = lk*8+s][n = b*32+li]
                const float *b_s = Bs + buf * BSZ + (lk * 8) * LDB + li;
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int s = 0; s < 8; ++s) bf[b][s] = b_s[s * LDB + b * 32];
            } else {
                const float *b_s = Bs + buf * BSZ + li * LDK + lk * 8;
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(b_s + b * 32 * LDK);
                    const f32x4 hi = *reinterpret_cast<const f32x4 *>(b_s + b * 32 * LDK + 4);
                    bf[b][0] = lo.x; bf[b][1] = lo.y; bf[b][2] = lo.z; bf[b][3] = lo.w;
                    bf[b][4] = hi.x; bf[b][5] = hi.y; bf[b][6] = hi.z; bf[b][7] = hi.w;
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
        };

    // One continuous software pipeline over (row tile, K-tile): while the MFMAs of K-tile kt run, the
    // loads of kt+1 are in flight; on the LAST K-tile of a row tile the loads already belong to the
    // first K-tile of the NEXT row tile, so the pipeline never drains between tiles and the epilogue
    // of tile t starts with tile t+1's operands already staged in LDS.
    int par = 0;                                   // LDS buffer holding the K-tile about to be computed
    if (tid0.row < p.row_tiles) {
        setup_tile(tid0.row);
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    for (int tile = tid0.row; tile < p.row_tiles; tile += tid0.stride) {
        const int m0 = tile * BM;
        const bool has_next = tile + tid0.stride < p.row_tiles;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const bool last = kt + 1 == KT;
            if (last && has_next) setup_tile(tile + tid0.stride);
            const bool fetch = !last || has_next;
            if (fetch) load_tile();
            compute(par);
            if (fetch) store_tile(par ^ 1);
            __syncthreads();
            par ^= 1;
        }

        // ---- epilogue: bias / accumulate / mask / relu, store, BatchNorm column statistics -------
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int col = n0 + b * 32 + li;
            const bool colok = col < d.Cout;
            const float bv = ((flags & DS_EPI_BIAS) && colok) ? p.bias[col] : 0.f;
            const float pv = (p.pivot && colok) ? p.pivot[col] : 0.f;      // statistics are taken about the pivot
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < p.M && colok) {
                        float v = acc[a][b][r] + bv;
                        const int64_t off = (int64_t)row * d.ldz + col;
                        if (flags & DS_EPI_ACCUM) v += zout[off];
                        if (flags & DS_EPI_MASK) v = p.mask[(int64_t)row * d.ldmask + col] > 0.f ? v : 0.f;
                        if (flags & DS_EPI_RELU) v = fmaxf(v, 0.f);
                        zout[off] = v;
                        const float u = v - pv;
                        s += u;
                        q += u * u;
                    }
                }
            }
            csum[b] += s;
            csq[b] += q;
        }
    }
    }

    if (flags & DS_EPI_STATS) {
        // rows of a column live in lanes l and l^32, and in the 4 waves stacked along M
        float *red = smem;   // [WM][BN][2]; safe: every wave passed the last K-loop barrier
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const float s = csum[b] + __shfl_xor(csum[b], 32);
            const float q = csq[b] + __shfl_xor(csq[b], 32);
            if (lk == 0) {
                const int c = b * 32 + li;
                red[(wm * BN + c) * 2 + 0] = s;
                red[(wm * BN + c) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.Cout && tid0.row < p.row_tiles) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s += red[(w * BN + tid) * 2 + 0];
                q += red[(w * BN + tid) * 2 + 1];
            }
            // partials laid out [2][Cout][P] (P = workgroups per column tile) so ds_bn_finalize reads them contiguously
            p.stats[(int64_t)(n0 + tid) * tid0.stride + tid0.row] = s;
            p.stats[((int64_t)d.Cout + n0 + tid) * tid0.stride + tid0.row] = q;
        }
    }
}

// ================================================================================================
// Register-direct variant: no LDS, no barriers.
//
// In the kernel above a wave only ever reads ITS OWN 32*MT rows of the A tile back from LDS, and the
// fp32 MFMA fragment layout (lane (i, kh) holds A[i][kh*8 .. kh*8+7]) is exactly two 16-byte loads of
// a row's K-chunk -- so each wave can fetch its A fragments straight from global memory into the
// registers the MFMA reads, and do the same for the weight fragments (two float4 per row when the
// weights are k-contiguous, eight coalesced dwords when they are n-contiguous; co-resident waves hit
// the same weight lines in L1/L2).  That removes every ds_read/ds_write and, more importantly, every
// workgroup barrier: the four waves of a workgroup become independent pipelines (load K-tile t+1 ->
// MFMAs of K-tile t), so a wave delayed by matrix-pipe contention on its SIMD no longer stalls its
// three block-mates on the other SIMDs.  Registers double-buffer one K-tile; the loop is unrolled by
// two so both buffers are statically indexed.
// ================================================================================================
template <int MT, int NT, bool BNMAJOR>
struct Frag {
    f32x4 alo[MT], ahi[MT];                  // A[row][kh*8 + 0..3], [.. + 4..7]
    f32x4 blo[BNMAJOR ? 1 : NT], bhi[BNMAJOR ? 1 : NT];
    float bs[BNMAJOR ? NT : 1][8];           // n-contiguous weights: B[kh*8 + s][col]
};

template <int MT, int NT, bool BNMAJOR, bool FOLD>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p) {
    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int n0 = blockIdx.y * BN;
    const int ohw = d.OH * d.OW;
    const int chunks = (d.Cin + BK - 1) / BK;
    const int KT_all = p.taps * chunks;
    const int kts = (KT_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int kt0 = (int)blockIdx.z * kts;
    const int KT = (kt0 + kts < KT_all ? kt0 + kts : KT_all) - kt0;
    float *const zout = p.z + (int64_t)blockIdx.z * p.d.z_split_stride;
    const int flags = d.flags;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);
    const int wave_id = blockIdx.x * 4 + wave, waves = gridDim.x * 4;

    unsigned b_off[NT];
    bool b_ok[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + b * 32 + li;
        b_ok[b] = n < d.Cout;
        b_off[b] = BNMAJOR ? (unsigned)n + (unsigned)(lk * 8) * (unsigned)d.w_k_stride
                           : (unsigned)n * (unsigned)d.w_n_stride + (unsigned)(lk * 8);
    }

    int ih0[MT], iw0[MT];
    unsigned xb[MT];
    int tap, c0, dh, dw;
    auto setup_tile = [&](int tile) {
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = tile * BM + a * 32 + li;
            const bool rv = m < p.M;
            const int mm = rv ? m : 0;
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oh = r / d.OW;
            const int ow = r - oh * d.OW;
            ih0[a] = rv ? oh * d.stride - d.pad_t : -(1 << 20);
            iw0[a] = ow * d.stride - d.pad_l;
            xb[a] = (unsigned)n * (unsigned)(d.H * d.W) * (unsigned)d.ldx;
        }
        // K-tile order: channel chunk OUTER, tap INNER -- the KH*KW taps of one 16-channel chunk touch the
        // same (tile + halo) pixels' 64-byte segments, ~15 KB per workgroup, which stays in the XCD's L2
        // (tap-outer order streamed 9 x the whole tile through L2: measured 8x over-fetch on the 3x3 layers)
        const int chunk = kt0 / p.taps;
        tap = kt0 - chunk * p.taps;
        c0 = chunk * BK;
        dh = tap / d.KW;
        dw = tap - dh * d.KW;
    };

    typedef Frag<MT, NT, BNMAJOR> F;
    auto load = [&](F &f) {
        const int k = c0 + lk * 8;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int ih = ih0[a] + dh, iw = iw0[a] + dw;
            const bool pix = (unsigned)ih < (unsigned)d.H;
            const bool ok0 = pix && (unsigned)(FOLD ? iw + k / d.fold_cin : iw) < (unsigned)d.W && k < d.Cin;
            const bool ok1 = pix && (unsigned)(FOLD ? iw + (k + 4) / d.fold_cin : iw) < (unsigned)d.W && k + 4 < d.Cin;
            const unsigned off = (xb[a] + (unsigned)(ih * d.W + iw) * (unsigned)d.ldx + (unsigned)k) * 4u;
            f.alo[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, ok0 ? off : kOOB, 0, 0));
            f.ahi[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, ok1 ? off + 16u : kOOB, 0, 0));
        }
        const int tap_eff = d.flip ? p.taps - 1 - tap : tap;
        const unsigned wt = (unsigned)tap_eff * (unsigned)d.w_tap_stride + (unsigned)c0 * (BNMAJOR ? (unsigned)d.w_k_stride : 1u);
        if (BNMAJOR) {
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bool ok = b_ok[b] && k + s < d.Cin;
                    const unsigned off = (wt + b_off[b] + (unsigned)s * (unsigned)d.w_k_stride) * 4u;
                    f.bs[b][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_w, ok ? off : kOOB, 0, 0));
                }
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const unsigned off = (wt + b_off[b]) * 4u;
                f.blo[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, (b_ok[b] && k < d.Cin) ? off : kOOB, 0, 0));
                f.bhi[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, (b_ok[b] && k + 4 < d.Cin) ? off + 16u : kOOB, 0, 0));
            }
        }
        ++tap;
        if (++dw == d.KW) {
            dw = 0;
            if (++dh == d.KH) { dh = 0; tap = 0; c0 += BK; }
        }
    };

    f32x16 acc[MT][NT];
    auto mma = [&](const F &f) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const float av = s < 4 ? f.alo[a][s & 3] : f.ahi[a][s & 3];
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const float bv = BNMAJOR ? f.bs[b][s] : (s < 4 ? f.blo[b][s & 3] : f.bhi[b][s & 3]);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                }
            }
    };

    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;

    F f0, f1;
    if (wave_id < p.row_tiles) {
        setup_tile(wave_id);
        load(f0);
    }
    for (int tile = wave_id; tile < p.row_tiles; tile += waves) {
        const int m0 = tile * BM;
        const bool has_next = tile + waves < p.row_tiles;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            load(f1);                                   // K-tile kt+1 (always inside this row tile)
            mma(f0);
            if (kt + 2 < KT) load(f0);                  // K-tile kt+2 ...
            else if (has_next) { setup_tile(tile + waves); load(f0); }   // ... or the next row tile's first
            mma(f1);
        }
        const bool odd = kt < KT;
        if (odd) {                                      // odd K-tile count: one more step out of f0
            if (has_next) { setup_tile(tile + waves); load(f1); }
            mma(f0);
        }

        // ---- epilogue (same contract as the LDS kernel) -----------------------------------------------
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int col = n0 + b * 32 + li;
            const bool colok = col < d.Cout;
            const float bv = ((flags & DS_EPI_BIAS) && colok) ? p.bias[col] : 0.f;
            const float pv = (p.pivot && colok) ? p.pivot[col] : 0.f;      // statistics are taken about the pivot
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < p.M && colok) {
                        float v = acc[a][b][r] + bv;
                        const int64_t off = (int64_t)row * d.ldz + col;
                        if (flags & DS_EPI_ACCUM) v += zout[off];
                        if (flags & DS_EPI_MASK) v = p.mask[(int64_t)row * d.ldmask + col] > 0.f ? v : 0.f;
                        if (flags & DS_EPI_RELU) v = fmaxf(v, 0.f);
                        zout[off] = v;
                        const float u = v - pv;
                        s += u;
                        q += u * u;
                    }
                }
            }
            csum[b] += s;
            csq[b] += q;
        }
        if (odd) f0 = f1;      // the next row tile's first K-tile was fetched into f1: move it (after the epilogue, so the loads had time to land)
    }

    if (flags & DS_EPI_STATS) {      // one partial per wave: [2][Cout][P], P = 4 * gridDim.x
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const float s = csum[b] + __shfl_xor(csum[b], 32);
            const float q = csq[b] + __shfl_xor(csq[b], 32);
            const int col = n0 + b * 32 + li;
            if (lk == 0 && col < d.Cout) {
                p.stats[(int64_t)col * waves + wave_id] = s;
                p.stats[((int64_t)d.Cout + col) * waves + wave_id] = q;
            }
        }
    }
}

// ================================================================================================
// LDS-DMA variant (gfx950): K-tile 32, operands go global -> LDS with buffer_load_dwordx4 ... lds.
//
// Same implicit GEMM, same epilogue, different staging.  The loop above moves every operand through
// VGPRs (buffer_load -> s_waitcnt -> ds_write_b128): 12 staging registers per thread, three 13-cycle
// LDS stores per K-tile and a vmcnt(0) in the middle of every K-step.  Here each lane's 16 bytes land
// directly in LDS (a wave-instruction fills 1 KiB at M0 + lane*16; out-of-range lanes write zeros --
// probed in scripts/microbench/glds_test.hip), so there are no staging registers and no LDS stores, and the
// freed registers pay for a 32-deep K-tile: half as many barriers and fragment-read restarts per MFMA.
// Rows are 128 bytes and unpadded (the DMA destination is lane-linear), so the bank-conflict-free layout
// is an XOR swizzle applied on the SOURCE side: LDS slot (row, pc) holds the row's logical 16-byte chunk
// pc ^ ((row >> 1) & 7), and fragment reads apply the same involution (cdna_hip_programming.md rule 21).
// With ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} the eight rows of equal parity
// in a group map to eight different chunk columns: conflict-free.
// Per K-step: read all fragments of the current tile, issue the DMAs of the next tile, run the MFMAs,
// __syncthreads() (hipcc puts the vmcnt(0) for the DMAs there).
// ================================================================================================
constexpr int GK = 32;            // K-tile (floats)
constexpr int GCH = GK / 4;       // 16-byte chunks per row

template <int NT, bool BNMAJOR>
__global__ __launch_bounds__(256, NT == 1 ? 4 : (NT == 2 ? 3 : 2)) void conv_glds_kernel(const ConvParams p) {
    constexpr int BM = 128, BN = NT * 32;
    constexpr int ASZ = BM * GK, BSZ = BN * GK;       // floats per buffer
    constexpr int AJ = BM * GCH / 256;                // A DMA instructions per thread per K-tile (4)
    constexpr int BJ = BN * GCH / 256;                // B DMA instructions per thread per K-tile (NT)
    __shared__ __attribute__((aligned(128))) float smem[2 * ASZ + 2 * BSZ];
    float *As = smem;
    float *Bs = smem + 2 * ASZ;

    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const TileId tid0 = tile_id(p);
    const int n0 = tid0.col * BN;
    const int ohw = d.OH * d.OW;
    const int chunks = (d.Cin + GK - 1) / GK;
    const int KT_all = p.taps * chunks;
    const int kts = (KT_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int kt0 = (int)blockIdx.z * kts;
    const int KT = (kt0 + kts < KT_all ? kt0 + kts : KT_all) - kt0;
    float *const zout = p.z + (int64_t)blockIdx.z * p.d.z_split_stride;
    const int flags = d.flags;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);

    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;

    // ---- DMA slot of this thread: slot = j*256 + tid -> (row j*32 + tid/8, physical chunk tid%8) ------
    const int srow = tid >> 3;                                   // + 32*j
    const int slc4 = ((tid & 7) ^ ((tid >> 4) & 7)) * 4;         // logical k offset of the chunk it fetches
    // B, k-contiguous weights: same slot shape, rows are output channels
    // B, n-contiguous weights: tile is [k][n], slot -> (k = slot / (BN/4), n4 = slot % (BN/4)), no swizzle
    int b_row[BJ], b_col[BJ];
    bool b_ok[BJ];
#pragma unroll
    for (int i = 0; i < BJ; ++i) {
        const int slot = i * 256 + tid;
        if (BNMAJOR) {
            b_row[i] = slot / (BN / 4);                          // k inside the tile
            b_col[i] = n0 + (slot % (BN / 4)) * 4;               // first of 4 output channels
            b_ok[i] = b_col[i] < d.Cout;
        } else {
            b_row[i] = n0 + i * 32 + srow;                       // output channel
            b_col[i] = slc4;                                     // k inside the tile
            b_ok[i] = b_row[i] < d.Cout;
        }
    }

    int ih0[AJ], iw0[AJ];
    unsigned xb[AJ];
    int tap, c0, dh, dw;
    f32x16 acc[NT];

    auto setup_tile = [&](int tile) {
        const int m0 = tile * BM;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0 + j * 32 + srow;
            const bool rv = m < p.M;
            const int mm = rv ? m : 0;
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oh = r / d.OW;
            const int ow = r - oh * d.OW;
            ih0[j] = rv ? oh * d.stride - d.pad_t : -(1 << 20);
            iw0[j] = ow * d.stride - d.pad_l;
            xb[j] = (unsigned)n * (unsigned)(d.H * d.W) * (unsigned)d.ldx;
        }
        const int chunk = kt0 / p.taps;                          // channel chunk outer, tap inner
        tap = kt0 - chunk * p.taps;
        c0 = chunk * GK;
        dh = tap / d.KW;
        dw = tap - dh * d.KW;
    };

    auto issue_tile = [&](int buf) {       // DMA the K-tile the loader points at into buffer `buf`, then advance
        float *a_s = As + buf * ASZ + wm * 256;                  // this wave's 1 KiB window of pass j = 0
        const int ka = c0 + slc4;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int ih = ih0[j] + dh;
            const int iw = iw0[j] + dw;
            const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W && ka < d.Cin;
            const unsigned off = xb[j] + (unsigned)(ih * d.W + iw) * (unsigned)d.ldx + (unsigned)ka;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_x, (lds_ptr)(a_s + j * 1024), 16, ok ? off * 4u : kOOB, 0, 0, 0);
        }
        const int tap_eff = d.flip ? p.taps - 1 - tap : tap;
        const unsigned wt = (unsigned)tap_eff * (unsigned)d.w_tap_stride;
        float *b_s = Bs + buf * BSZ + wm * 256;
#pragma unroll
        for (int i = 0; i < BJ; ++i) {
            bool ok;
            unsigned off;
            if (BNMAJOR) {
                const int k = c0 + b_row[i];
                ok = b_ok[i] && k < d.Cin;
                off = wt + (unsigned)k * (unsigned)d.w_k_stride + (unsigned)b_col[i];
            } else {
                const int k = c0 + b_col[i];
                ok = b_ok[i] && k < d.Cin;
                off = wt + (unsigned)b_row[i] * (unsigned)d.w_n_stride + (unsigned)k;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr)(b_s + i * 1024), 16, ok ? off * 4u : kOOB, 0, 0, 0);
        }
        ++tap;
        if (++dw == d.KW) {
            dw = 0;
            if (++dh == d.KH) { dh = 0; tap = 0; c0 += GK; }
        }
    };

    // fragment addresses: lane (li, lk) owns k = lk*16 .. lk*16+15 of its row = logical chunks lk*4 .. lk*4+3
    const int arow = wm * 32 + li;
    const int asw = (arow >> 1) & 7;
    int a_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a_off[c] = arow * GK + (((lk * 4 + c) ^ asw) * 4);
    int b_offk[NT][4];
    if (!BNMAJOR) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int brow = b * 32 + li;
#pragma unroll
            for (int c = 0; c < 4; ++c) b_offk[b][c] = brow * GK + (((lk * 4 + c) ^ ((brow >> 1) & 7)) * 4);
        }
    }

    float af[16], bf[NT][16];
    auto read_frags = [&](int buf) {
        const float *a_s = As + buf * ASZ;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(a_s + a_off[c]);
            af[4 * c + 0] = v.x; af[4 * c + 1] = v.y; af[4 * c + 2] = v.z; af[4 * c + 3] = v.w;
        }
        const float *b_s = Bs + buf * BSZ;
        if (BNMAJOR) {
            const float *bp = b_s + (lk * 16) * BN + li;
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int s = 0; s < 16; ++s) bf[b][s] = bp[s * BN + b * 32];
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(b_s + b_offk[b][c]);
                    bf[b][4 * c + 0] = v.x; bf[b][4 * c + 1] = v.y; bf[b][4 * c + 2] = v.z; bf[b][4 * c + 3] = v.w;
                }
        }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[b][s], acc[b], 0, 0, 0);
    };

    int par = 0;
    if (tid0.row < p.row_tiles) {
        setup_tile(tid0.row);
        issue_tile(0);
    }
    __syncthreads();
    for (int tile = tid0.row; tile < p.row_tiles; tile += tid0.stride) {
        const int m0 = tile * BM;
        const bool has_next = tile + tid0.stride < p.row_tiles;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const bool last = kt + 1 == KT;
            read_frags(par);
            if (last && has_next) setup_tile(tile + tid0.stride);
            if (!last || has_next) issue_tile(par ^ 1);
            __builtin_amdgcn_sched_barrier(0);     // DMAs are issued before the MFMAs ...
            mfmas();
            __builtin_amdgcn_sched_barrier(0);     // ... and the MFMAs stay above the barrier's vmcnt(0)
            __syncthreads();
            par ^= 1;
        }

        // ---- epilogue: bias / accumulate / mask / relu, store, BatchNorm column statistics -------
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int col = n0 + b * 32 + li;
            const bool colok = col < d.Cout;
            const float bv = ((flags & DS_EPI_BIAS) && colok) ? p.bias[col] : 0.f;
            const float pv = (p.pivot && colok) ? p.pivot[col] : 0.f;      // statistics are taken about the pivot
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < p.M && colok) {
                    float v = acc[b][r] + bv;
                    const int64_t off = (int64_t)row * d.ldz + col;
                    if (flags & DS_EPI_ACCUM) v += zout[off];
                    if (flags & DS_EPI_MASK) v = p.mask[(int64_t)row * d.ldmask + col] > 0.f ? v : 0.f;
                    if (flags & DS_EPI_RELU) v = fmaxf(v, 0.f);
                    zout[off] = v;
                    const float u = v - pv;
                    s += u;
                    q += u * u;
                }
            }
            csum[b] += s;
            csq[b] += q;
        }
    }

    if (flags & DS_EPI_STATS) {
        float *red = smem;   // [4][BN][2]; safe: every wave passed the last K-loop barrier and no DMA is in flight
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const float s = csum[b] + __shfl_xor(csum[b], 32);
            const float q = csq[b] + __shfl_xor(csq[b], 32);
            if (lk == 0) {
                const int c = b * 32 + li;
                red[(wm * BN + c) * 2 + 0] = s;
                red[(wm * BN + c) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.Cout && tid0.row < p.row_tiles) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += red[(w * BN + tid) * 2 + 0];
                q += red[(w * BN + tid) * 2 + 1];
            }
            p.stats[(int64_t)(n0 + tid) * tid0.stride + tid0.row] = s;
            p.stats[((int64_t)d.Cout + n0 + tid) * tid0.stride + tid0.row] = q;
        }
    }
}


// ================================================================================================
// bf16-multiply variant (ds_conv_desc.dtype == DS_DTYPE_BF16): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// BASELINE configs[4] groundwork.  Storage is unchanged -- activations, weights (masters), BatchNorm statistics
// and z stay fp32 in HBM -- only the multiply runs on the bf16 matrix pipe (16x the fp32 MFMA rate): operands are
// rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way from the staging registers into LDS.
// Same implicit GEMM, same SRD loads with hardware zero fill, same epilogue (bias / accumulate / mask / relu /
// BatchNorm column statistics about the pivot, all in fp32 from the fp32 accumulators).
//   * K-tile 32 channels of one tap; both tiles live in LDS as [row][k] bf16 rows of 64 B padded to 80 B
//     (20-bank row stride: the sixteen rows of a ds_read_b128 lane group cover the 64 banks exactly once);
//   * A fragment of the 32x32x16 MFMA = lane (i, kh) holds A[i][8 kh .. 8 kh + 7] = one ds_read_b128; the B tile is
//     stored [n][k] for both weight layouts (n-contiguous HWIO weights are transposed on the LDS store by packing
//     the (k, k+1) pair of each of the four columns of a float4 into one dword);
//   * at 16x the matrix rate the loop is bound by operand staging, so tiles are 128 x 128 (NT = 4) wherever Cout
//     allows: 32 KB of fp32 operands staged per 1 MFLOP.
// Numerics: products of bf16-rounded operands, exact fp32 accumulation in k order; against the fp32 path the
// relative error is ~2^-9 per operand (documented tolerance of the bf16 parity tests: 1e-2 of max |ref|).
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int HK = 32;       // K-tile (channels)
constexpr int HLD = 40;      // LDS row stride in bf16 elements (80 B)

template <int NT, bool BNMAJOR, bool FOLD>
__global__ __launch_bounds__(256, NT <= 2 ? 3 : 2) void conv_bf16_kernel(const ConvParams p) {
    constexpr int BM = 128, BN = NT * 32;
    constexpr int AJ = 4;                                                // float4 A loads per thread per K-tile
    constexpr int BP = BNMAJOR ? (4 * BN + 255) / 256 : 0;               // (k, k+1) x float4(n) pairs per thread
    constexpr int BJ = BNMAJOR ? 0 : BN / 32;                            // float4(k) loads per thread
    constexpr int NB = BNMAJOR ? 2 * BP : BJ;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BM * HLD + 2 * BN * HLD];
    __bf16 *As = smem;
    __bf16 *Bs = smem + 2 * BM * HLD;

    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const TileId tid0 = tile_id(p);
    const int n0 = tid0.col * BN;
    const int ohw = d.OH * d.OW;
    const int chunks = (d.Cin + HK - 1) / HK;
    const int KT = p.taps * chunks;
    float *const zout = p.z;
    const int flags = d.flags;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);

    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;

    const int srow = tid >> 3, c4 = (tid & 7) * 4;        // A slot: rows srow + 32 j, channels c4 .. c4 + 3 of the K-tile
    int ih0[AJ], iw0[AJ];
    unsigned xb[AJ];
    int tap, c0, dh, dw;
    f32x4 ra[AJ], rb[NB > 0 ? NB : 1];
    f32x16 acc[NT];

    auto setup_tile = [&](int tile) {
        const int m0 = tile * BM;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0 + srow + 32 * j;
            const bool rv = m < p.M;
            const int mm = rv ? m : 0;
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oh = r / d.OW;
            const int ow = r - oh * d.OW;
            ih0[j] = rv ? oh * d.stride - d.pad_t : -(1 << 20);
            iw0[j] = ow * d.stride - d.pad_l;
            xb[j] = (unsigned)n * (unsigned)(d.H * d.W) * (unsigned)d.ldx;
        }
        tap = 0; c0 = 0; dh = 0; dw = 0;                  // channel chunk outer, tap inner (as the fp32 kernels)
    };

    auto load_tile = [&]() {
        const int k = c0 + c4;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int ih = ih0[j] + dh;
            const int iw = iw0[j] + dw;
            const int iwc = FOLD ? iw + k / d.fold_cin : iw;
            const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iwc < (unsigned)d.W && k < d.Cin;
            const unsigned off = xb[j] + (unsigned)(ih * d.W + iw) * (unsigned)d.ldx + (unsigned)k;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, ok ? off * 4u : kOOB, 0, 0));
        }
        const int tap_eff = d.flip ? p.taps - 1 - tap : tap;
        const unsigned wt = (unsigned)tap_eff * (unsigned)d.w_tap_stride;
        if (BNMAJOR) {
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int pi = tid + 256 * i;
                const int kp = pi / (BN / 4), n4 = pi - kp * (BN / 4);
                const int kk = c0 + 2 * kp, nn = n0 + 4 * n4;
                const bool okn = pi < 4 * BN && nn < d.Cout;
                const unsigned off = wt + (unsigned)kk * (unsigned)d.w_k_stride + (unsigned)nn;
                rb[2 * i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, (okn && kk < d.Cin) ? off * 4u : kOOB, 0, 0));
                rb[2 * i + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       srd_w, (okn && kk + 1 < d.Cin) ? (off + (unsigned)d.w_k_stride) * 4u : kOOB, 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < BJ; ++i) {
                const int nn = n0 + srow + 32 * i;
                const bool ok = nn < d.Cout && k < d.Cin;
                const unsigned off = wt + (unsigned)nn * (unsigned)d.w_n_stride + (unsigned)k;
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, ok ? off * 4u : kOOB, 0, 0));
            }
        }
        ++tap;
        if (++dw == d.KW) {
            dw = 0;
            if (++dh == d.KH) { dh = 0; tap = 0; c0 += HK; }
        }
    };

    auto store_tile = [&](int buf) {
        __bf16 *a_s = As + buf * BM * HLD;
        __bf16 *b_s = Bs + buf * BN * HLD;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            *reinterpret_cast<bf16x4 *>(a_s + (srow + 32 * j) * HLD + c4) = __builtin_convertvector(ra[j], bf16x4);
        if (BNMAJOR) {
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int pi = tid + 256 * i;
                const int kp = pi / (BN / 4), n4 = pi - kp * (BN / 4);
                if (pi < 4 * BN) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x2 pr = {rb[2 * i][jj], rb[2 * i + 1][jj]};
                        *reinterpret_cast<bf16x2 *>(b_s + (4 * n4 + jj) * HLD + 2 * kp) = __builtin_convertvector(pr, bf16x2);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BJ; ++i)
                *reinterpret_cast<bf16x4 *>(b_s + (srow + 32 * i) * HLD + c4) = __builtin_convertvector(rb[i], bf16x4);
        }
    };

    auto compute = [&](int buf) {
        const __bf16 *a_s = As + buf * BM * HLD + (wm * 32 + li) * HLD + kh * 8;
        const __bf16 *b_s = Bs + buf * BN * HLD + li * HLD + kh * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(a_s + ks * 16);
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const bf16x8 w = *reinterpret_cast<const bf16x8 *>(b_s + b * 32 * HLD + ks * 16);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, acc[b], 0, 0, 0);
            }
        }
    };

    int par = 0;
    if (tid0.row < p.row_tiles) {
        setup_tile(tid0.row);
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    for (int tile = tid0.row; tile < p.row_tiles; tile += tid0.stride) {
        const int m0 = tile * BM;
        const bool has_next = tile + tid0.stride < p.row_tiles;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const bool last = kt + 1 == KT;
            if (last && has_next) setup_tile(tile + tid0.stride);
            const bool fetch = !last || has_next;
            if (fetch) load_tile();
            compute(par);
            if (fetch) store_tile(par ^ 1);
            __syncthreads();
            par ^= 1;
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int col = n0 + b * 32 + li;
            const bool colok = col < d.Cout;
            const float bv = ((flags & DS_EPI_BIAS) && colok) ? p.bias[col] : 0.f;
            const float pv = (p.pivot && colok) ? p.pivot[col] : 0.f;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < p.M && colok) {
                    float v = acc[b][r] + bv;
                    const int64_t off = (int64_t)row * d.ldz + col;
                    if (flags & DS_EPI_ACCUM) v += zout[off];
                    if (flags & DS_EPI_MASK) v = p.mask[(int64_t)row * d.ldmask + col] > 0.f ? v : 0.f;
                    if (flags & DS_EPI_RELU) v = fmaxf(v, 0.f);
                    zout[off] = v;
                    const float u = v - pv;
                    s += u;
                    q += u * u;
                }
            }
            csum[b] += s;
            csq[b] += q;
        }
    }

    if (flags & DS_EPI_STATS) {
        float *red = reinterpret_cast<float *>(smem);   // [4][BN][2] floats; every wave passed the last K-loop barrier
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const float s = csum[b] + __shfl_xor(csum[b], 32);
            const float q = csq[b] + __shfl_xor(csq[b], 32);
            if (kh == 0) {
                const int c = b * 32 + li;
                red[(wm * BN + c) * 2 + 0] = s;
                red[(wm * BN + c) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.Cout && tid0.row < p.row_tiles) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += red[(w * BN + tid) * 2 + 0];
                q += red[(w * BN + tid) * 2 + 1];
            }
            p.stats[(int64_t)(n0 + tid) * tid0.stride + tid0.row] = s;
            p.stats[((int64_t)d.Cout + n0 + tid) * tid0.stride + tid0.row] = q;
        }
    }
}


// ================================================================================================
// Wide-tile 1x1 / GEMM variant: one wave per 32 rows x (NB * 32) columns, A operand register-direct.
//
// The 1x1 layers (Branch_0/1/2 Conv2d_0a_1x1 fused, Branch_3 Conv2d_0b_1x1, Conv2d_2b_1x1 and their dgrads) have no
// spatial footprint: row m of the A operand IS pixel m.  So the structure that made the Winograd kernel fast applies
// without the transform: lane (i, kh) of the 32x32x2 MFMA loads channels 4 kh .. 4 kh + 3 of row i as one float4 and
// that IS its A fragment -- no LDS round trip for A, no barrier on its account -- and every A fragment is used for
// NB * 4 MFMAs (NB up to 8: 256 columns, 128 accumulator registers) instead of 2 * 4 in the 128 x 64 LDS tile.  The B
// operand (a [16 k][NB*32] slice of the weights per K step, read in place from the HWIO tensor) goes global -> LDS by
// LDS-DMA, double buffered, shared by the workgroup's four waves (four 32-row groups); its fragments are read
// position by position in the shadow of the MFMAs, and the loads / DMAs of the next K step are issued BETWEEN the
// MFMA groups (conv_wino.hip: a burst of row-scattered loads stalls the in-order wave behind the texture path).
// Epilogue: stores (DS_EPI_ACCUM: z += result) + BatchNorm column statistics about the pivot (DS_EPI_STATS).
// ================================================================================================
// BNB (dgrad of a 1x1 conv + BatchNorm + ReLU layer, ds_conv_desc.bnb): x holds the layer's z and the A fragment becomes
// dz = bn_bwd_dz(z, dy, ...) as it is loaded -- a second float4 (dy, from up to three channel ranges whose boundaries are
// multiples of the 16-channel K step, so a step's range is wave-uniform) and five per-channel values from LDS per A
// float4, ~8 VALU per element against NB MFMAs: the layer's ds_bn_bwd_apply pass (12 B per element) disappears.
// Both A loads of the next K step are issued back to back: a row's 16 channels are ONE 64-byte sector, and with the second
// load a column block (8 MFMAs x the resident waves) behind the first the sector had left the 32 KB L1 again -- 19.8 M
// L1 -> L2 read requests per launch on the 28 x 28, 256 -> 288 layer against 9.6 M sectors of A (rocprofv3 TCP_TCC_READ_REQ,
// profiles/r05_notes.md).  Fifteen 1x1 shapes: dgrad with accumulate + sums 2399 -> 2334 us, step 14.51 -> 14.39 ms.
#ifndef DS_WIDE_A_B2B
#define DS_WIDE_A_B2B 1
#endif
// POOL (ds_conv_desc.pool_argmax: an Inception block's Branch_3, MaxPool 3x3/1 SAME -> Conv2d_0b_1x1, as ONE launch): the A
// fragment of pixel (h, w) becomes the maximum of x over its 3x3 neighbourhood as it is loaded.  A wave's 32-row block holds
// WHOLE image rows (28 of 32 lanes on the 28 / 14 / 7-wide maps), so that a pixel's left and right neighbours are the
// neighbouring lanes: lane (pixel, kh) loads its own column -- rows h-1, h, h+1, the same 2 x 4 channels -- takes the column
// maximum and the first row that holds it, gets its neighbours' column maxima by two DPP wave shifts and keeps the
// row-major-first winner (kh * 3 + kw: what maxpool3_fwd_rolling records), 3 loads per output instead of 9 and no LDS.
// With norm_rstd / norm_shift the maximum is taken over the raw z and relu(max * rstd + shift) applied once (rstd > 0), as
// ds_maxpool_bn_relu_fwd does.  The winners' bytes are the only thing besides z that is written: the pooled tensor
// (4 B per element out of the pool kernel and 4 B back into this one) never exists.
// AST > 0 (round 6 experiment): the A operand through an AST-stage LDS-DMA ring instead of registers.  A wave
// DMAs its own 32 rows x 16 channels of K step ks + AST - 1 (two 1 KB instructions, 16-byte chunks XOR-swizzled on the source
// side so the fragment reads are conflict-free) while K step ks runs; B is DMA'd AST - 2 steps ahead into AST - 1 buffers.  One
// in-order memory counter: a step issues its B DMAs first and its A DMAs last, and the step's barrier waits with
// vmcnt(2 (AST - 2) + DJ (AST - 3)) -- everything but the youngest requests -- so an A tile has AST - 1 K steps to arrive
// (today: under one).  Same MFMA sequence per output element: bit-identical to the register form.
template <int N>
__device__ __forceinline__ void barrier_keep_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

template <int NB, bool BNMAJOR, bool BNB = false, bool POOL = false, int AST = 0>
__global__ __launch_bounds__(256, NB <= 4 ? 3 : 2) void gemm_wide_kernel(const ConvParams p) {
    constexpr int BN = NB * 32, WK = 16;                       // K step: 16 channels = two float4 per lane
    constexpr int DJ = (WK * BN / 4 + 255) / 256;              // 16-byte DMA slots per thread per K step: ceil(NB / 2)
    constexpr int BSZ = DJ * 1024;                             // floats per B buffer (odd NB: the last DMA is half used)
    constexpr int BST = AST > 0 ? AST - 1 : 2;                 // B buffers
    static_assert(AST == 0 || (AST >= 3 && !BNB && !POOL), "the A ring is built for the plain and the normalising loader");
    __shared__ __attribute__((aligned(128))) float smem[BST * BSZ + 256 + AST * 2048];
    // BatchNorm + ReLU on load (ds_conv_desc.norm_rstd / norm_shift): rstd and shift of all Cin reduction channels;
    // BNB: rstd, shift, mean, coef[0], coef[1]
    __shared__ __attribute__((aligned(16))) float nrm[BNB ? 5 : 2][1024 + WK];
    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const TileId t0 = tile_id(p);
    const int n0 = t0.col * BN;
    const bool item = t0.row < p.row_tiles;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);
    const int K = d.Cin;
    // this wave's 32-row block: rows mrow0 ... of the output, the first Mw - mrow0 of them real
    int mrow0 = t0.row * 128 + wave * 32, Mw = p.M;
    bool hasU = false, hasD = false, hasL = false, hasR = false;      // POOL: which neighbours of the lane's pixel exist
    int prow = 0;                                                     // POOL: bytes between image rows of x
    if constexpr (POOL) {
        const int blk = t0.row * 4 + wave;                           // (uniform)
        const int n = blk / p.pool_bpi, rb = blk - n * p.pool_bpi;
        const int r0 = rb * p.pool_rpb;
        const int rows = (d.H - r0) < p.pool_rpb ? (d.H - r0) : p.pool_rpb;
        const int cnt = (item && n < d.N) ? rows * d.W : 0;
        mrow0 = (n * d.H + r0) * d.W;
        Mw = mrow0 + cnt;
        const int dr = li / d.W, ow = li - dr * d.W, oh = r0 + dr;
        const bool ok = li < cnt;
        hasU = ok && oh > 0;
        hasD = ok && oh + 1 < d.H;
        hasL = ok && ow > 0;
        hasR = ok && ow + 1 < d.W;
        prow = d.W * d.ldx * 4;
    }
    const int m = mrow0 + li;
    const unsigned voff = (item && m < Mw) ? ((unsigned)m * (unsigned)d.ldx + 4u * kh) * 4u : kOOB;

    // B DMA slots.  n-contiguous weights (forward): buffer layout [k][n], slot -> (k = idx / (BN/4), n4 = idx % (BN/4)).
    // k-contiguous weights (dgrad): buffer layout [n][16 k] with the four 16-byte chunks of a row XOR-swizzled by
    // (n >> 2) & 3 on the SOURCE side (the DMA destination is lane-linear), conflict-free for ds_read_b128.
    unsigned uoff[DJ];
    int ukq[DJ];
#pragma unroll
    for (int i = 0; i < DJ; ++i) {
        const int idx = i * 256 + tid;
        if (BNMAJOR) {
            const int k = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
            ukq[i] = k;
            uoff[i] = (n < d.Cout && k < WK) ? ((unsigned)k * (unsigned)d.w_k_stride + (unsigned)n) * 4u : kOOB;
        } else {
            const int nl = idx >> 2, pc = idx & 3, kq = pc ^ ((nl >> 2) & 3);
            ukq[i] = kq * 4;
            uoff[i] = (n0 + nl < d.Cout && nl < BN) ? ((unsigned)(n0 + nl) * (unsigned)d.w_n_stride + 4u * kq) * 4u : kOOB;
        }
    }
    auto dma_b = [&](int buf, int c0, int i) {
        // past the end of the reduction the weights must read as zeros (the A operand there is whatever follows in
        // the row): out-of-range offset
        const bool ok = c0 + ukq[i] < K;
        const unsigned off = BNMAJOR ? uoff[i] + (unsigned)c0 * (unsigned)d.w_k_stride * 4u : uoff[i] + (unsigned)c0 * 4u;
        // (written as an if: hipcc's host pass silently dropped this kernel's stub with a `cond ? off : kOOB` here)
        unsigned o = off;
        if (!ok || uoff[i] == 0x80000000u) o = 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr)(smem + buf * BSZ + wave * 256 + i * 1024), 16, o, 0, 0, 0);
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    // POOL: the rows above and below (out-of-range where they do not exist), the winners' bytes, the neighbour exchange
    const unsigned vup = (POOL && hasU) ? voff - (unsigned)prow : kOOB, vdn = (POOL && hasD) ? voff + (unsigned)prow : kOOB;
    const __amdgpu_buffer_rsrc_t srd_am = make_srd(POOL ? (const void *)d.pool_argmax : (const void *)p.x,
                                                   POOL ? (unsigned)((int64_t)p.M * K) : 0u);
    const unsigned vam = (POOL && item && m < Mw && t0.col == 0) ? (unsigned)m * (unsigned)K + 4u * kh : kOOB;
    f32x4 ru0, ru1, rd0, rd1;                      // POOL: the same channels of the pixel above / below
    auto load_ud = [&](int c0) {
        ru0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vup, c0 * 4, 0));
        ru1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vup, c0 * 4 + 32, 0));
        rd0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vdn, c0 * 4, 0));
        rd1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vdn, c0 * 4 + 32, 0));
    };
    // lane i <- lane i - 1 / lane i + 1 of the wave (DPP wave shifts: one VALU move each, no LDS).  The block starts and ends
    // at image-row boundaries, so the lanes the shift wraps into (0, 31 | 32, 63) never use what they get.
    auto from_left = [](float v) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
    };
    auto from_right = [](float v) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
    };
    // c <- max over the 3x3 neighbourhood (c = the centre row's values), returns the four winners kh * 3 + kw as bytes:
    // column maximum and its FIRST row, then among the (up to) three columns that reach the overall maximum the smallest
    // kh * 3 + kw -- the first maximum in row-major order, maxpool3_fwd_rolling's rule
    auto pool4 = [&](f32x4 &c, const f32x4 &u, const f32x4 &dn) -> unsigned {
        unsigned bytes = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float uu = hasU ? u[j] : -INFINITY, dd = hasD ? dn[j] : -INFINITY;
            const float cm = fmaxf(fmaxf(uu, c[j]), dd);
            const int t = uu == cm ? 0 : (c[j] == cm ? 3 : 6);
            const float lraw = from_left(cm), rraw = from_right(cm);
            const int tl = __builtin_amdgcn_update_dpp(0, t, 0x138, 0xf, 0xf, false);
            const int tr = __builtin_amdgcn_update_dpp(0, t, 0x130, 0xf, 0xf, false);
            const float lv = hasL ? lraw : -INFINITY, rv = hasR ? rraw : -INFINITY;
            const float best = fmaxf(fmaxf(lv, cm), rv);
            const int il = lv == best ? tl : 99, ic = cm == best ? t + 1 : 99, ir = rv == best ? tr + 2 : 99;
            const int arg = min(min(il, ic), ir);
            c[j] = best;
            bytes |= (unsigned)arg << (8 * j);
        }
        return bytes;
    };
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (AST == 0) {
        a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, voff, 0, 0));
        a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, voff, 32, 0));
    }
    if constexpr (POOL) load_ud(0);
    // BNB: the gradient dy of the layer's activation, per channel range its own descriptor and row offset
    const ds_bn_bwd_on_load &bb = p.bnb;
    __amdgpu_buffer_rsrc_t srd_dy[3];
    unsigned vdy[3];
    if (BNB) {
#pragma unroll
        for (int sg = 0; sg < 3; ++sg) {
            const bool has = sg < bb.nseg;
            const int cb = sg == 0 ? 0 : bb.c_end[sg - 1];
            const int ld = has ? bb.ld[sg] : 0;
            srd_dy[sg] = make_srd(has ? bb.dy[sg] : bb.dy[0], has ? (unsigned)(((int64_t)(p.M - 1) * ld + (bb.c_end[sg] - cb)) * 4) : 0u);
            vdy[sg] = (has && item && m < p.M) ? ((unsigned)m * (unsigned)ld + 4u * kh) * 4u : kOOB;
        }
    }
    auto load_dy = [&](int c0, int half) -> f32x4 {          // channels c0 + 8 half + 4 kh ... of the step that starts at c0
        const int sg = (c0 < bb.c_end[0] || bb.nseg == 1) ? 0 : ((c0 < bb.c_end[1] || bb.nseg == 2) ? 1 : 2);     // (uniform)
        const int cb = sg == 0 ? 0 : bb.c_end[sg - 1];
        const __amdgpu_buffer_rsrc_t r = sg == 0 ? srd_dy[0] : (sg == 1 ? srd_dy[1] : srd_dy[2]);
        const unsigned v = sg == 0 ? vdy[0] : (sg == 1 ? vdy[1] : vdy[2]);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v, (c0 - cb) * 4 + 32 * half, 0));
    };
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    if (BNB) {
        d0 = load_dy(0, 0);
        d1 = load_dy(0, 1);
    }
    if constexpr (AST == 0) {
#pragma unroll
        for (int i = 0; i < DJ; ++i) dma_b(0, 0, i);
    }
    const int ksteps = (K + WK - 1) / WK;
    // x holds pre-BatchNorm conv outputs: the A fragment becomes relu(x * rstd[k] + shift[k]) as it is loaded (two
    // packed multiply-adds and two packed max per float4, against 4 NB MFMAs); channels past Cin get (0, 0) -> 0
    const bool norm = d.norm_rstd != nullptr;          // (uniform)
    if (norm) {
        for (int c = tid; c < ksteps * WK; c += 256) {
            nrm[0][c] = c < K ? d.norm_rstd[c] : 0.f;
            nrm[1][c] = c < K ? d.norm_shift[c] : 0.f;
        }
    }
    auto apply_norm = [&](f32x4 &a, int c) {           // channels c .. c + 3
        const f32x4 r = *reinterpret_cast<const f32x4 *>(&nrm[0][c]), sh = *reinterpret_cast<const f32x4 *>(&nrm[1][c]);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_elementwise_max(__builtin_elementwise_fma(a, r, sh), zero);
    };
    if (BNB) {          // channels past Cin get zeros: dz = 0 there (and the weights read as zeros anyway)
        for (int c = tid; c < ksteps * WK; c += 256) {
            nrm[0][c] = c < K ? bb.rstd[c] : 0.f;
            nrm[1][c] = c < K ? bb.shift[c] : 0.f;
            nrm[BNB ? 2 : 0][c] = c < K ? bb.mean[c] : 0.f;
            nrm[BNB ? 3 : 0][c] = c < K ? bb.coef[c] : 0.f;
            nrm[BNB ? 4 : 0][c] = c < K ? bb.coef[K + c] : 0.f;
        }
    }
    auto apply_bnb = [&](f32x4 &a, const f32x4 &dy, int c) {      // a holds z of channels c .. c + 3
        const f32x4 r = *reinterpret_cast<const f32x4 *>(&nrm[0][c]), sh = *reinterpret_cast<const f32x4 *>(&nrm[1][c]);
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(&nrm[BNB ? 2 : 0][c]);
        const f32x4 k1 = *reinterpret_cast<const f32x4 *>(&nrm[BNB ? 3 : 0][c]), k2 = *reinterpret_cast<const f32x4 *>(&nrm[BNB ? 4 : 0][c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = ds::bn_bwd_dz(a[j], dy[j], r[j], sh[j], mu[j], k1[j], k2[j]);
    };
    __syncthreads();
    auto pool_step = [&](int c0) {          // a0 / a1 hold the centre row of channels c0 ...: pool them, record the winners
        const unsigned w0 = pool4(a0, ru0, rd0), w1 = pool4(a1, ru1, rd1);
        __builtin_amdgcn_raw_buffer_store_b32(w0, srd_am, vam, c0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(w1, srd_am, vam, c0 + 8, 0);
    };
    if constexpr (POOL) pool_step(0);
    if (norm && AST == 0) {
        apply_norm(a0, 4 * kh);
        apply_norm(a1, 8 + 4 * kh);
    }
    if (BNB) {
        apply_bnb(a0, d0, 4 * kh);
        apply_bnb(a1, d1, 8 + 4 * kh);
    }
    if constexpr (AST > 0) {
        // ---- the A ring (see the template's comment) ---------------------------------------------------------------------
        float *const ring = smem + BST * BSZ + 256;
        unsigned aoff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 16 * j + (lane >> 2), chunk = (lane & 3) ^ ((row >> 1) & 3);
            const int mr = mrow0 + row;
            aoff[j] = (item && mr < Mw) ? ((unsigned)mr * (unsigned)d.ldx + 4u * chunk) * 4u : kOOB;
        }
        auto dma_a = [&](int stage, int c0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_x, (lds_ptr)(ring + (stage * 4 + wave) * 512 + j * 256), 16, aoff[j], c0 * 4, 0, 0);
        };
        auto dma_b_all = [&](int buf, int c0) {
#pragma unroll
            for (int i = 0; i < DJ; ++i) dma_b(buf, c0, i);
        };
        // requests in the order of the steady state: ..., B(j + AST - 2), A(j + AST - 1), ...
#pragma unroll
        for (int j = -(AST - 1); j < 0; ++j) {
            if (j + AST - 2 >= 0 && j + AST - 2 < ksteps) dma_b_all((j + AST - 2) % BST, (j + AST - 2) * WK);
            if (j + AST - 1 < ksteps) dma_a((j + AST - 1) % AST, (j + AST - 1) * WK);
        }
        constexpr int KEEP = 2 * (AST - 2) + DJ * (AST - 3);
        if (ksteps > AST - 1) barrier_keep_vm<KEEP>(); else barrier_keep_vm<0>();
        int bst = 0, ast = 0;
        const int swz = (li >> 1) & 3;
        for (int ks = 0; ks < ksteps; ++ks) {
            const float *ra = ring + (ast * 4 + wave) * 512 + li * 16;
            a0 = *reinterpret_cast<const f32x4 *>(ra + ((kh ^ swz) * 4));
            a1 = *reinterpret_cast<const f32x4 *>(ra + (((2 + kh) ^ swz) * 4));
            if (norm) {
                apply_norm(a0, ks * WK + 4 * kh);
                apply_norm(a1, ks * WK + 8 + 4 * kh);
            }
            const float *b_s = smem + bst * BSZ;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float bf[8];
                if (BNMAJOR) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf[j] = b_s[(4 * kh + j) * BN + 32 * b + li];
                        bf[4 + j] = b_s[(8 + 4 * kh + j) * BN + 32 * b + li];
                    }
                } else {
                    const int nl = 32 * b + li, sw = (nl >> 2) & 3;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(b_s + nl * 16 + ((kh ^ sw) * 4));
                    const f32x4 hi = *reinterpret_cast<const f32x4 *>(b_s + nl * 16 + (((2 + kh) ^ sw) * 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bf[j] = lo[j]; bf[4 + j] = hi[j]; }
                }
                if (b == 0) {
                    if (ks + AST - 2 < ksteps) dma_b_all((bst + AST - 2) % BST, (ks + AST - 2) * WK);
                    if (ks + AST - 1 < ksteps) dma_a((ast + AST - 1) % AST, (ks + AST - 1) * WK);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bf[j], acc[b], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bf[4 + j], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + AST - 1 < ksteps) barrier_keep_vm<KEEP>(); else barrier_keep_vm<0>();
            bst = bst + 1 == BST ? 0 : bst + 1;
            ast = ast + 1 == AST ? 0 : ast + 1;
        }
    }
    for (int ks = 0; AST == 0 && ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        const int cn = (ks + 1) * WK;
        const float *b_s = smem + (ks & 1) * BSZ;
        f32x4 n0v = a0, n1v = a1;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // B fragments of column block b: lane (n, kh) needs B[k = 4 kh + j (+ 8)][32 b + n]
            float bf[8];
            if (BNMAJOR) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf[j] = b_s[(4 * kh + j) * BN + 32 * b + li];
                    bf[4 + j] = b_s[(8 + 4 * kh + j) * BN + 32 * b + li];
                }
            } else {
                const int nl = 32 * b + li, sw = (nl >> 2) & 3;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(b_s + nl * 16 + ((kh ^ sw) * 4));
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(b_s + nl * 16 + (((2 + kh) ^ sw) * 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) { bf[j] = lo[j]; bf[4 + j] = hi[j]; }
            }
            if (more) {                                    // next K step's operands, spread over the column blocks
                if (b == 0) {
                    n0v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, voff, cn * 4, 0));
                    if (BNB) d0 = load_dy(cn, 0);
                }
                if (b == ((NB > 1 && !DS_WIDE_A_B2B) ? 1 : 0)) {
                    n1v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, voff, cn * 4 + 32, 0));
                    if (BNB) d1 = load_dy(cn, 1);
                    if constexpr (POOL) load_ud(cn);      // (the previous step's rows were consumed before this K step began)
                }
                if (b < DJ) dma_b((ks + 1) & 1, cn, b);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bf[j], acc[b], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bf[4 + j], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more && DJ > NB) {
#pragma unroll
            for (int i = NB; i < DJ; ++i) dma_b((ks + 1) & 1, cn, i);
        }
        a0 = n0v;
        a1 = n1v;
        if constexpr (POOL) {
            if (more) pool_step(cn);
        }
        if (norm && more) {
            apply_norm(a0, cn + 4 * kh);
            apply_norm(a1, cn + 8 + 4 * kh);
        }
        if (BNB && more) {
            apply_bnb(a0, d0, cn + 4 * kh);
            apply_bnb(a1, d1, cn + 8 + 4 * kh);
        }
        __syncthreads();
    }

    // ---- epilogue: store, BatchNorm column statistics ---------------------------------------------------------------
    // Two phases.  (1) per column block: the accumulate / mask reads, the sums, the LDS combine -- results stay in the
    // accumulator registers; (2) all stores.  With the stores of block b in front of the reads of block b + 1 every
    // block cost a memory round trip (a load behind a store waits for it: one in-order counter).
    const int flags = d.flags;
    float *red = smem + BST * BSZ;
    // (outputs of 2 GiB and more stay on the LDS-tile kernels: wide_nb)
    // POOL: the descriptors end with this wave's last real row (Mw, wave-uniform), so the rows a 28-pixel block leaves
    // unused -- they would alias the NEXT block's pixels -- are dropped by the same range check as the rows past M
    const bool any_row = Mw > mrow0;          // (uniform; POOL: a surplus wave past the last image has none)
    const __amdgpu_buffer_rsrc_t srd_z = make_srd(p.z, any_row ? (unsigned)(((int64_t)(Mw - 1) * d.ldz + d.Cout) * 4) : 0u);
    const __amdgpu_buffer_rsrc_t srd_m = make_srd((flags & DS_EPI_BNSUMS) ? p.mask : p.z,
                                                  ((flags & DS_EPI_BNSUMS) && any_row) ? (unsigned)(((int64_t)(Mw - 1) * d.ldmask + d.Cout) * 4) : 0u);
    const int rz = d.ldz * 4, rm = d.ldmask * 4;                 // bytes per row
    const int rbase = mrow0 + 4 * kh;                            // this lane's first row; accumulator element r: + (r & 3) + 8 (r >> 2)
    // z (and the accumulate / activation reads at the same rows and columns) through buffer descriptors: ONE 32-bit lane
    // offset per column block plus the row's wave-uniform byte offset, added into the VECTOR offset (the scalar offset of a
    // buffer instruction is not range-checked); rows past M and columns past Cout (offset kOOB = 2^31, which the row offsets
    // cannot wrap) fall out of the descriptor's range: loads return 0, stores are dropped.  The 64-bit address per row and
    // the per-row branches of the pointer form cost more than the accesses: fifteen 1x1 shapes, forward with statistics
    // 1747 -> 1658 us, dgrad with accumulate + sums 2346 -> 2121 us, step 14.34 -> 14.10 ms (three interleaved pairs).
    auto ld_row = [&](__amdgpu_buffer_rsrc_t srd, unsigned v, int r, int row_bytes) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, v + (unsigned)(((r & 3) + 8 * (r >> 2)) * row_bytes), 0, 0));
    };
    auto st_row = [&](float val, unsigned v, int r) {
        const unsigned bits = __builtin_bit_cast(unsigned, val);          // (of a COPY: bit_cast of a vector-element lvalue reads element 0)
        __builtin_amdgcn_raw_buffer_store_b32(bits, srd_z, v + (unsigned)(((r & 3) + 8 * (r >> 2)) * rz), 0, 2 /* nt */);
    };
    float pss[NB], pqq[NB];                          // (threads 0..31) this workgroup's partials of column block b
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = n0 + 32 * b + li;
        const bool colok = item && col < d.Cout;
        const float pv = (p.pivot && colok) ? p.pivot[col] : 0.f;
        const unsigned vz = colok ? (unsigned)(rbase * d.ldz + col) * 4u : kOOB;
        const unsigned vm = colok ? (unsigned)(rbase * d.ldmask + col) * 4u : kOOB;
        float s = 0.f, q = 0.f;
        if (flags & DS_EPI_ACCUM) {                  // the sixteen previous values, requested together
            float zv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) zv[r] = ld_row(srd_z, vz, r, rz);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] += zv[r];
        }
        if (flags & DS_EPI_BNSUMS) {
            // dgrad whose result dy feeds a BatchNorm + ReLU backward: column sums of g = dy (y > 0) and g * y, y = the
            // consumer layer's forward activation (same rows / columns as dy)
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                yv[r] = ld_row(srd_m, vm, r, rm);
            if (d.mask_rstd && colok) {      // `mask` holds z of the consumer layers: y = relu(z * rstd + shift) per column
                const float mr = d.mask_rstd[col], ms = d.mask_shift[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    yv[r] = row < Mw ? fmaxf(fmaf(yv[r], mr, ms), 0.f) : 0.f;      // (rows past M read 0, which a shift > 0 would turn on)
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float u = yv[r] > 0.f ? acc[b][r] : 0.f;          // (out of range: y = 0)
                s += u;
                q += u * yv[r];
            }
        } else if (flags & DS_EPI_STATS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < Mw && colok) {
                    const float u = acc[b][r] - pv;
                    s += u;
                    q += u * u;
                }
            }
        }
        pss[b] = pqq[b] = 0.f;
        if (flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) {
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            __syncthreads();
            if (kh == 0) {
                red[(wave * 32 + li) * 2 + 0] = s;
                red[(wave * 32 + li) * 2 + 1] = q;
            }
            __syncthreads();
            if (tid < 32) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pss[b] += red[(w * 32 + tid) * 2 + 0];
                    pqq[b] += red[(w * 32 + tid) * 2 + 1];
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = n0 + 32 * b + li;
        const bool colok = item && col < d.Cout;
        const unsigned vz = colok ? (unsigned)(rbase * d.ldz + col) * 4u : kOOB;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            st_row(acc[b][r], vz, r);
        if ((flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) && tid < 32 && item && n0 + 32 * b + tid < d.Cout) {
            float *const sp = p.stats + (int64_t)(n0 + 32 * b + tid) * t0.stride + t0.row;
            float *const qp = p.stats + ((int64_t)d.Cout + n0 + 32 * b + tid) * t0.stride + t0.row;
            if (p.fin.ticket) {          // (uniform) read back by another workgroup of THIS launch: write-through
                __hip_atomic_store(reinterpret_cast<unsigned *>(sp), __float_as_uint(pss[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned *>(qp), __float_as_uint(pqq[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *sp = pss[b];
                *qp = pqq[b];
            }
        }
    }
    // ds_bn_finalize inside the launch (ds_conv_desc.fin): the last workgroup of a column tile to publish its partials
    // finalizes the tile's columns -- one wave per column, bn_finalize_kernel's summation tree (ds_common.h) -- so the
    // dependent ds_bn_finalize launch disappears.  Hand-off as in lstm_seq.hip: write-through payload, every storing wave
    // drained, ONE relaxed agent-scope increment; no fence (a fence would write back the XCD's whole L2, z included).
    if (p.fin.ticket && item && (flags & DS_EPI_STATS)) {
        __builtin_amdgcn_s_waitcnt(0);          // this wave's partial stores are acknowledged
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.fin.ticket + t0.col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[0] = old + 1 == (unsigned)p.row_tiles ? 1.f : 0.f;
        }
        __syncthreads();
        if (red[0] != 0.f) {                     // (uniform)
            const int cend = min(n0 + BN, d.Cout);
            for (int c = n0 + wave; c < cend; c += 4)
                ds::bn_finalize_channel_by_wave(p.stats, t0.stride, d.Cout, c, p.fin_inv_count, p.fin, p.pivot);
            if (tid == 0) __hip_atomic_store(p.fin.ticket + t0.col, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ================================================================================================
// bf16 register-direct implicit GEMM (ds_conv_bf16): v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 storage.
//
// Second generation of the DS_DTYPE_BF16 path (conv_bf16_kernel above stages both operands through registers AND
// LDS one K-tile ahead: 184 TFLOP/s, every fetch latency exposed against 256 matrix cycles).  Here:
//   * the weights are converted ONCE per weight update into the kernel's own order (ds_weights_to_bf16):
//     wb[it][column][16 k] bf16 with it = channel chunk * taps + tap -- exactly the K-loop order -- zero padded in k, so
//     the B operand of a step (four (chunk, tap) iterations x NB*32 columns x 16 k = NB * 4 KB) is a linear run that
//     LDS-DMA drops into LDS in fragment layout; no conversion, no transpose, no K-tail logic in the loop;
//   * the A operand is register-direct: lane (i, kh) of the 32x32x16 MFMA loads channels 8 kh .. 8 kh + 7 of ITS
//     pixel at the current tap as two float4 (SRD loads; padding pixels and rows past M read zeros from an
//     out-of-range offset), rounds them to bf16 (v_cvt_pk_bf16_f32, RNE) and that IS its A fragment;
//   * a wave owns 32 pixels x NB*32 columns (NB <= 8: 128 accumulator registers, two waves per SIMD); each A
//     fragment feeds NB MFMAs; the next step's eight A loads and NB weight DMAs are issued between the MFMA groups.
// Epilogue: fp32 stores + BatchNorm column statistics about the pivot (flags 0 or DS_EPI_STATS).
// ================================================================================================
// X3 (ds_conv_f32x3): fp32 PRODUCTS on the bf16 matrix cores.  Every fp32 operand is the sum of three bf16 pieces
// (8 + 8 + 8 mantissa bits: a = a0 + a1 + a2 exactly to 2^-24 |a|); a*b is accumulated in fp32 as
// a1*b1 + a2*b0 + a0*b2 + a1*b0 + a0*b1 + a0*b0 (the dropped terms are below 2^-24 |ab|): six v_mfma_f32_32x32x16_bf16
// (8 passes each) per 16 reduction channels and 32x32 tile instead of eight v_mfma_f32_32x32x2_f32 (16 passes each), i.e.
// 2.67x fewer matrix cycles at the accuracy of the fp32 MFMA (scripts/microbench/mfma_x3.hip: relative rms error against fp64
// 3.26e-7 vs 3.22e-7 over K = 512; loop 1.6 - 1.9x faster including the splitting).  The weights are split once per
// weight update (ds_weights_to_f32x3: three planes per K-loop iteration), the A fragment in registers (~40 VALU per 8
// values, shared by the NB column blocks).  Measured on the 1x1 layers of the joint step (B = 256): 1.15 - 1.5x over the
// wide fp32 kernel on twelve of fourteen shapes, up to 163 TFLOP/s of fp32-accurate GEMM -- above the fp32 MFMA peak.
template <int NB, int KS, bool XB, bool X3 = false>      // KS x KS taps (1 or 3); XB: x is stored as bf16 (16-bit activation storage)
__global__ __launch_bounds__(256, 2) void conv_bf16d_kernel(const ConvParams p) {
    // iterations per step: X3 keeps three weight planes in LDS, so shorter steps (SI * NB % 4 == 0: whole DMA slots)
    constexpr int BN = NB * 32, SI = !X3 ? 4 : NB == 1 ? 4 : NB <= 4 ? 2 : 1, TAPS = KS * KS;
    constexpr int PL = X3 ? 3 : 1;                             // weight planes per iteration (X3: the three bf16 pieces)
    constexpr int NSL = SI * PL * NB / 4;                      // 16-byte DMA slots per thread and step
    static_assert(SI * PL * NB % 4 == 0, "a step's weight tile must be whole 256-lane DMA instructions");
    constexpr int BSZ = SI * PL * BN * 16;                     // bf16 elements per B buffer
    static_assert(!(X3 && XB), "ds_conv_f32x3 reads fp32 activations");
    __shared__ __attribute__((aligned(128))) __bf16 smem[2 * BSZ + 512];
    // X3 + ds_conv_desc.norm_rstd / norm_shift (1x1 only): x holds pre-BatchNorm values, relu(x*rstd + shift) on load
    __shared__ __attribute__((aligned(16))) float nrm[2][X3 ? 1024 + 16 : 4];
    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const TileId t0 = tile_id(p);
    const int n0 = t0.col * BN;
    const bool item = t0.row < p.row_tiles;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);
    const int chunks = (d.Cin + 15) >> 4;
    const int iters = chunks * TAPS;
    const int nsteps = (iters + SI - 1) / SI;

    // ---- this lane's pixel and the byte offset of its input pixel at every tap (channel 8 kh) -----------------------
    const int m = t0.row * 128 + wave * 32 + li;
    const bool rv = item && m < p.M;
    const int ohw = d.OH * d.OW;
    const int n = (rv ? m : 0) / ohw;
    const int r = (rv ? m : 0) - n * ohw;
    const int oh = r / d.OW, ow = r - oh * d.OW;
    unsigned voff[TAPS];
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int b = 0; b < KS; ++b) {
            const int ih = oh * d.stride - d.pad_t + a, iw = ow * d.stride - d.pad_l + b;
            const bool ok = rv && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
            voff[a * KS + b] = ok ? ((unsigned)((n * d.H + ih) * d.W + iw) * (unsigned)d.ldx + 8u * kh) * (XB ? 2u : 4u) : kOOB;
        }
    // ---- B DMA slots of a step: slot -> (local iteration, column, 8-k half); source is linear in wb ---------------
    const int ncols = p.col_total;                              // columns of wb (Cout rounded up to 32)
    unsigned uoff[NSL];
    int uit[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int sl = i * 256 + tid;
        const int itl = sl / (BN * 2), nn = (sl >> 1) % BN, half = sl & 1;
        uit[i] = itl;
        uoff[i] = (n0 + nn < ncols) ? ((unsigned)((itl * ncols + n0 + nn) * 16 + 8 * half)) * 2u : kOOB;
    }
    const unsigned it_bytes = (unsigned)ncols * 32u;            // bytes of one iteration's weights
    auto dma_b = [&](int buf, int step, int i) {
        unsigned o = uoff[i] + (unsigned)(step * SI * PL) * it_bytes;
        if (step * SI * PL + uit[i] >= iters * PL || uoff[i] == 0x80000000u) o = 0x80000000u;     // past the reduction: zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr)(smem + buf * BSZ + wave * 512 + i * 2048), 16, o, 0, 0, 0);
    };
    // A loads of one (chunk, tap) iteration
    auto load_a = [&](int it, f32x4 &lo, f32x4 &hi) {
        const int chunk = it / TAPS, tap = it - chunk * TAPS;
        unsigned vo = voff[0];
#pragma unroll
        for (int k = 1; k < TAPS; ++k) vo = tap == k ? voff[k] : vo;
        if (XB) {           // eight bf16 channels = one 16-byte load, already the fragment
            lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vo, chunk * 32, 0));
        } else {
            lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vo, chunk * 64, 0));
            hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, vo, chunk * 64 + 16, 0));
        }
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    f32x4 alo[SI], ahi[SI];
#pragma unroll
    for (int j = 0; j < SI; ++j) load_a(j, alo[j], ahi[j]);
#pragma unroll
    for (int i = 0; i < NSL; ++i) dma_b(0, 0, i);
    bool norm = false;
    if constexpr (X3) {
        norm = d.norm_rstd != nullptr;          // (uniform)
        if (norm)
            for (int c = tid; c < chunks * 16; c += 256) {
                nrm[0][c] = c < d.Cin ? d.norm_rstd[c] : 0.f;
                nrm[1][c] = c < d.Cin ? d.norm_shift[c] : 0.f;
            }
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const bool more = st + 1 < nsteps;
        const __bf16 *b_s = smem + (st & 1) * BSZ + li * 16 + kh * 8;
        f32x4 nlo[SI], nhi[SI];
#pragma unroll
        for (int j = 0; j < SI; ++j) {
            // A fragment: 8 consecutive channels rounded to bf16 (X3: split into three bf16 pieces)
            bf16x8 af, af1, af2;
            if (XB) {
                af = __builtin_bit_cast(bf16x8, alo[j]);
            } else {
                f32x4 xl = alo[j], xh = ahi[j];
                if constexpr (X3) {
                    if (norm) {
                        const int c = ((st * SI + j) / TAPS) * 16 + 8 * kh;
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        xl = __builtin_elementwise_max(__builtin_elementwise_fma(xl, *reinterpret_cast<const f32x4 *>(&nrm[0][c]),
                                                                                 *reinterpret_cast<const f32x4 *>(&nrm[1][c])), zero);
                        xh = __builtin_elementwise_max(__builtin_elementwise_fma(xh, *reinterpret_cast<const f32x4 *>(&nrm[0][c + 4]),
                                                                                 *reinterpret_cast<const f32x4 *>(&nrm[1][c + 4])), zero);
                    }
                }
                bf16x4 l4 = __builtin_convertvector(xl, bf16x4), h4 = __builtin_convertvector(xh, bf16x4);
                af = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (X3) {
                    xl -= __builtin_convertvector(l4, f32x4);
                    xh -= __builtin_convertvector(h4, f32x4);
                    l4 = __builtin_convertvector(xl, bf16x4);
                    h4 = __builtin_convertvector(xh, bf16x4);
                    af1 = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                    xl -= __builtin_convertvector(l4, f32x4);
                    xh -= __builtin_convertvector(h4, f32x4);
                    af2 = __builtin_shufflevector(__builtin_convertvector(xl, bf16x4), __builtin_convertvector(xh, bf16x4), 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            if (more) {             // next step's operands between the MFMA groups
                load_a((st + 1) * SI + j, nlo[j], nhi[j]);
#pragma unroll
                for (int i = j; i < NSL; i += SI) dma_b((st + 1) & 1, st + 1, i);
            }
            bf16x8 bfr[PL][NB];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int b = 0; b < NB; ++b) bfr[pl][b] = *reinterpret_cast<const bf16x8 *>(b_s + ((j * PL + pl) * BN + b * 32) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if constexpr (X3) {         // small terms first
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, bfr[1][b], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2, bfr[0][b], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[PL - 1][b], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, bfr[0][b], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[PL > 1 ? 1 : 0][b], acc[b], 0, 0, 0);
                }
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[0][b], acc[b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < SI; ++j) { alo[j] = nlo[j]; ahi[j] = nhi[j]; }
        __syncthreads();
    }

    // ---- epilogue: store, BatchNorm column statistics (two phases as gemm_wide_kernel: reads and sums, then stores) -----
    const int flags = d.flags;
    float *red = reinterpret_cast<float *>(smem + 2 * BSZ);
    const int mrow0 = t0.row * 128 + wave * 32;
    // (as gemm_wide_kernel) z / accumulate / activation accesses through buffer descriptors: 32-bit lane offset + row offset
    // in the vector offset, hardware range check for rows past M and columns past Cout (kOOB)
    const bool m16 = d.mask_dtype == DS_DTYPE_BF16;      // (uniform) BatchNorm-sums activation in bf16 storage
    const bool z16 = d.z_dtype == DS_DTYPE_BF16;         // (uniform) z stored rounded to bf16 (forward, no accumulate)
    const unsigned zeb = z16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t srd_z = make_srd(p.z, (unsigned)(((int64_t)(p.M - 1) * d.ldz + d.Cout) * zeb));
    const __amdgpu_buffer_rsrc_t srd_m = make_srd((flags & DS_EPI_BNSUMS) ? p.mask : p.z,
                                                  (flags & DS_EPI_BNSUMS) ? (unsigned)(((int64_t)(p.M - 1) * d.ldmask + d.Cout) * (m16 ? 2 : 4)) : 0u);
    const int rz = d.ldz * (int)zeb, rm = d.ldmask * (m16 ? 2 : 4);
    const int rbase = mrow0 + 4 * kh;
    auto roff = [](int r, int row_bytes) -> unsigned { return (unsigned)(((r & 3) + 8 * (r >> 2)) * row_bytes); };
    float pss[NB], pqq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = n0 + 32 * b + li;
        const bool colok = item && col < d.Cout;
        const float pv = (p.pivot && colok) ? p.pivot[col] : 0.f;
        const unsigned vz = colok ? (unsigned)(rbase * d.ldz + col) * zeb : kOOB;
        const unsigned vm = colok ? (unsigned)(rbase * d.ldmask + col) * (m16 ? 2u : 4u) : kOOB;
        float s = 0.f, q = 0.f;
        if (flags & DS_EPI_ACCUM) {
            float zv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                zv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_z, vz + roff(r, rz), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] += zv[r];
        }
        if (flags & DS_EPI_BNSUMS) {
            // (as gemm_wide_kernel) dgrad whose result dy feeds a BatchNorm + ReLU backward: column sums of
            // g = dy (y > 0) and g * y; `mask` holds y (fp32, or bf16 under 16-bit activation storage), or z with
            // mask_rstd / mask_shift
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (m16) yv[r] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(srd_m, vm + roff(r, rm), 0, 0) << 16);
                else yv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_m, vm + roff(r, rm), 0, 0));
            }
            if (d.mask_rstd && colok) {
                const float mr = d.mask_rstd[col], ms = d.mask_shift[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    yv[r] = row < p.M ? fmaxf(fmaf(yv[r], mr, ms), 0.f) : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float u = yv[r] > 0.f ? acc[b][r] : 0.f;          // (out of range: y = 0)
                s += u;
                q += u * yv[r];
            }
        } else if (flags & DS_EPI_STATS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < p.M && colok) {
                    const float u = acc[b][r] - pv;
                    s += u;
                    q += u * u;
                }
            }
        }
        pss[b] = pqq[b] = 0.f;
        if (flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) {
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            __syncthreads();
            if (kh == 0) {
                red[(wave * 32 + li) * 2 + 0] = s;
                red[(wave * 32 + li) * 2 + 1] = q;
            }
            __syncthreads();
            if (tid < 32) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pss[b] += red[(w * 32 + tid) * 2 + 0];
                    pqq[b] += red[(w * 32 + tid) * 2 + 1];
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = n0 + 32 * b + li;
        const bool colok = item && col < d.Cout;
        const unsigned vz = colok ? (unsigned)(rbase * d.ldz + col) * zeb : kOOB;
        if (z16) {
            // z CENTRED about the pivot before it is rounded: the BatchNorm passes use z - mean, and for a channel with
            // |mean| >> sigma a bf16 z would carry (|mean| / sigma) 2^-9 of error into xhat; z - pivot (pivot = the previous
            // step's mean) is of the size of sigma, its rounding error 2^-9 of xhat itself.  The consumers get mean - pivot and
            // beta - (mean - pivot) rstd (ds_bn_finalize_centered)
            const float pvz = (p.pivot && colok) ? p.pivot[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const __bf16 hv = (__bf16)(acc[b][r] - pvz);          // round to nearest even
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), srd_z, vz + roff(r, rz), 0, 2 /* nt */);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float val = acc[b][r];          // (bit_cast of a vector-element lvalue reads element 0: copy first)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_z, vz + roff(r, rz), 0, 2 /* nt */);
            }
        }
        if ((flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) && tid < 32 && item && n0 + 32 * b + tid < d.Cout) {
            p.stats[(int64_t)(n0 + 32 * b + tid) * t0.stride + t0.row] = pss[b];
            p.stats[((int64_t)d.Cout + n0 + 32 * b + tid) * t0.stride + t0.row] = pqq[b];
        }
    }
}

// wb[it][column][16] bf16, it = chunk * taps + tap, from the TF HWIO filter w [taps][Cin][Cout] (fp32).
//   dgrad == 0: column = co, k = ci (forward);  dgrad == 1: column = ci, k = co, taps flipped (Conv2DBackpropInput).
// Columns are padded to a multiple of 32 and k to a multiple of 16 with zeros.
__global__ __launch_bounds__(256) void weights_to_bf16_kernel(const float *w, __bf16 *wb, int Cin, int Cout, int taps,
                                                              int dgrad) {
    const int K = dgrad ? Cout : Cin, Ncol = dgrad ? Cin : Cout;
    const int chunks = (K + 15) >> 4, ncols = (Ncol + 31) / 32 * 32;
    const int64_t total = (int64_t)chunks * taps * ncols * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i & 15);
        const int64_t rest = i >> 4;
        const int col = (int)(rest % ncols);
        const int it = (int)(rest / ncols);
        const int chunk = it / taps, tap = it - chunk * taps;
        const int k = chunk * 16 + j;
        float v = 0.f;
        if (col < Ncol && k < K) {
            const int ci = dgrad ? col : k, co = dgrad ? k : col, tp = dgrad ? taps - 1 - tap : tap;
            v = w[((int64_t)tp * Cin + ci) * Cout + co];
        }
        wb[i] = (__bf16)v;
    }
}

// the three bf16 pieces of every weight, in the K-loop order of conv_bf16d_kernel<.., X3>: [iteration][piece][column][16 k]
__global__ __launch_bounds__(256) void weights_to_f32x3_kernel(const float *w, __bf16 *wb, int Cin, int Cout, int taps,
                                                               int dgrad) {
    const int K = dgrad ? Cout : Cin, Ncol = dgrad ? Cin : Cout;
    const int chunks = (K + 15) >> 4, ncols = (Ncol + 31) / 32 * 32;
    const int64_t total = (int64_t)chunks * taps * ncols * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i & 15);
        const int64_t rest = i >> 4;
        const int col = (int)(rest % ncols);
        const int it = (int)(rest / ncols);
        const int chunk = it / taps, tap = it - chunk * taps;
        const int k = chunk * 16 + j;
        float v = 0.f;
        if (col < Ncol && k < K) {
            const int ci = dgrad ? col : k, co = dgrad ? k : col, tp = dgrad ? taps - 1 - tap : tap;
            v = w[((int64_t)tp * Cin + ci) * Cout + co];
        }
        const __bf16 p0 = (__bf16)v;
        const float r1 = v - (float)p0;
        const __bf16 p1 = (__bf16)r1;
        const __bf16 p2 = (__bf16)(r1 - (float)p1);
        const int64_t o = (((int64_t)it * 3) * ncols + col) * 16 + j;
        wb[o] = p0;
        wb[o + (int64_t)ncols * 16] = p1;
        wb[o + (int64_t)ncols * 32] = p2;
    }
}

// ---- host-side dispatch -----------------------------------------------------------------------
struct TileCfg {
    int mt, nt;
    bool direct;     // register-direct kernel (tile = per-WAVE 32*mt x 32*nt) instead of the LDS kernel
    bool glds;       // LDS-DMA kernel, K-tile 32 (128 x 32*nt tile)
    bool bf16;       // bf16-multiply kernel (128 x 32*nt tile, nt <= 4)
    int wide;        // > 0: wide 1x1 kernel, 128 rows x 32*wide columns per workgroup
};

typedef void (*KernelFn)(const ConvParams);

int64_t conv_M(const ds_conv_desc *d) { return (int64_t)d->N * d->OH * d->OW; }

struct Variant {
    bool bnmajor, fold, vec;
};

// vector (16-byte) loads are legal when every stride is a multiple of 4 floats; pointer alignment is
// checked separately at launch
bool dims_vec(const ds_conv_desc *d) {
    const bool bnmajor = d->w_n_stride == 1 && d->w_k_stride != 1;
    const bool a_vec = (d->ldx % 4 == 0) && (d->Cin % 4 == 0);
    const bool b_vec = bnmajor ? (d->Cout % 4 == 0) && (d->w_k_stride % 4 == 0) && (d->w_tap_stride % 4 == 0)
                               : (d->Cin % 4 == 0) && (d->w_n_stride % 4 == 0) && (d->w_tap_stride % 4 == 0);
    return a_vec && b_vec;
}

Variant variant_of(const ds_conv_desc *d, const float *x, const float *w) {
    Variant v;
    v.bnmajor = d->w_n_stride == 1 && d->w_k_stride != 1;
    v.fold = d->fold_cin > 0;
    v.vec = dims_vec(d) && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0);
    return v;
}

template <int MT, int NT>
KernelFn lds_kernel_mn(Variant v) {
    if (v.bnmajor) {
        if (v.fold) return conv_igemm_kernel<MT, NT, true, true, true>;
        return v.vec ? conv_igemm_kernel<MT, NT, true, false, true> : conv_igemm_kernel<MT, NT, true, false, false>;
    }
    return v.vec ? conv_igemm_kernel<MT, NT, false, false, true> : conv_igemm_kernel<MT, NT, false, false, false>;
}

template <int MT>
KernelFn lds_kernel_m(int nt, Variant v) {
    switch (nt) {
        case 1: return lds_kernel_mn<MT, 1>(v);
        case 2: return lds_kernel_mn<MT, 2>(v);
        case 3: return lds_kernel_mn<MT, 3>(v);
        case 4: return lds_kernel_mn<MT, 4>(v);
        case 5: return lds_kernel_mn<MT, 5>(v);
        default: return lds_kernel_mn<MT, 6>(v);
    }
}

template <int MT, int NT>
KernelFn direct_kernel_mn(Variant v) {
    if (v.bnmajor) return v.fold ? conv_direct_kernel<MT, NT, true, true> : conv_direct_kernel<MT, NT, true, false>;
    return conv_direct_kernel<MT, NT, false, false>;
}

template <int MT>
KernelFn direct_kernel_m(int nt, Variant v) {
    switch (nt) {
        case 1: return direct_kernel_mn<MT, 1>(v);
        case 2: return direct_kernel_mn<MT, 2>(v);
        case 3: return direct_kernel_mn<MT, 3>(v);
        default: return direct_kernel_mn<MT, 4>(v);
    }
}

KernelFn glds_kernel(int nt, Variant v) {
    switch (nt) {
        case 1: return v.bnmajor ? conv_glds_kernel<1, true> : conv_glds_kernel<1, false>;
        case 2: return v.bnmajor ? conv_glds_kernel<2, true> : conv_glds_kernel<2, false>;
        default: return v.bnmajor ? conv_glds_kernel<3, true> : conv_glds_kernel<3, false>;
    }
}

KernelFn bf16_kernel(int nt, Variant v) {
    if (v.fold) {
        switch (nt) {
            case 1: return conv_bf16_kernel<1, true, true>;
            case 2: return conv_bf16_kernel<2, true, true>;
            case 3: return conv_bf16_kernel<3, true, true>;
            default: return conv_bf16_kernel<4, true, true>;
        }
    }
    switch (nt) {
        case 1: return v.bnmajor ? conv_bf16_kernel<1, true, false> : conv_bf16_kernel<1, false, false>;
        case 2: return v.bnmajor ? conv_bf16_kernel<2, true, false> : conv_bf16_kernel<2, false, false>;
        case 3: return v.bnmajor ? conv_bf16_kernel<3, true, false> : conv_bf16_kernel<3, false, false>;
        default: return v.bnmajor ? conv_bf16_kernel<4, true, false> : conv_bf16_kernel<4, false, false>;
    }
}

void launch_wide(int nb, bool bnmajor, bool bnb, dim3 grid, hipStream_t st, const ConvParams &p) {
    if (p.d.pool_argmax) {          // (pool3_nb: forward, one column tile of at most four blocks)
        switch (nb) {
            case 1: hipLaunchKernelGGL((gemm_wide_kernel<1, true, false, true>), grid, dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((gemm_wide_kernel<2, true, false, true>), grid, dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL((gemm_wide_kernel<3, true, false, true>), grid, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL((gemm_wide_kernel<4, true, false, true>), grid, dim3(256), 0, st, p); break;
        }
        return;
    }
    if (bnb) {          // (wide_nb: at most two column blocks per wave with ds_conv_desc.bnb)
        if (nb == 1) hipLaunchKernelGGL((gemm_wide_kernel<1, false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_wide_kernel<2, false, true>), grid, dim3(256), 0, st, p);
        return;
    }
    // round-6 experiment (tuning build only: DS_WIDE_RING = 3 / 4): the A operand through an LDS-DMA ring, NB = 2 .. 4.
    // Bit-identical and SLOWER on the sixteen 1x1 shapes of the step (profiles/r06_wide_ring.txt: forward with statistics
    // 1705 -> 1804 us at three stages / 1957 at four, dgrad with accumulate + sums 2138 -> 2172 / 2482): the ring's LDS caps
    // the kernel at three (two) workgroups per CU where the register form runs seven waves per SIMD -- it lives on occupancy
#ifdef DS_TUNING
    static int ring = -1;
    if (ring < 0) {
        const char *e = ds::tune_env("DS_WIDE_RING");
        ring = e ? atoi(e) : 0;
    }
    if ((ring == 3 || ring == 4) && nb >= 2 && nb <= 4) {
#define DS_RING(NBV, ASTV)                                                                                                   \
        if (nb == NBV && ring == ASTV) {                                                                                     \
            if (bnmajor) hipLaunchKernelGGL((gemm_wide_kernel<NBV, true, false, false, ASTV>), grid, dim3(256), 0, st, p);   \
            else hipLaunchKernelGGL((gemm_wide_kernel<NBV, false, false, false, ASTV>), grid, dim3(256), 0, st, p);          \
            return;                                                                                                          \
        }
        DS_RING(2, 3) DS_RING(3, 3) DS_RING(4, 3) DS_RING(2, 4) DS_RING(3, 4) DS_RING(4, 4)
#undef DS_RING
    }
#endif
#define DS_WIDE(NBV)                                                                                      \
    case NBV:                                                                                             \
        if (bnmajor) hipLaunchKernelGGL((gemm_wide_kernel<NBV, true>), grid, dim3(256), 0, st, p);        \
        else hipLaunchKernelGGL((gemm_wide_kernel<NBV, false>), grid, dim3(256), 0, st, p);               \
        break;
    switch (nb) {
        DS_WIDE(1)
        DS_WIDE(2)
        DS_WIDE(3)
        DS_WIDE(4)
        DS_WIDE(5)
        DS_WIDE(6)
        DS_WIDE(7)
        default:
        DS_WIDE(8)
    }
#undef DS_WIDE
}

// Column blocks per wave of the wide 1x1 kernel for `d`, or 0 when the layer stays on the LDS-tile kernels:
// plain 1x1 stride-1 GEMM shape, flags within {STATS}, vector-aligned, no split-K, enough workgroups to fill the chip.
int force_wide = -1;

// MaxPool 3x3/1 on load (ds_conv_desc.pool_argmax): forward 1x1 conv with n-contiguous weights on a map at most 32 wide
// (a wave's 32-row block holds whole image rows), whole 16-channel K steps, ONE column tile of at most four blocks (the
// pooling is the loader's work and would be redone per column tile).  Returns the column blocks per wave, 0 = unsupported.
int pool3_nb(const ds_conv_desc *d, bool vec) {
    if (!vec || d->dtype != DS_DTYPE_F32) return 0;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->fold_cin || d->splits > 1 || d->bnb) return 0;
    if (d->flags & ~DS_EPI_STATS) return 0;
    if (!(d->w_n_stride == 1 && d->w_k_stride != 1)) return 0;
    if (d->W > 32 || d->Cin % 16 != 0 || d->Cin < 32 || d->Cin > 1024 || d->Cout > 128) return 0;
    const int64_t M = conv_M(d);
    if (((M - 1) * d->ldz + d->Cout) * 4 >= (1ll << 31) || M * d->Cin >= (1ll << 31)) return 0;
    return (d->Cout + 31) / 32;
}

// POOL: image rows per 32-pixel block and blocks per image
void pool3_blocks(const ds_conv_desc *d, int *rpb, int *bpi) {
    *rpb = 32 / d->W;
    *bpi = (d->H + *rpb - 1) / *rpb;
}

int wide_nb(const ds_conv_desc *d, bool vec, bool bnb_cap = false) {
    if (d->pool_argmax) return pool3_nb(d, vec);
    if (force_wide < 0) {
        const char *e = ds::tune_env("DS_CONV_WIDE");          // A/B aid: 0 = never
        force_wide = e ? atoi(e) : 1;
    }
    if (!force_wide || !vec || d->dtype != DS_DTYPE_F32) return 0;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->fold_cin || d->splits > 1) return 0;
    if (d->flags & ~(DS_EPI_STATS | DS_EPI_ACCUM | DS_EPI_BNSUMS)) return 0;
    if (d->Cin % 8 != 0 || d->Cin < 32) return 0;
    const int64_t M = conv_M(d);
    // the epilogue addresses z (and the BatchNorm-sums activation) through 32-bit buffer offsets
    if (((M - 1) * d->ldz + d->Cout) * 4 >= (1ll << 31)) return 0;
    if ((d->flags & DS_EPI_BNSUMS) && ((M - 1) * d->ldmask + d->Cout) * 4 >= (1ll << 31)) return 0;
    const int N = d->Cout;
    // Every workgroup puts one wave on each SIMD of its CU and the resident waves of a SIMD share its matrix pipe, so a
    // launch takes ceil(workgroups / CUs) x (one wave's work): NB blocks of MFMAs per K step plus about half a block
    // for the A fetch it does not share.  Pick the NB with the least of that (ties to the wider tile).
    int best = 0, best_pad = 0;
    int64_t best_cost = (int64_t)1 << 60;
    static int cA = -1;
    if (cA < 0) {
        const char *e = ds::tune_env("DS_WIDE_COST");
        cA = e ? atoi(e) : 0;                                    // in halves of a block
    }
    const int64_t row_tiles = (M + 127) / 128;
    // (BatchNorm backward on load is kept for narrow dgrads only -- it wins on Conv2d_2b and nowhere else,
    // profiles/r04_bnb_layers.txt -- so its loader is instantiated for one and two column blocks per wave)
    const int nb_max = (d->bnb || bnb_cap) ? 2 : 8;      // (bnb_cap: ds_conv_igemm_bnb_supported asks before bnb is attached)
    for (int nb = nb_max; nb >= (N <= 32 ? 1 : 2); --nb) {      // one block per wave only where two would be half padding
        const int tiles = (N + 32 * nb - 1) / (32 * nb);
        const int64_t rounds = (row_tiles * tiles + ds::kCUs - 1) / ds::kCUs;
        const int64_t cost = rounds * (2 * nb + cA);
        if (cost < best_cost) { best_cost = cost; best = nb; best_pad = tiles * 32 * nb; }
    }
    static int pin_nb = -1;                                      // tuning aid: DS_WIDE_NB pins the column blocks per wave
    if (pin_nb < 0) {
        const char *e = ds::tune_env("DS_WIDE_NB");
        pin_nb = e ? atoi(e) : 0;
    }
    if (pin_nb >= 1 && pin_nb <= nb_max) {
        best = pin_nb;
        best_pad = (N + 32 * best - 1) / (32 * best) * 32 * best;
    }
    const int64_t wgs = (M + 127) / 128 * ((N + 32 * best - 1) / (32 * best));
    // measured per shape (profiles/r02_wide_layers.txt): wins 1.1-1.4x except with one mostly padded column tile or too
    // few workgroups to cover the CUs (7x7 maps with N <= 128)
    static int min_wgs = -1;
    if (min_wgs < 0) {
        const char *e = ds::tune_env("DS_WIDE_MINWGS");
        min_wgs = e ? atoi(e) : 128;
    }
    if (force_wide < 2 && (wgs < min_wgs || best_pad * 100 > N * 125)) return 0;
    return best;
}

KernelFn kernel_for(TileCfg c, Variant v) {
    if (c.bf16) return bf16_kernel(c.nt, v);
    if (c.glds) return glds_kernel(c.nt, v);
    if (c.direct) return c.mt == 2 ? direct_kernel_m<2>(c.nt, v) : direct_kernel_m<1>(c.nt, v);
    return c.mt == 2 ? lds_kernel_m<2>(c.nt, v) : lds_kernel_m<1>(c.nt, v);
}

// Resident workgroups per CU of one instantiation (register/LDS limited), queried once and cached.
// The persistent grid is sized to exactly one resident wave of workgroups so no CU idles while a
// partial second wave runs; correctness never depends on it (no inter-workgroup communication).
int resident_per_cu(TileCfg c, Variant v) {
    static int cache[4][2][8][2][2][2];
    int &slot = cache[c.bf16 ? 3 : (c.glds ? 2 : (int)c.direct)][c.mt - 1][c.nt - 1][v.bnmajor][v.fold][v.vec];
    if (slot == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kernel_for(c, v), 256, 0) != hipSuccess || n < 1)
            n = 1;
        slot = n > 8 ? 8 : n;
    }
    return slot;
}

// tuning aids: ds_debug_conv_set_tile() / DS_CONV_CFG="mt,nt" pin the tile; ds_debug_conv_set_path() /
// DS_CONV_PATH=lds|direct pins the kernel family
int force_mt = -1, force_nt = -1, force_path = -1;

TileCfg pick_cfg(const ds_conv_desc *d, bool vec) {
    if (force_mt < 0) {
        force_mt = force_nt = 0;
        if (const char *e = ds::tune_env("DS_CONV_CFG")) sscanf(e, "%d,%d", &force_mt, &force_nt);
    }
    if (force_path < 0) {
        force_path = 0;
        if (const char *e = ds::tune_env("DS_CONV_PATH")) force_path = e[0] == 'l' ? 1 : (e[0] == 'd' ? 2 : (e[0] == 'g' ? 3 : 0));
    }
    const int64_t M = conv_M(d);
    const int N = d->Cout;
    const int pad32 = (N + 31) / 32 * 32, pad64 = (N + 63) / 64 * 64;
    TileCfg c = {1, 1, false, false, false, 0};
    if (int nb = wide_nb(d, vec)) {
        c.wide = nb;
        c.nt = nb;
        return c;
    }
    if (d->dtype == DS_DTYPE_BF16) {
        // staging-bound at 16x the matrix rate: the widest tile that does not pad Cout beyond the next 32
        c.bf16 = true;
        int best = 4, best_pad = 1 << 30;
        for (int nt = 4; nt >= 2; --nt) {
            const int pad = (N + 32 * nt - 1) / (32 * nt) * (32 * nt);
            if (pad < best_pad) { best_pad = pad; best = nt; }
        }
        c.nt = N <= 32 ? 1 : best;
        if (d->tile_nt > 0) c.nt = d->tile_nt > 4 ? 4 : d->tile_nt;
        if (force_nt > 0) c.nt = force_nt > 4 ? 4 : force_nt;
        return c;
    }
    // Measured (profiles/r01_lds_vs_direct_sweep.txt): the LDS-staged kernel wins on every shape of this
    // model -- fragment-shaped global loads touch 32 cache lines per wave instruction and are TA-bound --
    // so the register-direct family is opt-in only.
    c.direct = vec && force_path == 2;
    if (c.direct) {
        // per-WAVE tile 32*mt x 32*nt
        const int64_t row_tiles = (M + 31) / 32;
        if (pad64 * 100 <= pad32 * 112 && row_tiles * (pad64 / 64) >= 12 * ds::kCUs) c.nt = 2;
    } else {
        // Measured on MI355X over every conv/GEMM shape of the joint step (profiles/r01_tile_sweep.txt):
        // latency-bound per wave, so small tiles at 3-5 resident workgroups per CU beat wide ones.
        const int64_t row_tiles = (M + 127) / 128;
        if (pad64 * 100 <= pad32 * 112 && row_tiles * (pad64 / 64) >= 5 * ds::kCUs / 2) c.nt = 2;
    }
    if (d->tile_nt > 0 && !c.direct) c.nt = d->tile_nt > 6 ? 6 : d->tile_nt;     // per-layer choice (tuning table)
    if (force_mt > 0) c.mt = force_mt;
    if (force_nt > 0) c.nt = force_nt;
    if (c.direct && c.nt > 4) c.nt = 4;
    // LDS-DMA kernel (K-tile 32): needs 16-byte aligned operands and no folded stem.  Automatic for the
    // multi-tap convs whose reduction the 32-deep K-tile pads by no more than ~12 % over the 16-deep one:
    // measured 3-10 % faster there (long K loops), 5-10 % slower on the 1x1 layers whose 9-26 K-tiles
    // leave the prologue/epilogue exposed (profiles/r01_glds_sweep.txt).
    if (!c.direct && vec && d->fold_cin == 0 && force_path != 1) {
        const int k16 = (d->Cin + 15) / 16 * 16, k32 = (d->Cin + 31) / 32 * 32;
        if (force_path == 3 || (d->KH * d->KW >= 4 && k32 * 100 <= k16 * 112)) {
            c.glds = true;
            c.mt = 1;
            if (c.nt > 3) c.nt = 3;
        }
    }
    return c;
}

// Launch geometry (measured over every shape of the joint step, profiles/r01_grid_search.txt):
//   * up to kOneTilePerWg row tiles: one workgroup per tile, handed out by the hardware dispatcher as
//     slots free up.  A static stride over few tiles per workgroup quantises badly (98 row tiles on
//     96 workgroups = two rounds; 392 tiles on 256 = half the chip doing double work) and cost
//     20-45 % on the 14x14 and 7x7 maps.
//   * larger maps (conv2b/2c, stem: 6272 / 25088 row tiles): persistent workgroups, one resident wave,
//     striding over row tiles -- same speed within 2 %, and the BatchNorm statistics stay at a few
//     hundred partials per channel instead of one per row tile.
constexpr int kOneTilePerWg = 2048;

void grid_for(const ds_conv_desc *d, TileCfg c, Variant v, int *gx, int *gy, int *row_tiles, bool *one_per_tile = nullptr) {
    const int64_t M = conv_M(d);
    const int bm = (c.direct ? 32 : 128) * (c.glds ? 1 : c.mt), bn = 32 * c.nt;
    *row_tiles = (int)((M + bm - 1) / bm);
    if (c.wide && d->pool_argmax) {          // 32-row blocks of whole image rows, four per workgroup
        int rpb, bpi;
        pool3_blocks(d, &rpb, &bpi);
        *row_tiles = (int)(((int64_t)d->N * bpi + 3) / 4);
    }
    *gy = (d->Cout + bn - 1) / bn;
    const int wg_tiles = c.direct ? (*row_tiles + 3) / 4 : *row_tiles;     // row tiles in units of workgroups
    int x;
    bool one = false;
    if (c.wide) {                       // wide 1x1 kernel: always one workgroup per (row group, column tile)
        x = wg_tiles;
        one = true;
    } else if (wg_tiles <= kOneTilePerWg && !c.direct) {
        x = wg_tiles;
        one = true;
    } else {
        int target = (resident_per_cu(c, v) * ds::kCUs) / *gy;             // one resident wave of workgroups
        if (target < 8) target = 8;
        x = wg_tiles < target ? wg_tiles : target;
        if (x >= 8) x &= ~7;                  // multiple of 8: column tiles of a row tile share an XCD
    }
    if (d->grid_x > 0 && !c.direct) {       // per-layer override
        x = d->grid_x < wg_tiles ? d->grid_x : wg_tiles;
        one = x == wg_tiles;
    }
    *gx = x;
    if (one_per_tile) *one_per_tile = one;
}

}  // namespace

#ifdef DS_TUNING
extern "C" int ds_debug_conv_set_tile(int mt, int nt) {
    DS_REQUIRE((mt == 0 && nt == 0) || ((mt == 1 || mt == 2) && nt >= 1 && nt <= 6), "ds_debug_conv_set_tile: mt in {1,2}, nt in 1..6, or 0,0 = automatic");
    force_mt = mt;
    force_nt = nt;
    return DS_OK;
}
#endif

#ifdef DS_TUNING
extern "C" int ds_debug_conv_set_wide(int mode) {
    DS_REQUIRE(mode >= 0 && mode <= 2, "ds_debug_conv_set_wide: 0 = never, 1 = automatic, 2 = wherever the shape allows");
    force_wide = mode;
    return DS_OK;
}
#endif

#ifdef DS_TUNING
extern "C" int ds_debug_conv_set_path(int path) {
    DS_REQUIRE(path >= 0 && path <= 3, "ds_debug_conv_set_path: 0 = automatic, 1 = register-staged LDS kernel, 2 = register-direct kernel, 3 = LDS-DMA kernel");
    force_path = path;
    return DS_OK;
}
#endif

extern "C" int ds_conv_igemm_partials(const ds_conv_desc *d) {
    // BatchNorm convs are always 16-byte aligned in this model; ds_conv_igemm refuses DS_EPI_STATS
    // when the operands turn out not to be, so the count returned here is the one the launch uses.
    Variant v;
    v.bnmajor = d->w_n_stride == 1 && d->w_k_stride != 1;
    v.fold = d->fold_cin > 0;
    v.vec = dims_vec(d);
    const TileCfg c = pick_cfg(d, v.vec);
    int gx, gy, rt;
    grid_for(d, c, v, &gx, &gy, &rt);
    return c.direct ? gx * 4 : gx;
}

extern "C" int ds_conv_igemm_bnsums_supported(const ds_conv_desc *d) {
    // DS_EPI_BNSUMS lives in the wide 1x1 kernel's epilogue: supported where that kernel is the one chosen (16-byte
    // aligned operands assumed, as for ds_conv_igemm_partials)
    if (!d) return 0;
    ds_conv_desc t = *d;
    t.flags = (t.flags & DS_EPI_ACCUM) | DS_EPI_BNSUMS;
    return wide_nb(&t, dims_vec(&t)) > 0 ? 1 : 0;
}

extern "C" int ds_conv_igemm_norm_supported(const ds_conv_desc *d) {
    // BatchNorm + ReLU on load lives in the wide 1x1 kernel's loader (Cin <= 1024: the per-channel values sit in LDS)
    if (!d || d->Cin > 1024) return 0;
    return wide_nb(d, dims_vec(d)) > 0 ? 1 : 0;
}

extern "C" int ds_conv_igemm_pool3_supported(const ds_conv_desc *d) {
    // the 3x3 / 1 max pool on load lives in the wide 1x1 kernel's loader (forward instantiation)
    if (!d) return 0;
    return pool3_nb(d, dims_vec(d)) > 0 ? 1 : 0;
}

extern "C" int ds_conv_igemm_finalize_tickets(const ds_conv_desc *d) {
    // ds_bn_finalize inside the launch lives in the wide 1x1 kernel's epilogue: DS_EPI_STATS launches with at most 256
    // partials per channel (one wave re-reads a column's partials with four loads per lane)
    if (!d || !(d->flags & DS_EPI_STATS)) return 0;
    ds_conv_desc t = *d;
    t.partials = 0;
    t.fin = nullptr;
    const Variant v = {t.w_n_stride == 1 && t.w_k_stride != 1, t.fold_cin > 0, dims_vec(&t)};
    const TileCfg c = pick_cfg(&t, v.vec);
    if (!c.wide) return 0;
    int gx, gy, rt;
    grid_for(&t, c, v, &gx, &gy, &rt);
    return rt <= 256 ? gy : 0;
}

extern "C" int ds_conv_igemm_bnb_supported(const ds_conv_desc *d) {
    // BatchNorm backward on load lives in the wide 1x1 kernel's loader, k-contiguous-weights (dgrad) instantiation
    if (!d || d->Cin > 1024 || (d->w_n_stride == 1 && d->w_k_stride != 1) || d->norm_rstd) return 0;
    return wide_nb(d, dims_vec(d), true) > 0 ? 1 : 0;      // with the launch's own cap of two column blocks per wave
}

extern "C" int ds_conv_igemm(const ds_conv_desc *d, const float *x, const float *w, float *z, const float *bias,
                             const float *mask, float *stats, const float *pivot, void *stream) {
    DS_REQUIRE(d && x && w && z, "ds_conv_igemm: null argument");
    DS_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->N > 0 && d->OH > 0 && d->OW > 0, "ds_conv_igemm: bad dims");
    DS_REQUIRE(d->w_n_stride == 1 || d->w_k_stride == 1, "ds_conv_igemm: one weight stride must be 1");
    DS_REQUIRE(!(d->flags & DS_EPI_BIAS) || bias, "ds_conv_igemm: DS_EPI_BIAS without bias");
    DS_REQUIRE(!(d->flags & DS_EPI_MASK) || mask, "ds_conv_igemm: DS_EPI_MASK without mask");
    DS_REQUIRE(!(d->flags & DS_EPI_STATS) || stats, "ds_conv_igemm: DS_EPI_STATS without stats buffer");
    DS_REQUIRE(!(d->flags & DS_EPI_BNSUMS) || (stats && mask && !(d->flags & (DS_EPI_STATS | DS_EPI_MASK)) && d->ldmask >= d->Cout),
               "ds_conv_igemm: DS_EPI_BNSUMS needs the partials buffer (stats), y (mask, ldmask) and excludes STATS / MASK");
    DS_REQUIRE(conv_M(d) < (1ll << 31), "ds_conv_igemm: M too large");
    DS_REQUIRE(d->dtype == DS_DTYPE_F32 || d->dtype == DS_DTYPE_BF16, "ds_conv_igemm: dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16");
    const Variant v = variant_of(d, x, w);
    DS_REQUIRE(!(d->flags & DS_EPI_STATS) || v.vec == dims_vec(d),
               "ds_conv_igemm: DS_EPI_STATS needs 16-byte aligned operand pointers");

    ConvParams p;
    p.d = *d;
    p.x = x; p.w = w; p.z = z; p.bias = bias; p.mask = mask; p.stats = stats;
    p.pivot = (d->flags & DS_EPI_STATS) ? pivot : nullptr;
    p.M = (int)conv_M(d);
    p.taps = d->KH * d->KW;
    static int prio_mode = -1;
    if (prio_mode < 0) {
        const char *e = ds::tune_env("DS_CONV_PRIO");
        prio_mode = e ? atoi(e) : 0;   // opt-in: measured neutral-to-negative (profiles/r01_notes.md)
    }
    p.prio_mode = prio_mode;
    // extents of the two buffer descriptors (bytes from the operand pointer to the last float read)
    const int64_t x_elems = ((int64_t)d->N * d->H * d->W - 1) * d->ldx + (v.fold ? d->fold_cin : d->Cin);
    const int64_t w_elems = (int64_t)(p.taps - 1) * d->w_tap_stride + (int64_t)(d->Cout - 1) * d->w_n_stride +
                            (int64_t)(d->Cin - 1) * d->w_k_stride + 1;
    DS_REQUIRE(x_elems * 4 < (1ll << 31) && w_elems * 4 < (1ll << 31),
               "ds_conv_igemm: operand larger than 2 GiB (split the batch)");
    p.x_bytes = (unsigned)(x_elems * 4);
    p.w_bytes = (unsigned)(w_elems * 4);
    DS_REQUIRE(!v.fold || (v.bnmajor && v.vec && d->KW == 1 && d->fold_cin % 4 == 0 && d->ldx == d->fold_cin),
               "ds_conv_igemm: fold_cin needs KW=1, n-contiguous 16-byte-aligned weights, ldx==fold_cin");

    DS_REQUIRE(d->dtype != DS_DTYPE_BF16 || (v.vec && d->splits <= 1),
               "ds_conv_igemm: DS_DTYPE_BF16 needs 16-byte aligned operands with channel counts / strides divisible by 4, no split-K");
    const int splits = d->splits > 1 ? d->splits : 1;
    DS_REQUIRE(splits == 1 || (d->flags == 0 && d->z_split_stride >= (int64_t)(conv_M(d) - 1) * d->ldz + d->Cout),
               "ds_conv_igemm: split-K needs flags == 0 and non-overlapping output slabs");
    if (splits == 1) p.d.z_split_stride = 0;
    const TileCfg c = pick_cfg(d, v.vec);
    int gx, gy, rt;
    bool one;
    grid_for(d, c, v, &gx, &gy, &rt, &one);
    // the statistics partial count the caller planned with (ds_conv_igemm_partials at plan time) must be the one this
    // launch writes: the tile choice depends on debug switches / environment caches that may have changed since
    DS_REQUIRE((!d->norm_rstd && !d->norm_shift) || (c.wide && d->norm_rstd && d->norm_shift && d->Cin <= 1024),
               "ds_conv_igemm: norm_rstd / norm_shift are implemented by the wide 1x1 kernel only (ds_conv_igemm_norm_supported)");
    DS_REQUIRE((!d->mask_rstd && !d->mask_shift) || ((d->flags & DS_EPI_BNSUMS) && d->mask_rstd && d->mask_shift),
               "ds_conv_igemm: mask_rstd / mask_shift go with DS_EPI_BNSUMS");
    DS_REQUIRE(!(d->flags & DS_EPI_BNSUMS) || c.wide,
               "ds_conv_igemm: DS_EPI_BNSUMS is implemented by the wide 1x1 kernel only (ds_conv_igemm_bnsums_supported)");
    DS_REQUIRE(!(d->flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) || d->partials <= 0 || d->partials == (c.direct ? gx * 4 : gx),
               "ds_conv_igemm: the launch would write %d statistics partials but the plan was made for %d "
               "(a ds_debug_conv_set_* switch changed after planning?)", c.direct ? gx * 4 : gx, d->partials);
    p.row_tiles = rt;
    static int xcd_remap = -1;
    if (xcd_remap < 0) {
        const char *e = ds::tune_env("DS_CONV_XCD_REMAP");      // A/B aid; default on
        xcd_remap = e ? atoi(e) : 1;
    }
    dim3 grid(gx, gy, splits);
    p.col_tiles = 0;
    if (one && (c.wide || (xcd_remap && gy > 1)) && !c.direct) {    // one workgroup per tile: 1-D XCD-aware launch (TileId)
        p.col_tiles = gy;
        grid = dim3((rt * gy + 7) / 8 * 8, 1, splits);
    }
    if (d->bnb) {
        const ds_bn_bwd_on_load &b = *d->bnb;
        DS_REQUIRE(c.wide && !v.bnmajor && d->Cin <= 1024 && !d->norm_rstd,
                   "ds_conv_igemm: ds_conv_desc.bnb is implemented by the wide 1x1 kernel for k-contiguous weights only "
                   "(ds_conv_igemm_bnb_supported)");
        DS_REQUIRE(b.mean && b.rstd && b.shift && b.coef && b.nseg >= 1 && b.nseg <= 3 && b.c_end[b.nseg - 1] == d->Cin,
                   "ds_conv_igemm: bnb needs mean / rstd / shift / coef and 1..3 channel ranges that end at Cin");
        for (int i = 0; i < b.nseg; ++i) {
            const int cb = i ? b.c_end[i - 1] : 0;
            DS_REQUIRE(b.dy[i] && (((uintptr_t)b.dy[i]) & 15) == 0 && b.ld[i] % 4 == 0 && b.c_end[i] > cb && cb % 16 == 0 &&
                           b.ld[i] >= b.c_end[i] - cb && ((int64_t)(p.M - 1) * b.ld[i] + (b.c_end[i] - cb)) * 4 < (1ll << 31),
                       "ds_conv_igemm: bnb channel range %d (16-byte aligned dy, ld %% 4 == 0, boundaries %% 16 == 0)", i);
        }
        p.bnb = b;
    }
    p.fin = ds_bn_finalize_in_launch{};
    p.fin_inv_count = 0.0;
    if (d->fin) {
        DS_REQUIRE(c.wide && (d->flags & DS_EPI_STATS) && rt <= 256 && d->fin->ticket && d->fin->beta && d->fin->mean &&
                       d->fin->rstd && d->fin->shift && d->fin->count > 0,
                   "ds_conv_igemm: ds_conv_desc.fin needs a DS_EPI_STATS launch of the wide 1x1 kernel with at most 256 partials "
                   "(ds_conv_igemm_finalize_tickets) and beta / mean / rstd / shift / ticket / count");
        p.fin = *d->fin;
        p.fin_inv_count = 1.0 / (double)d->fin->count;
    }
    p.pool_rpb = p.pool_bpi = 0;
    if (d->pool_argmax) {
        DS_REQUIRE(c.wide && v.bnmajor && pool3_nb(d, v.vec) == c.wide && !d->bnb,
                   "ds_conv_igemm: ds_conv_desc.pool_argmax is implemented by the wide 1x1 kernel's forward instantiation only "
                   "(ds_conv_igemm_pool3_supported)");
        pool3_blocks(d, &p.pool_rpb, &p.pool_bpi);
    }
    if (c.wide) launch_wide(c.wide, v.bnmajor, d->bnb != nullptr, grid, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(kernel_for(c, v), grid, dim3(256), 0, (hipStream_t)stream, p);
    return ds::check_launch("ds_conv_igemm");
}

// ---- bf16 register-direct path -------------------------------------------------------------------------------------
namespace {
int g_bf16d_max_nb = 8;       // ds_debug_conv_bf16_set_max_nb (tuning)
int bf16d_nb(int Cout, bool x16_1x1) {
    // a workgroup's cost per K step ~ (A fetch + NB MFMAs); minimise column tiles x (c + NB), c ~ 4, ties to wider.
    // 1x1 launches from 16-bit activation storage are pure streaming (8 passes per 16 channels): at most four column blocks
    // per wave, i.e. 3-4 waves per SIMD instead of 2 -- twelve 1x1 shapes of the tower, forward with statistics:
    // 479 -> 434 us (scripts/bf16_wide_sweep.py); the input gradients (fp32 dz) keep eight (492 against 521)
    const int top = x16_1x1 && g_bf16d_max_nb > 4 ? 4 : g_bf16d_max_nb;
    int best = top, best_cost = 1 << 30;
    for (int nb = top; nb >= 1; --nb) {
        const int tiles = (Cout + 32 * nb - 1) / (32 * nb);
        const int cost = tiles * (4 + nb);
        if (cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}
bool bf16d_ok(const ds_conv_desc *d) {
    return (d->KH == d->KW) && (d->KH == 1 || d->KH == 3) && d->fold_cin == 0 && d->Cin % 8 == 0 && d->ldx % 4 == 0 &&
           !(d->flags & ~(DS_EPI_STATS | DS_EPI_ACCUM | DS_EPI_BNSUMS)) &&
           !((d->flags & DS_EPI_BNSUMS) && (d->flags & DS_EPI_STATS)) && d->splits <= 1 &&
           // the epilogue addresses z (and the BatchNorm-sums activation) through 32-bit buffer offsets
           ((conv_M(d) - 1) * d->ldz + d->Cout) * 4 < (1ll << 31) &&
           (!(d->flags & DS_EPI_BNSUMS) || ((conv_M(d) - 1) * d->ldmask + d->Cout) * 4 < (1ll << 31));
}
}  // namespace

extern "C" size_t ds_weights_bf16_bytes(int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad) {
    const int K = dgrad ? Cout : Cin, Ncol = dgrad ? Cin : Cout;
    return (size_t)((K + 15) / 16) * taps * ((Ncol + 31) / 32 * 32) * 16 * 2;
}

extern "C" int ds_weights_to_bf16(const float *w, void *wb, int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad,
                                  void *stream) {
    DS_REQUIRE(w && wb && Cin > 0 && Cout > 0 && taps > 0, "ds_weights_to_bf16: bad argument");
    const int64_t total = (int64_t)ds_weights_bf16_bytes(Cin, Cout, taps, dgrad) / 2;
    hipLaunchKernelGGL(weights_to_bf16_kernel, dim3(ds::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16 *)wb, Cin, Cout, taps, dgrad);
    return ds::check_launch("ds_weights_to_bf16");
}

extern "C" int ds_conv_bf16_supported(const ds_conv_desc *d) { return d && bf16d_ok(d) ? 1 : 0; }

#ifdef DS_TUNING
extern "C" int ds_debug_conv_bf16_set_max_nb(int nb) {
    DS_REQUIRE(nb >= 1 && nb <= 8, "ds_debug_conv_bf16_set_max_nb: 1 .. 8 column blocks per wave");
    g_bf16d_max_nb = nb;
    return 0;
}
#endif

extern "C" int ds_conv_bf16_partials(const ds_conv_desc *d) { return (int)((conv_M(d) + 127) / 128); }

extern "C" int ds_conv_bf16(const ds_conv_desc *d, const void *x, const void *wb, float *z, const void *mask, float *stats,
                            const float *pivot, void *stream) {
    DS_REQUIRE(d && x && wb && z, "ds_conv_bf16: null argument");
    DS_REQUIRE(!(d->flags & DS_EPI_BNSUMS) || (mask && stats && d->ldmask >= d->Cout &&
                                               (d->mask_dtype == DS_DTYPE_F32 || d->mask_dtype == DS_DTYPE_BF16)),
               "ds_conv_bf16: DS_EPI_BNSUMS needs mask (row stride ldmask, mask_dtype) and a partials buffer");
    DS_REQUIRE(!d->mask_rstd && !d->norm_rstd && !d->bnb, "ds_conv_bf16: the on-load transforms belong to the fp32 kernels");
    DS_REQUIRE(d->x_dtype == DS_DTYPE_F32 || (d->x_dtype == DS_DTYPE_BF16 && d->ldx % 8 == 0),
               "ds_conv_bf16: x_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16 (then ldx %% 8 == 0)");
    const bool xb = d->x_dtype == DS_DTYPE_BF16;
    DS_REQUIRE(bf16d_ok(d), "ds_conv_bf16: needs a 1x1 or 3x3 conv, Cin %% 8 == 0, ldx %% 4 == 0, flags within DS_EPI_STATS, or "
                            "DS_EPI_ACCUM | DS_EPI_BNSUMS for a dgrad");
    DS_REQUIRE(((((uintptr_t)x | (uintptr_t)wb) & 15) == 0) && conv_M(d) < (1ll << 31), "ds_conv_bf16: operands must be 16-byte aligned");
    DS_REQUIRE(!(d->flags & DS_EPI_STATS) || stats, "ds_conv_bf16: DS_EPI_STATS without stats buffer");
    DS_REQUIRE(d->z_dtype == DS_DTYPE_F32 || (d->z_dtype == DS_DTYPE_BF16 && !(d->flags & (DS_EPI_ACCUM | DS_EPI_BNSUMS))),
               "ds_conv_bf16: the output in bf16 storage excludes the accumulate / BatchNorm-sums epilogues");
    ConvParams p = {};
    p.d = *d;
    p.x = (const float *)x; p.w = (const float *)wb; p.z = z; p.stats = stats; p.mask = (const float *)mask;
    p.pivot = (d->flags & DS_EPI_STATS) ? pivot : nullptr;
    p.M = (int)conv_M(d);
    p.taps = d->KH * d->KW;
    const int64_t x_elems = ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin;
    const int64_t wb_bytes = (int64_t)((d->Cin + 15) / 16) * p.taps * ((d->Cout + 31) / 32 * 32) * 32;
    DS_REQUIRE(x_elems * 4 < (1ll << 31) && wb_bytes < (1ll << 31), "ds_conv_bf16: operand larger than 2 GiB");
    p.x_bytes = (unsigned)(x_elems * (xb ? 2 : 4));
    p.w_bytes = (unsigned)wb_bytes;
    p.col_total = (d->Cout + 31) / 32 * 32;
    const int nb = bf16d_nb(d->Cout, xb && d->KH == 1);
    p.row_tiles = (int)((conv_M(d) + 127) / 128);
    p.col_tiles = (d->Cout + 32 * nb - 1) / (32 * nb);
    const dim3 grid((unsigned)(((int64_t)p.row_tiles * p.col_tiles + 7) / 8 * 8));
    hipStream_t st = (hipStream_t)stream;
#define DS_B16(NBV)                                                                                        \
    case NBV:                                                                                              \
        if (xb) {                                                                                          \
            if (d->KH == 1) hipLaunchKernelGGL((conv_bf16d_kernel<NBV, 1, true>), grid, dim3(256), 0, st, p);  \
            else hipLaunchKernelGGL((conv_bf16d_kernel<NBV, 3, true>), grid, dim3(256), 0, st, p);             \
        } else {                                                                                           \
            if (d->KH == 1) hipLaunchKernelGGL((conv_bf16d_kernel<NBV, 1, false>), grid, dim3(256), 0, st, p); \
            else hipLaunchKernelGGL((conv_bf16d_kernel<NBV, 3, false>), grid, dim3(256), 0, st, p);            \
        }                                                                                                  \
        break;
    switch (nb) {
        DS_B16(1)
        DS_B16(2)
        DS_B16(3)
        DS_B16(4)
        DS_B16(5)
        DS_B16(6)
        DS_B16(7)
        default:
        DS_B16(8)
    }
#undef DS_B16
    return ds::check_launch("ds_conv_bf16");
}

// ---- fp32 products on the bf16 matrix cores (see conv_bf16d_kernel<.., X3>) ----------------------------------------
extern "C" size_t ds_weights_f32x3_bytes(int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad) {
    return 3 * ds_weights_bf16_bytes(Cin, Cout, taps, dgrad);
}

extern "C" int ds_weights_to_f32x3(const float *w, void *wb, int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad,
                                   void *stream) {
    DS_REQUIRE(w && wb && Cin > 0 && Cout > 0 && taps > 0, "ds_weights_to_f32x3: bad argument");
    const int64_t total = (int64_t)ds_weights_bf16_bytes(Cin, Cout, taps, dgrad) / 2;
    hipLaunchKernelGGL(weights_to_f32x3_kernel, dim3(ds::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16 *)wb, Cin, Cout, taps, dgrad);
    return ds::check_launch("ds_weights_to_f32x3");
}

extern "C" int ds_conv_f32x3_supported(const ds_conv_desc *d) {
    if (!d) return 0;
    ds_conv_desc t = *d;
    t.flags &= ~(DS_EPI_ACCUM | DS_EPI_BNSUMS);             // (the X3 epilogue also accumulates and emits BatchNorm sums)
    if ((d->flags & DS_EPI_BNSUMS) && (d->flags & DS_EPI_STATS)) return 0;
    return bf16d_ok(&t) && d->x_dtype == DS_DTYPE_F32 && (!d->norm_rstd || (d->KH == 1 && d->Cin <= 1024)) ? 1 : 0;
}

extern "C" int ds_conv_f32x3_partials(const ds_conv_desc *d) { return (int)((conv_M(d) + 127) / 128); }

extern "C" int ds_conv_f32x3(const ds_conv_desc *d, const float *x, const void *wb, float *z, const float *mask,
                             float *stats, const float *pivot, void *stream) {
    DS_REQUIRE(d && x && wb && z, "ds_conv_f32x3: null argument");
    DS_REQUIRE(ds_conv_f32x3_supported(d),
               "ds_conv_f32x3: needs a 1x1 or 3x3 conv on fp32 x, Cin %% 8 == 0, ldx %% 4 == 0, flags within DS_EPI_STATS | "
               "DS_EPI_ACCUM | DS_EPI_BNSUMS (norm_rstd / norm_shift: 1x1 only, Cin <= 1024)");
    DS_REQUIRE((!d->norm_rstd) == (!d->norm_shift) && (!d->mask_rstd) == (!d->mask_shift) &&
                   (!d->mask_rstd || (d->flags & DS_EPI_BNSUMS)),
               "ds_conv_f32x3: norm_rstd / norm_shift and mask_rstd / mask_shift come in pairs (the latter with DS_EPI_BNSUMS)");
    DS_REQUIRE(!(d->flags & DS_EPI_BNSUMS) || (mask && stats && d->ldmask >= d->Cout),
               "ds_conv_f32x3: DS_EPI_BNSUMS needs mask (row stride ldmask) and a partials buffer");
    DS_REQUIRE(((((uintptr_t)x | (uintptr_t)wb) & 15) == 0) && conv_M(d) < (1ll << 31), "ds_conv_f32x3: operands must be 16-byte aligned");
    DS_REQUIRE(!(d->flags & DS_EPI_STATS) || stats, "ds_conv_f32x3: DS_EPI_STATS without stats buffer");
    ConvParams p = {};
    p.d = *d;
    p.x = x; p.w = (const float *)wb; p.z = z; p.stats = stats; p.mask = mask;
    p.pivot = (d->flags & DS_EPI_STATS) ? pivot : nullptr;
    p.M = (int)conv_M(d);
    p.taps = d->KH * d->KW;
    const int64_t x_elems = ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin;
    const int64_t wb_bytes = 3 * (int64_t)((d->Cin + 15) / 16) * p.taps * ((d->Cout + 31) / 32 * 32) * 32;
    DS_REQUIRE(x_elems * 4 < (1ll << 31) && wb_bytes < (1ll << 31), "ds_conv_f32x3: operand larger than 2 GiB");
    p.x_bytes = (unsigned)(x_elems * 4);
    p.w_bytes = (unsigned)wb_bytes;
    p.col_total = (d->Cout + 31) / 32 * 32;
    // two column blocks per wave (one where Cout <= 32): measured best on all fourteen 1x1 layer shapes (NB = 4: the
    // split is better amortised but two waves per SIMD no longer fit; profiles/r03_f32x3_layers.txt)
    const int nb = d->Cout <= 32 ? 1 : 2;
    p.row_tiles = (int)((conv_M(d) + 127) / 128);
    p.col_tiles = (d->Cout + 32 * nb - 1) / (32 * nb);
    const dim3 grid((unsigned)(((int64_t)p.row_tiles * p.col_tiles + 7) / 8 * 8));
    hipStream_t st = (hipStream_t)stream;
    if (nb == 1) {
        if (d->KH == 1) hipLaunchKernelGGL((conv_bf16d_kernel<1, 1, false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_bf16d_kernel<1, 3, false, true>), grid, dim3(256), 0, st, p);
    } else {
        if (d->KH == 1) hipLaunchKernelGGL((conv_bf16d_kernel<2, 1, false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_bf16d_kernel<2, 3, false, true>), grid, dim3(256), 0, st, p);
    }
    return ds::check_launch("ds_conv_f32x3");
}
