// Conv2d_1a_7x7 (image_model/inception_v1.py:63: 7x7, stride 2, SAME, 3 -> 64 channels) on fp32 MFMA, reading the
// packed RGB images as the caller hands them over.
//
// The generic kernel runs this layer through a zero-padded 4-channel copy of the input with KW folded into the
// channel axis: K = 7 x 28 = 196 (147 real), both operands staged through LDS, ~60 TFLOP/s -- the slowest conv of the
// step.  This one is register-direct like gemm_wide_kernel:
//   * which physical k plays "k = 2 s + kh" of a v_mfma_f32_32x32x2_f32 step is free as long as A and B agree, so
//     lane (i, kh) loads the THREE channels of input pixel (2 j + kh) of kernel row dh with one 12-byte load
//     (buffer_load_dwordx3 straight from the [N, H, W, 3] image) and feeds them to three K steps: K = 7 rows x 4 pixel
//     pairs x 3 channels = 84 steps (168 k, 147 real: the eighth pixel of a row is a zero weight row and is not
//     loaded);
//   * the whole filter, reordered to that K order ([dh][j][c][kh][64]: 43 KB), is built in LDS once per workgroup
//     from the HWIO weights -- no separate packed copy to keep fresh; a B fragment is one conflict-free ds_read_b32;
//   * a wave owns 64 output pixels (two 32-row blocks) x 64 channels: every B fragment feeds two MFMAs, 64
//     accumulator registers, two workgroups per CU; the 8 loads of the next kernel row are issued ahead of the 48
//     MFMAs of the current one;
//   * workgroups are persistent (grid = 2 per CU) and keep the BatchNorm column sums about the pivot in registers
//     across their tiles: one partial per workgroup.
// SAME padding (pad 2 top/left for 224 -> 112) and rows past the end come back as zeros from an out-of-range offset.
//
// BF (ds_conv_stem_bf16, the 16-bit configurations): the same kernel on v_mfma_f32_32x32x16_bf16.  A lane's twelve values of
// a kernel row (four pixels x three channels) do not fill whole 8-value fragments, so TWO kernel rows (24 values) feed three
// MFMAs: K = 4 row pairs x 24 = 96 slots (147 real; the eighth row and the eighth pixel are zero weights), 11 MFMAs per
// accumulator instead of 84 -- 15x fewer matrix cycles, which leaves the image reads and the 64 output columns' stores.  x and
// w are rounded to bf16 (RNE) as they are packed; the filter lives in LDS as [mfma][kh][co][8 k] bf16 (22.5 KB), a B fragment
// is one ds_read_b128.  The generic path it replaces in those configurations was a zero-padded 4-channel copy of the batch
// (96 us) + the LDS-staged bf16 kernel (578 us).
//
// POOL (ds_conv_stem_pool, round 6): Conv2d_1a_7x7 -> [BatchNorm -> ReLU ->] MaxPool_2a_3x3 (inception_v1.py:63-67) with the
// 3x3 / 2 max pool INSIDE the conv kernel.  The pool commutes with relu(rstd * . + shift) (rstd > 0), and the frozen stem's
// backward pass needs only pooled tensors (ConvBN._bn_bwd_sums), so the full-resolution conv output -- 822 MB at B = 256,
// written here and read back by the pool pass -- never has to exist: the kernel writes max(z) over every window (a quarter
// of the elements) and the statistics of the FULL map.  A workgroup owns a contiguous range of conv ROW PAIRS (rows 2 i,
// 2 i + 1 = pooled row i minus its third row) of the batch, walks it in 256-pixel tiles as before, and folds every
// accumulator into a ring of pooled rows in LDS with ds_max_f32 (a conv pixel lies in up to 2 x 2 windows); a pooled row
// is complete once conv row 2 i + 2 has passed -- written out with 16-byte stores and reset -- so three ring slots cover a
// 256-pixel tile of the 112-wide map.  The one conv row behind the range that the last pooled row still needs (it belongs
// to the next workgroup's first pair) is computed here too (1 row in 57: +1.8 % of the matrix work, no exchange between
// workgroups) and left out of the statistics.  LDS: the filter without its zero rows (37.6 KB) + the ring (43 KB) = 80.6 KB,
// two workgroups per CU as before.
#include <math.h>
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

namespace {

constexpr unsigned kOOB = 0x80000000u;
constexpr int CO = 64;

struct StemParams {
    const float *x;      // [N, H, W, 3]
    const float *w;      // HWIO [7][7][cin_store][64]
    float *z;            // [N*OH*OW, ldz]
    float *stats;        // [2][64][P], P = gridDim.x
    const float *pivot;
    int N, H, W, OH, OW, pad_t, pad_l, cin_store, ldz;
    int M, tiles;        // output pixels, 256-pixel tiles
    unsigned x_bytes;
    int PH, PW, rp_total, nslots;      // POOL: pooled map, row pairs of the batch (N * PH), ring slots (pooled rows) in LDS
};

constexpr int kRingFloats = 3 * 56 * CO;      // POOL: three pooled rows of the 112-wide map (more rows of narrower ones)

// *p = max(*p, v) in LDS (ds_max_f32, no return value)
__device__ __forceinline__ void ds_fmax(float *p, float v) {
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool BF, bool POOL = false>
__global__ __launch_bounds__(256, (BF && !POOL) ? 3 : 2) void conv_stem_kernel(const StemParams p) {
    // row blocks of 32 output pixels per wave: two on the fp32 matrix cores (every B fragment feeds two MFMAs); ONE for BF,
    // which is bound by the image reads and the stores: three waves per SIMD, no spills
    constexpr int NA = BF ? 1 : 2, TP = 128 * NA;
    // [dh][j][c][kh][co]; POOL: without the zero rows of pixel 7 ([dh][21 (j, c, kh)][co]); BF: [mfma][kh][co][8] bf16
    constexpr int WROW = POOL ? 21 * CO : 4 * 3 * 2 * CO;
    __shared__ __attribute__((aligned(16))) float wl[BF ? 11 * 2 * CO * 8 / 2 : 7 * WROW];
    __shared__ __attribute__((aligned(16))) float ring_s[POOL ? kRingFloats : 4 * CO * 2];
    float *const red = ring_s;                 // (POOL: the statistics combine reuses the ring after its last row has left)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    if constexpr (BF) {
        __bf16 *wb = reinterpret_cast<__bf16 *>(wl);
        for (int i = tid; i < 11 * 2 * CO * 8; i += 256) {
            const int slot = i & 7, co = (i >> 3) & 63, k2 = (i >> 9) & 1, m = i >> 10;
            const int t = m / 3, q = (m - 3 * t) * 8 + slot;          // row pair, position among its 24 values
            const int r = q / 12, jj = (q - 12 * r) / 3, c = q % 3;
            const int dh = 2 * t + r, px = 2 * jj + k2;
            wb[i] = (__bf16)((dh < 7 && px < 7) ? p.w[((dh * 7 + px) * p.cin_store + c) * CO + co] : 0.f);
        }
    } else {
        // (all 42 reads of a thread requested before the first LDS write: one memory round trip instead of 42 in a row --
        // ~60 us at the head of every persistent workgroup)
        // POOL: the (j = 3, kh = 1) rows are not kept (their A operand is an out-of-range load = 0): 21 rows per kernel row,
        // walked in DESTINATION order -- rows 0 .. 17 = (j, c, kh) for j < 3, rows 18 .. 20 = (3, c, 0)
        constexpr int WN = (7 * WROW + 255) / 256;
        float wv[WN];
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            const int i = t * 256 + tid;
            const int co = i & 63;
            int k2, c, j, dh;
            if constexpr (POOL) {
                const int e = (i >> 6) % 21;
                dh = (i >> 6) / 21;
                j = e < 18 ? e / 6 : 3;
                c = e < 18 ? (e % 6) >> 1 : e - 18;
                k2 = e < 18 ? e & 1 : 0;
            } else {
                k2 = (i >> 6) & 1; c = (i >> 7) % 3; j = ((i >> 7) / 3) & 3; dh = (i >> 7) / 12;
            }
            const int px = 2 * j + k2;
            const bool ok = px < 7 && dh < 7;
            const float wq = p.w[ok ? ((dh * 7 + px) * p.cin_store + c) * CO + co : 0];          // (unconditional load: no branch)
            wv[t] = ok ? wq : 0.f;
        }
#pragma unroll
        for (int t = 0; t < WN; ++t)
            if (!POOL || t * 256 + tid < 7 * WROW) wl[t * 256 + tid] = wv[t];
    }
    if constexpr (POOL) {
        for (int i = tid; i < kRingFloats; i += 256) ring_s[i] = -INFINITY;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_z = __builtin_amdgcn_make_buffer_rsrc(p.z, 0, (int)(((int64_t)(p.M - 1) * p.ldz + CO) * 4), 0x00020000);
    const float pv0 = p.pivot ? p.pivot[li] : 0.f, pv1 = p.pivot ? p.pivot[32 + li] : 0.f;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    const int ohw = p.OH * p.OW;

    // the lane's two output pixels (row blocks a = 0, 1) of a tile, and the byte offsets of a kernel row's four pixels
    struct Pix { int ih0[NA], iw0[NA], nb[NA]; bool rv[NA]; };
    int mlim = p.M;                    // POOL: the end of this workgroup's pixel range (set below)
    int mbase = 0;                     // POOL: its first pixel; tile t starts at mbase + t * TP
    auto coords = [&](int tile, Pix &c) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int m = mbase + tile * TP + wave * (32 * NA) + a * 32 + li;
            c.rv[a] = m < mlim;
            const int mm = c.rv[a] ? m : 0;
            const int n = mm / ohw, r = mm - n * ohw;
            const int oh = r / p.OW, ow = r - oh * p.OW;
            c.ih0[a] = 2 * oh - p.pad_t;
            c.iw0[a] = 2 * ow - p.pad_l + kh;             // pixel 2 j + kh of the kernel row
            c.nb[a] = n * p.H;
        }
    };
    auto row_offsets_of = [&](const Pix &c, int dh, unsigned (&vo)[NA][4]) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int ih = c.ih0[a] + dh;
            const bool rok = c.rv[a] && (unsigned)ih < (unsigned)p.H;
            const int base = (c.nb[a] + ih) * p.W;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = c.iw0[a] + 2 * j;
                const bool ok = rok && (unsigned)iw < (unsigned)p.W && (2 * j + kh) < 7;
                vo[a][j] = ok ? (unsigned)(base + iw) * 12u : kOOB;
            }
        }
    };
    // store + statistics: element e of a block is output row (e&3) + 8 (e>>2) + 4 kh, column li
    // (stores through a buffer descriptor: one 32-bit lane offset per block + the row's offset, rows past M dropped by the
    // hardware range check -- as gemm_wide_kernel's epilogue; the second 32 columns ride in the instruction offset)
    auto epilogue = [&](int tile, const f32x16 (&acc)[NA][2]) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int row0 = tile * TP + wave * (32 * NA) + a * 32 + 4 * kh;
            const unsigned vz = (unsigned)(row0 * p.ldz + li) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (e & 3) + 8 * (e >> 2);
                const float v0 = acc[a][0][e], v1 = acc[a][1][e];
                const unsigned vo = vz + (unsigned)(((e & 3) + 8 * (e >> 2)) * p.ldz * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), srd_z, vo, 0, 2 /* nt */);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), srd_z, vo, 128, 2 /* nt */);
                if (row < p.M) {
                    const float u0 = v0 - pv0, u1 = v1 - pv1;
                    s0 += u0; q0 += u0 * u0;
                    s1 += u1; q1 += u1 * u1;
                }
            }
        }
    };

    // ---- POOL: this workgroup's row pairs, the ring, the pooled epilogue ------------------------------------------------
    int rp0 = 0, rp1 = 0, qf = 0, ntiles = 0, stat_end = 0;
    if constexpr (POOL) {
        rp0 = (int)((int64_t)blockIdx.x * p.rp_total / gridDim.x);
        rp1 = (int)((int64_t)(blockIdx.x + 1) * p.rp_total / gridDim.x);
        mbase = rp0 * 2 * p.OW;
        stat_end = rp1 * 2 * p.OW;
        mlim = stat_end + ((rp1 % p.PH) != 0 ? p.OW : 0);          // + conv row 2 rp1, the last pooled row's third row
        ntiles = (mlim - mbase + TP - 1) / TP;
        qf = rp0;
    }
    // pooled row q (row pair index over the batch) lives in ring slot q % nslots; out: [N * PH][PW][64] = max of z
    int qf_slot = POOL ? qf % p.nslots : 0;
    int qf_i = POOL ? qf % p.PH : 0;      // row qf's index inside its image
    auto flush_row = [&](int q) {
        float *slot = ring_s + qf_slot * (p.PW * CO);
        if (++qf_slot == p.nslots) qf_slot = 0;
        float *dst = p.z + (int64_t)q * p.PW * p.ldz;
        const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int i = tid; i < p.PW * (CO / 4); i += 256) {
            const int j = i >> 4, c4 = (i & 15) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(slot + j * CO + c4);
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst + (int64_t)j * p.ldz + c4));
            *reinterpret_cast<f32x4 *>(slot + j * CO + c4) = ninf;
        }
    };
    // accumulators -> statistics (pixels of the own row pairs only) and the ring: conv pixel (row R of the batch, column c)
    // lies in the windows of pooled rows R / 2 and -- R even, not the first row of its image -- R / 2 - 1, pooled columns
    // c / 2 and -- c even, c > 0 -- c / 2 - 1.  Branch-free: a window that does not apply is replaced by one that does (the
    // maximum is idempotent), a pixel past the range by -inf; no division in here (the lane's position advances by a fixed
    // step per tile).
    struct Pos { int R, c, slot, i; };          // first pixel of the lane's block: conv row of the batch, column; slot of its
    Pos pos[NA];                                // pooled row R / 2 and that row's index inside its image
    int cstep = 0, rstep = 0;
    if constexpr (POOL) {
        cstep = TP % p.OW;
        rstep = TP / p.OW;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int mb = mbase + wave * (32 * NA) + a * 32 + 4 * kh;
            pos[a].R = mb / p.OW;
            pos[a].c = mb - pos[a].R * p.OW;
            pos[a].slot = (pos[a].R >> 1) % p.nslots;
            pos[a].i = (pos[a].R >> 1) % p.PH;
        }
    }
    auto pool_epilogue = [&](int tile, const f32x16 (&acc)[NA][2]) {
        // Per QUAD of consecutive pixels (accumulator elements 4 g .. 4 g + 3: columns c .. c + 3 of one conv row -- OW and
        // every block start are multiples of 4, so c is too): pooled column c / 2 takes max(v0, v1, v2), column c / 2 + 1
        // max(v2, v3), column c / 2 - 1 takes v0 -- three LDS maxima per pooled row instead of six, one position per quad.
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int mb = mbase + tile * TP + wave * (32 * NA) + a * 32 + 4 * kh;      // this lane's first pixel of the block
            const int Rb = pos[a].R, cb = pos[a].c;
            const int qb = Rb >> 1;
            // (the lane's sixteen pixels span at most three conv rows = pooled rows qb, qb + 1: OW >= 16)
            const int sl0 = pos[a].slot, sl1 = (sl0 + 1 == p.nslots) ? 0 : sl0 + 1, slm = (sl0 == 0) ? p.nslots - 1 : sl0 - 1;
            const bool top0 = pos[a].i == 0, top1 = pos[a].i + 1 == p.PH;               // first row pair of an image
            float *const rb0 = ring_s + sl0 * (p.PW * CO) + li, *const rb1 = ring_s + sl1 * (p.PW * CO) + li;
            float *const rbm = ring_s + slm * (p.PW * CO) + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = mb + 8 * g;
                int c = cb + 8 * g, R = Rb;
                const bool w1 = c >= p.OW;
                c -= w1 ? p.OW : 0; R += w1 ? 1 : 0;
                const bool w2 = c >= p.OW;
                c -= w2 ? p.OW : 0; R += w2 ? 1 : 0;
                const bool valid = m < mlim, instat = m < stat_end;       // (both limits are multiples of 4: whole quads)
                const int q = R >> 1;
                const bool second = q != qb;                              // pooled row qb + 1
                const bool own = q < rp1;                                 // (the row behind the range only feeds row q - 1)
                const bool up = !(R & 1) && !(second ? top1 : top0) && q > rp0;
                const int jo = (c >> 1) * CO;
                float *const t_own = (second ? rb1 : rb0) + jo, *const t_up = (second ? rb0 : rbm) + jo;
                float *A = own ? t_own : t_up;
                float *Bq = up ? t_up : A;
                A = valid ? A : ring_s + li;
                Bq = valid ? Bq : ring_s + li;
                const int lo = (valid && c > 0) ? -CO : 0;               // pooled column c / 2 - 1, or c / 2 again (idempotent)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const float v0 = acc[a][nb][4 * g], v1 = acc[a][nb][4 * g + 1], v2 = acc[a][nb][4 * g + 2], v3 = acc[a][nb][4 * g + 3];
                    const float pv = nb ? pv1 : pv0;
                    const float u0 = v0 - pv, u1 = v1 - pv, u2 = v2 - pv, u3 = v3 - pv;
                    const float su = (u0 + u1) + (u2 + u3), qu = (u0 * u0 + u1 * u1) + (u2 * u2 + u3 * u3);
                    if (nb) { s1 += instat ? su : 0.f; q1 += instat ? qu : 0.f; }
                    else { s0 += instat ? su : 0.f; q0 += instat ? qu : 0.f; }
                    const float h0 = valid ? fmaxf(fmaxf(v0, v1), v2) : -INFINITY, h1 = valid ? fmaxf(v2, v3) : -INFINITY;
                    const float h2 = valid ? v0 : -INFINITY;
                    ds_fmax(A + 32 * nb, h0); ds_fmax(A + CO + 32 * nb, h1); ds_fmax(A + lo + 32 * nb, h2);
                    ds_fmax(Bq + 32 * nb, h0); ds_fmax(Bq + CO + 32 * nb, h1); ds_fmax(Bq + lo + 32 * nb, h2);
                }
            }
            // the block's position in the next tile (conditional subtractions: stem_pool_ok bounds the steps)
            int c2 = cb + cstep, R2 = Rb + rstep;
            const bool w = c2 >= p.OW;
            c2 -= w ? p.OW : 0; R2 += w ? 1 : 0;
            const int dq = (R2 >> 1) - qb;
            int sl = sl0 + dq, ii = pos[a].i + dq;
            sl -= sl >= p.nslots ? p.nslots : 0;
            sl -= sl >= p.nslots ? p.nslots : 0;
            ii -= ii >= p.PH ? p.PH : 0;
            ii -= ii >= p.PH ? p.PH : 0;
            pos[a].R = R2; pos[a].c = c2; pos[a].slot = sl; pos[a].i = ii;
        }
    };

    // after the barrier behind a tile's epilogue: the pooled rows whose last conv row has passed leave the ring
    auto flush_done_rows = [&](int tile) {
        const int done = min(mbase + (tile + 1) * TP, mlim);
        while (qf < rp1) {                            // (uniform)
            const int last = (qf_i == p.PH - 1) ? 2 : 3;
            if ((2 * qf + last) * p.OW > done) break;
            flush_row(qf);
            ++qf;
            if (++qf_i == p.PH) qf_i = 0;
        }
    };

    if constexpr (BF) {
        // Row pair t = kernel rows 2 t, 2 t + 1 (row 7 does not exist: zeros against zero weights).  Two pairs of requests are
        // always in flight -- across tiles too: the first two pairs of the NEXT tile are requested before this tile's last
        // MFMAs and stores -- and a pair is packed to bf16 fragments (24 registers) as soon as it has arrived.
        f32x3 rawA[2][NA][4], rawB[2][NA][4];          // [row of the pair][row block a][pixel j]
        auto issue = [&](const Pix &c, int t, f32x3 (&dst)[2][NA][4]) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (2 * t + r >= 7) continue;
                unsigned vo[NA][4];
                row_offsets_of(c, 2 * t + r, vo);
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        dst[r][a][j] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(srd, vo[a][j], 0, 0));
            }
        };
        bf16x8 pk[NA][3];
        auto pack = [&](int t, const f32x3 (&src)[2][NA][4]) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int ml = 0; ml < 3; ++ml) {
                    float v[8];
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) {
                        const int q = ml * 8 + sl, r = q / 12, jj = (q - 12 * r) / 3, c = q % 3;
                        v[sl] = (2 * t + r < 7) ? src[r][a][jj][c] : 0.f;
                    }
                    const bf16x2 p0 = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), p1 = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
                    const bf16x2 p2 = __builtin_convertvector(f32x2{v[4], v[5]}, bf16x2), p3 = __builtin_convertvector(f32x2{v[6], v[7]}, bf16x2);
                    pk[a][ml] = bf16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
                }
        };
        const __bf16 *wb = reinterpret_cast<const __bf16 *>(wl) + (kh * CO + li) * 8;
        f32x16 acc[NA][2];
        auto mfmas = [&](int t, bool first) {
#pragma unroll
            for (int ml = 0; ml < (t < 3 ? 3 : 2); ++ml) {
                const int m = 3 * t + ml;
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(wb + (m * 2 * CO) * 8);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(wb + (m * 2 * CO + 32) * 8);
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pk[a][ml], b0, (first && ml == 0) ? zero16 : acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pk[a][ml], b1, (first && ml == 0) ? zero16 : acc[a][1], 0, 0, 0);
                }
            }
        };
        Pix cu, nx;
        // (POOL: this workgroup's own tiles 0 .. ntiles - 1 of its row-pair range instead of a stride over the batch)
        const int tend = POOL ? ntiles : p.tiles, tstep = POOL ? 1 : (int)gridDim.x;
        int tile = POOL ? 0 : (int)blockIdx.x;
        if (tile < tend) {
            coords(tile, cu);
            issue(cu, 0, rawA);
            issue(cu, 1, rawB);
        }
        for (; tile < tend; tile += tstep) {
            const int ntile = tile + tstep;
            const bool more = ntile < tend;          // (uniform)
            if (more) coords(ntile, nx);
            pack(0, rawA);
            issue(cu, 2, rawA);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0, true);
            pack(1, rawB);
            issue(cu, 3, rawB);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1, false);
            pack(2, rawA);
            if (more) issue(nx, 0, rawA);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(2, false);
            pack(3, rawB);
            if (more) issue(nx, 1, rawB);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(3, false);
            if constexpr (POOL) {
                if (tile > 0) __syncthreads();                // every wave is through the previous tile's row flush
                pool_epilogue(tile, acc);
                __syncthreads();                              // the tile's maxima are in the ring
                flush_done_rows(tile);
            } else {
                epilogue(tile, acc);
            }
            cu = nx;
        }
        if constexpr (POOL) __syncthreads();                  // `red` below lies over the ring
    } else {
        // fp32: with two waves per SIMD the vector instructions between the MFMAs add to the matrix time (ablation builds,
        // profiles/r05_notes.md: the launch without any memory access still took 596 us against 392 us of matrix cycles), so
        // the per-tile index math is kept small: column offsets / validity once per tile, the kernel-row loop unrolled with
        // ping-pong registers (no copies), the first MFMA of every accumulator with C = 0, and for tiles that lie wholly
        // inside M the store row offsets in the instruction's scalar offset.
        struct Cols { unsigned o[2][4]; };          // byte offset of pixel 2 j + kh inside its image row, or out of range
        auto setup = [&](int tile, Pix &c, Cols &co) {
            coords(tile, c);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int iw = c.iw0[a] + 2 * j;
                    co.o[a][j] = (c.rv[a] && (unsigned)iw < (unsigned)p.W && (2 * j + kh) < 7) ? (unsigned)iw * 12u : kOOB;
                }
        };
        f32x3 buf[2][2][4];
        auto issue = [&](const Pix &c, const Cols &co, int dh, f32x3 (&dst)[2][4]) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ih = c.ih0[a] + dh;
                // (an out-of-range column offset stays out of range with the row offset added: both < 2^31)
                const unsigned rowo = (unsigned)ih < (unsigned)p.H ? (unsigned)((c.nb[a] + ih) * p.W) * 12u : kOOB;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dst[a][j] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(srd, (rowo | co.o[a][j]) & kOOB ? kOOB : rowo + co.o[a][j], 0, 0));
            }
        };
        // the B fragments of a kernel row (12 K steps x 2 column blocks) are read one row AHEAD: read just in time, every
        // group of four MFMAs waited for its own ds_read (one MFMA of cover for ~100 cycles of LDS latency)
        float bw[2][12][2];
        auto read_b = [&](int dh, float (&dst)[12][2]) {
            const float *wr = wl + dh * WROW + kh * CO + li;
            const float *w3 = wl + dh * WROW + li;          // POOL, j = 3: the kh = 0 rows for both halves (kh = 1: A = 0)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (POOL && k >= 9) {
                    dst[k][0] = w3[(9 + k) * CO];
                    dst[k][1] = w3[(9 + k) * CO + 32];
                } else {
                    dst[k][0] = wr[k * 2 * CO];
                    dst[k][1] = wr[k * 2 * CO + 32];
                }
            }
        };
        // One tile (kernel row dh in buf[dh & 1]).  Requesting the NEXT tile's first row behind this tile's last one -- in front of
        // the 128 stores -- was built (two copies of this body, the parity alternates with seven rows) and bought nothing: 677 us.
        Pix cN;                 // POOL: the next tile's coordinates (set inside run_tile, ahead of the epilogue)
        Cols oN;
        auto run_tile = [&](int tile, const Pix &c, const Cols &co) {
            constexpr int P = 0;
            f32x16 acc[2][2];
            read_b(0, bw[0]);
#pragma unroll
            for (int dh = 0; dh < 7; ++dh) {
                if (dh < 6) {
                    issue(c, co, dh + 1, buf[(dh + 1 + P) & 1]);
                    read_b(dh + 1, bw[(dh + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const f32x3 (&cur)[2][4] = buf[(dh + P) & 1];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const float b0 = bw[dh & 1][j * 3 + cc][0], b1 = bw[dh & 1][j * 3 + cc][1];
                        if (dh == 0 && j == 0 && cc == 0) {
                            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0][j][cc], b0, zero16, 0, 0, 0);
                            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[1][j][cc], b0, zero16, 0, 0, 0);
                            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0][j][cc], b1, zero16, 0, 0, 0);
                            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[1][j][cc], b1, zero16, 0, 0, 0);
                        } else {
                            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0][j][cc], b0, acc[0][0], 0, 0, 0);
                            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[1][j][cc], b0, acc[1][0], 0, 0, 0);
                            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0][j][cc], b1, acc[0][1], 0, 0, 0);
                            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[1][j][cc], b1, acc[1][1], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (POOL) {
                // the next tile's first kernel row is requested BEFORE the epilogue: its latency passes under the ring work
                if (tile + 1 < ntiles) {
                    setup(tile + 1, cN, oN);
                    issue(cN, oN, 0, buf[0]);
                }
                if (tile > 0) __syncthreads();                // every wave is through the previous tile's row flush
                pool_epilogue(tile, acc);
            } else if ((tile + 1) * TP <= p.M) {          // (uniform) every row of the tile exists
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const unsigned vz = (unsigned)((tile * TP + wave * 64 + a * 32 + 4 * kh) * p.ldz + li) * 4u;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v0 = acc[a][0][e], v1 = acc[a][1][e];
                        const int so = ((e & 3) + 8 * (e >> 2)) * p.ldz * 4;          // (rows inside M: the unchecked scalar offset is safe)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), srd_z, vz, so, 2 /* nt */);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), srd_z, vz + 128u, so, 2 /* nt */);
                        const float u0 = v0 - pv0, u1 = v1 - pv1;          // (the arithmetic of `epilogue`, to the bit)
                        s0 += u0; q0 += u0 * u0;
                        s1 += u1; q1 += u1 * u1;
                    }
                }
            } else {
                epilogue(tile, acc);
            }
        };
        if constexpr (POOL) {
            Pix cA;
            Cols oA;
            if (ntiles > 0) {
                setup(0, cA, oA);
                issue(cA, oA, 0, buf[0]);
            }
            for (int tile = 0; tile < ntiles; ++tile) {
                run_tile(tile, cA, oA);
                __syncthreads();                              // the tile's maxima are in the ring
                flush_done_rows(tile);
                // (the barrier that keeps the next tile's maxima off the rows being reset sits in front of its epilogue, a whole
                // K loop away: nobody waits there)
                cA = cN;
                oA = oN;
            }
            __syncthreads();                                  // `red` below lies over the ring
        } else {
            for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
                Pix cA;
                Cols oA;
                setup(tile, cA, oA);
                issue(cA, oA, 0, buf[0]);
                run_tile(tile, cA, oA);
            }
        }
    }
    if (p.stats) {
        s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32);
        s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
        if (kh == 0) {
            red[(wave * CO + li) * 2] = s0;       red[(wave * CO + li) * 2 + 1] = q0;
            red[(wave * CO + 32 + li) * 2] = s1;  red[(wave * CO + 32 + li) * 2 + 1] = q1;
        }
        __syncthreads();
        if (tid < CO) {
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                ss += red[(w * CO + tid) * 2];
                qq += red[(w * CO + tid) * 2 + 1];
            }
            p.stats[(int64_t)tid * gridDim.x + blockIdx.x] = ss;
            p.stats[((int64_t)CO + tid) * gridDim.x + blockIdx.x] = qq;
        }
    }
}

int stem_grid(int64_t M, bool bf = false) {
    const int64_t tiles = bf ? (M + 127) / 128 : (M + 255) / 256;
    const int64_t g = (bf ? 3 : 2) * ds::kCUs;
    return (int)(tiles < g ? tiles : g);
}

// POOL: one workgroup per contiguous range of conv row pairs, two per CU (never more workgroups than row pairs)
int stem_pool_grid(int64_t row_pairs) {
    const int64_t g = 2 * ds::kCUs;
    return (int)(row_pairs < g ? row_pairs : g);
}

// the pooled stem needs: an even conv map (the 3x3 / 2 SAME pool then pads bottom / right only), rows of 16 .. 112 pixels, a
// multiple of 4 (a lane's sixteen pixels span at most three rows, its quads one; a pooled row fits a ring slot), and enough
// ring slots for a tile
bool stem_pool_ok(int OH, int OW) {
    // (OW % 4: a lane's four consecutive pixels share a conv row; OH >= 10: the per-tile position update, pool_epilogue)
    if (OH < 10 || OW < 16 || (OH & 1) || (OW & 3) || OW > 112) return false;
    // a 256-pixel tile spans at most R = 254 / OW + 2 conv rows, which lie in at most (R - 1) / 2 + 2 pooled rows -- the rows
    // that are live in the ring while the tile is folded in (complete rows leave at the end of every tile)
    const int PW = OW / 2, nslots = kRingFloats / (PW * CO), R = 254 / OW + 2;
    return nslots >= (R - 1) / 2 + 2;
}

}  // namespace

extern "C" int ds_conv_stem_pool_supported(int32_t H, int32_t W) { return stem_pool_ok((H + 1) / 2, (W + 1) / 2) ? 1 : 0; }
extern "C" int ds_conv_stem_pool_partials(int32_t N, int32_t OH, int32_t OW) { return stem_pool_grid((int64_t)N * (OH / 2)); }

extern "C" int ds_conv_stem_partials(int32_t N, int32_t OH, int32_t OW) { return stem_grid((int64_t)N * OH * OW); }
extern "C" int ds_conv_stem_bf16_partials(int32_t N, int32_t OH, int32_t OW) { return stem_grid((int64_t)N * OH * OW, true); }

namespace {
int stem_launch(bool bf, const float *x, const float *w, float *z, float *stats, const float *pivot, int32_t N, int32_t H,
                int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream) {
    DS_REQUIRE(x && w && z, "ds_conv_stem: null argument");
    DS_REQUIRE(Cout == CO && (cin_store == 3 || cin_store == 4) && ldz >= CO, "ds_conv_stem: the 7x7/2 stem has 3 input and 64 output channels");
    StemParams p = {};
    p.x = x; p.w = w; p.z = z; p.stats = stats; p.pivot = stats ? pivot : nullptr;
    p.N = N; p.H = H; p.W = W;
    p.OH = (H + 1) / 2; p.OW = (W + 1) / 2;
    const int ph = (p.OH - 1) * 2 + 7 - H, pw = (p.OW - 1) * 2 + 7 - W;     // TF SAME: the smaller half in front
    p.pad_t = (ph > 0 ? ph : 0) / 2; p.pad_l = (pw > 0 ? pw : 0) / 2;
    p.cin_store = cin_store; p.ldz = ldz;
    const int64_t M = (int64_t)N * p.OH * p.OW;
    const int64_t xb = (int64_t)N * H * W * 12;
    DS_REQUIRE(M < (1ll << 31) && xb < (1ll << 31) && ((M - 1) * ldz + CO) * 4 < (1ll << 31),
               "ds_conv_stem: input or output larger than 2 GiB (split the batch)");
    p.M = (int)M; p.tiles = (int)(bf ? (M + 127) / 128 : (M + 255) / 256);
    p.x_bytes = (unsigned)xb;
    if (bf) hipLaunchKernelGGL(conv_stem_kernel<true>, dim3(stem_grid(M, true)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv_stem_kernel<false>, dim3(stem_grid(M)), dim3(256), 0, (hipStream_t)stream, p);
    return ds::check_launch(bf ? "ds_conv_stem_bf16" : "ds_conv_stem");
}
}  // namespace

extern "C" int ds_conv_stem(const float *x, const float *w, float *z, float *stats, const float *pivot, int32_t N, int32_t H,
                            int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream) {
    return stem_launch(false, x, w, z, stats, pivot, N, H, W, cin_store, Cout, ldz, stream);
}

// Conv2d_1a_7x7 + MaxPool_2a_3x3 (inception_v1.py:63-67) in one launch: zmax [N, OH/2, OW/2, 64] (pixel stride ldz) = the 3x3 / 2
// SAME maximum of the conv output z, which is not written; stats = the column sums of the FULL map about the pivot,
// float[2][64][ds_conv_stem_pool_partials(N, OH, OW)].  relu(rstd * zmax + shift) is the pooled activation (rstd > 0).
namespace {
int stem_pool_launch(bool bf, const float *x, const float *w, float *zmax, float *stats, const float *pivot, int32_t N,
                     int32_t H, int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream) {
    DS_REQUIRE(x && w && zmax, "ds_conv_stem_pool: null argument");
    DS_REQUIRE(Cout == CO && (cin_store == 3 || cin_store == 4) && ldz >= CO && ldz % 4 == 0 && (((uintptr_t)zmax) & 15) == 0,
               "ds_conv_stem_pool: the 7x7/2 stem has 3 input and 64 output channels (16-byte aligned output rows)");
    StemParams p = {};
    p.x = x; p.w = w; p.z = zmax; p.stats = stats; p.pivot = stats ? pivot : nullptr;
    p.N = N; p.H = H; p.W = W;
    p.OH = (H + 1) / 2; p.OW = (W + 1) / 2;
    DS_REQUIRE(stem_pool_ok(p.OH, p.OW), "ds_conv_stem_pool: conv map %d x %d (even sizes, 16 .. 112 columns: ds_conv_stem_pool_supported)", p.OH, p.OW);
    const int ph = (p.OH - 1) * 2 + 7 - H, pw = (p.OW - 1) * 2 + 7 - W;     // TF SAME: the smaller half in front
    p.pad_t = (ph > 0 ? ph : 0) / 2; p.pad_l = (pw > 0 ? pw : 0) / 2;
    p.cin_store = cin_store; p.ldz = ldz;
    const int64_t M = (int64_t)N * p.OH * p.OW;
    const int64_t xb = (int64_t)N * H * W * 12;
    DS_REQUIRE(M + p.OW < (1ll << 31) && xb < (1ll << 31), "ds_conv_stem_pool: input or output larger than 2 GiB (split the batch)");
    p.M = (int)M; p.tiles = 0;
    p.x_bytes = (unsigned)xb;
    p.PH = p.OH / 2; p.PW = p.OW / 2;
    p.rp_total = N * p.PH;
    p.nslots = kRingFloats / (p.PW * CO);
    if (bf) hipLaunchKernelGGL((conv_stem_kernel<true, true>), dim3(stem_pool_grid(p.rp_total)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_stem_kernel<false, true>), dim3(stem_pool_grid(p.rp_total)), dim3(256), 0, (hipStream_t)stream, p);
    return ds::check_launch(bf ? "ds_conv_stem_pool_bf16" : "ds_conv_stem_pool");
}
}  // namespace

extern "C" int ds_conv_stem_pool(const float *x, const float *w, float *zmax, float *stats, const float *pivot, int32_t N,
                                 int32_t H, int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream) {
    return stem_pool_launch(false, x, w, zmax, stats, pivot, N, H, W, cin_store, Cout, ldz, stream);
}

// ... and of ds_conv_stem_bf16 (the 16-bit configurations): operands rounded to bf16, bf16 MFMA, fp32 maxima
extern "C" int ds_conv_stem_pool_bf16(const float *x, const float *w, float *zmax, float *stats, const float *pivot, int32_t N,
                                      int32_t H, int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream) {
    return stem_pool_launch(true, x, w, zmax, stats, pivot, N, H, W, cin_store, Cout, ldz, stream);
}

// The same layer for the 16-bit configurations: x and w rounded to bf16 (RNE), v_mfma_f32_32x32x16_bf16, fp32 accumulation.
extern "C" int ds_conv_stem_bf16(const float *x, const float *w, float *z, float *stats, const float *pivot, int32_t N, int32_t H,
                                 int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream) {
    return stem_launch(true, x, w, z, stats, pivot, N, H, W, cin_store, Cout, ldz, stream);
}
