// 3x3 stride-1 SAME convolution (forward and dgrad) as fused Winograd F(4x4, 3x3) on the fp32 matrix cores: Conv2d_2c_3x3
// at 56 x 56, the Branch_1 / Branch_2 3x3 layers of Mixed_3b / 3c at 28 x 28 (image_model/inception_v1.py:74-75,
// :86-115) and those 14 x 14 / 7 x 7 layers (:122-247) whose workgroup count suits it (w4_choose below).
//
// F(4x4, 3x3) trades the 144 multiplies of a 4x4 output tile and channel pair for 36 (F(2x2, 3x3), conv_wino.hip: 64):
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        (Lavin & Gray, interpolation points 0, +-1, +-2, inf)
// i.e. thirty-six independent GEMMs  M_xi[tile, co] = sum_ci V_xi[tile, ci] U_xi[ci, co]  -- 4x fewer MFMA passes than
// the direct convolution, 1.78x fewer than F(2x2, 3x3).  fp32 MFMA runs at the fp32 vector rate, so the multiplies
// saved are the time saved as long as the transforms stay small next to them.
//
// 36 accumulators of 32 x 32 do not fit one wave (576 registers), so unlike conv_wino.hip the positions are SPLIT over
// the four waves of a workgroup (nine each) and the transformed input goes through LDS once:
//   * workgroup = 32 tiles (4x4 outputs each; partial border tiles for any H, W) x 32 NB output channels (NB = 1, 2),
//     K step = 16 input channels;
//   * transform role: thread (tile t = tid / 8, channel pair c = tid % 8) loads the 36 pixels of its 6x6 patch for TWO
//     channels (8-byte SRD loads: the eight lanes of a pixel are one 64-byte run; pixel offsets are wave-uniform and
//     ride in the scalar offset, padding pixels read zeros from an out-of-range vector offset), runs B^T d B in
//     registers as 144 v_pk_fma_f32 and writes the 36 results to V[xi][t][2c..] in LDS (lane-linear, conflict free);
//   * matrix role: wave w owns positions 9 w .. 9 w + 8.  Its A fragments are one ds_read_b128 per position and
//     8-channel half step (lane (i, kh): tile i, channels 4 kh .. 4 kh + 3); its B fragments -- U for ITS positions
//     only, nothing another wave needs -- come straight from global memory into registers, one fully coalesced 1 KB
//     load per position and channel block (U is stored [36][Cin / 8][Cout][8] by ds_wino4_transform_weights for exactly
//     this), requested six positions (3072 matrix cycles) ahead into a register ring.  9 NB accumulators of 32 x 32
//     per wave: at NB = 2 that is 288 registers against 256 accumulation registers, so position 8's pair lives in the
//     architectural file (mfma_v);
//   * software pipeline: V is double buffered and the transform of K step k + 1 runs in twelve chunks BETWEEN the MFMA
//     groups of step k, the pixels of step k + 2 are requested behind it into the registers it freed: one barrier per
//     K step, no load / LDS / dependency latency in front of the matrix pipe (the VALU issue time itself still adds:
//     measured 5.7 us per step against 4.1 us of MFMAs alone);
//   * epilogue, per channel block: the waves park their accumulators in LDS as M[xi][tile][co] (144 KB -- gfx950's
//     160 KB LDS, the V buffers are dead by then), then thread (tile = tid / 8, channel quad = tid % 8) gathers its 36
//     positions with 16-byte reads, runs A^T M A on four channels at once and stores every output pixel as 16 bytes
//     (eight lanes = one 128-byte run); BatchNorm column statistics about the pivot (DS_EPI_STATS) or the
//     BatchNorm-backward sums of the consumer (DS_EPI_BNSUMS) as in conv_wino.hip.
// dgrad: the same kernel with U built from the flipped, transposed filter.
// Numerics: fp32 throughout, ordered reductions (deterministic).  The F(4x4) transforms carry constants up to 8 and
// cost about one decimal digit against F(2x2): relative rms error 2.4e-6 instead of 3.7e-7 on post-ReLU activations
// with 192 input channels (scripts/microbench/wino4_numerics.py), against 2.6e-7 for a direct fp32 sum.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOOB = 0x80000000u;

struct Wino4Params {
    const float *x;         // [N, H, W, ldx]
    const float *u;         // [36][Cin / 8][Cout][8]
    float *z;               // [N, H, W, ldz]
    float *stats;           // [2][Cout][P], P = groups
    const float *pivot;
    const float *y;         // DS_EPI_BNSUMS: forward activation of the layer that consumes z (= dy), pixel stride ldz
    int N, H, W, Cin, ldx, Cout, ldz;
    int TH, TW, Mt;         // output tiles per column / row / in total
    int groups, ncol;       // 32-tile groups, (32 NB)-channel blocks (x ksplit: a workgroup per block AND reduction slice)
    int ksplit, kchunk;     // split K (ds_conv_wino4_splitk): reduction slices per output block, K steps per slice; 1 = off
    long long zslab;        // ... floats between the slices' partial outputs (z then points at slab 0, pixel stride ldz)
    unsigned x_bytes, u_bytes, z_bytes;
    int flags;
#ifdef DS_W4_PROF
    unsigned long long *prof;      // [workgroup][16] s_memtime stamps (scripts/wino4_phase_prof.py builds this variant)
#endif
};

// Phase profiler (only in the -DDS_W4_PROF build of scripts/wino4_phase_prof.py; the library build compiles none of it):
// wave 0 of every workgroup stamps s_memtime at its phase boundaries (scalar registers) and lane 0 stores the stamps last.
// DS_W4_ABL (same build) compiles pieces of the K loop out: 1 weight-fragment loads, 2 pixel loads, 4 transform VALU,
// 8 LDS writes of V.
#ifdef DS_W4_PROF
#define W4_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)"); prof_t[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#endif
#ifndef DS_W4_ABL
#define DS_W4_ABL 0
#endif
// tuning knobs (compile time; the values below are the measured choice, scripts/wino4_phase_prof.py --def sweeps them)
#ifndef DS_W4_PIX
#define DS_W4_PIX 0
#endif
#ifndef DS_W4_RING1
#define DS_W4_RING1 6
#endif
#ifndef DS_W4H_FIRST
#define DS_W4H_FIRST 1        // AR: C = 0 in the first MFMA of every accumulator (0: zero the accumulators up front)
#endif
#ifndef DS_W4H_UNILOOP
#define DS_W4H_UNILOOP (NB == 2)      // AR: no separate last K step (see the K loop); measured: NB = 2 -7 .. -15 %, NB = 1 +2 .. 4 us
#endif
#ifndef DS_W4H_RP2
#define DS_W4H_RP2 3          // AR, NB = 2: positions of weight fragments in flight
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4srd(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// AR = 1 (ds_conv_wino4_bf16x2): a channel pair rounded to bf16 (RNE, one v_cvt_pk_bf16_f32) and the pair back in fp32
__device__ __forceinline__ unsigned pk_bf16(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 unpk_bf16(unsigned h) {
    return f32x2{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() is a release fence + s_barrier and therefore waits for
// vmcnt(0) as well: in the epilogue that is the drain of the 64 output stores every thread has just issued (measured:
// the second channel block's pass cost 9 us instead of ~3), in the K loop the prefetches in flight.  The exchanges here
// go through LDS alone, so only the LDS counter has to be zero.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The input transform works on channel PAIRS with packed fp32 instructions.  Left to hipcc, the subtractions and the
// multiply-adds whose constant is not an inline one (-5) came out as two scalar instructions each (72 v_fma_f32 + 40
// v_add_f32 beside 88 packed ones per K step), and with one wave per SIMD every VALU issue slot is a slot the matrix pipe
// idles: both forms are pinned here (constant as an SGPR pair, negation as the packed source modifier).
__device__ __forceinline__ f32x2 pfma(float k, f32x2 a, f32x2 b) {       // k * a + b
    f32x2 r;
    const f32x2 kk = {k, k};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(kk), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 padd(f32x2 a, f32x2 b) {                // a + b
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 psub(f32x2 a, f32x2 b) {                // a - b
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// v_mfma_f32_32x32x2_f32 with the accumulator in architectural vector registers
__device__ __forceinline__ void mfma_v(f32x16 &c, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// ... the first one of an accumulator: C = 0
__device__ __forceinline__ void mfma_v0(f32x16 &c, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}

// v_mfma_f32_32x32x16_bf16 with the accumulator in architectural vector registers (operands: eight bf16 = four registers)
__device__ __forceinline__ void mfma_hv(f32x16 &c, f32x4 a, f32x4 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_hv0(f32x16 &c, f32x4 a, f32x4 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}

// B^T d for one line of six: B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void in1d(f32x2 d0, f32x2 d1, f32x2 d2, f32x2 d3, f32x2 d4, f32x2 d5, f32x2 &t0, f32x2 &t1,
                                     f32x2 &t2, f32x2 &t3, f32x2 &t4, f32x2 &t5) {
    // (six independent first-level results, then six that use them: a packed op right behind the one it depends on costs
    // a wait state -- 39 s_nop per K step in the first version of this order)
    const f32x2 a = pfma(-4.f, d2, d4), b = pfma(-4.f, d1, d3);
    const f32x2 c = psub(d4, d2), e = psub(d3, d1);
    const f32x2 i0 = pfma(-5.f, d2, d4), i5 = pfma(-5.f, d3, d5);
    t1 = padd(a, b);
    t2 = psub(a, b);
    t3 = pfma(2.f, e, c);
    t4 = pfma(-2.f, e, c);
    t0 = pfma(4.f, d0, i0);
    t5 = pfma(4.f, d1, i5);
}

// A^T m for one line of six, on four channels at once: A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
// Every operation is a packed fp32 one on the channel pairs.  hipcc packs additions and multiply-adds but not subtractions
// (and it folds fma(-1, b, a) back into a subtraction): the four differences per line came out as four scalar v_sub_f32
// each, 288 of the epilogue's ~1200 VALU instructions at NB = 2, and every VALU slot is a slot the matrix pipe idles.  So
// a - b is written fma(m1, b, a) with m1 = -1 hidden from the optimiser in a scalar register (same value, one rounding).
// NOT inline asm as in the input transform: these results share registers with the data of the output stores just issued,
// and the gfx940 "VALU overwrites the data of a >64-bit store" wait states are not inserted in front of inline asm
// (measured: one corrupted channel per store).
__device__ __forceinline__ f32x4 qfma(f32x4 k, f32x4 a, f32x4 b) { return __builtin_elementwise_fma(k, a, b); }
__device__ __forceinline__ f32x4 q4(float k) { return f32x4{k, k, k, k}; }
__device__ __forceinline__ void out1d(float m1, f32x4 m0, f32x4 a1, f32x4 a2, f32x4 a3, f32x4 a4, f32x4 m5, f32x4 &y0, f32x4 &y1,
                                      f32x4 &y2, f32x4 &y3) {
    const f32x4 s0 = a1 + a2, s1 = qfma(q4(m1), a2, a1), s2 = a3 + a4, s3 = qfma(q4(m1), a4, a3);
    y0 = s2 + (s0 + m0);
    y1 = qfma(q4(2.f), s3, s1);
    y2 = qfma(q4(4.f), s2, s0);
    y3 = m5 + qfma(q4(8.f), s3, s1);
}

// AR = 1 (ds_conv_wino4_bf16x2, the 16-bit configurations): the thirty-six GEMMs on the bf16 matrix cores.  The operands of
// the CONVOLUTION are the bf16-rounded ones the configuration defines (x rounded as it is loaded, the filter rounded before
// G g G^T), but the Winograd-domain values V and U have constants up to 100 in front of cancelling sums, so each is carried
// as TWO bf16 pieces (hi + lo: 16 mantissa bits) and a product is Vlo Uhi + Vhi Ulo + Vhi Uhi: three v_mfma_f32_32x32x16_bf16
// (8 passes each) per 16 channels where the fp32 kernel runs eight v_mfma_f32_32x32x2_f32 (16 passes each) -- 5.3x fewer
// matrix cycles at ~2^-16 relative accuracy in the transform domain.  V in LDS: [xi][piece][tile][16 ci] bf16 (the same 2 KB per
// position), U: [xi][K step][piece][Cout][16 ci] bf16 (ds_wino4_transform_weights_bf16x2; the same bytes as the fp32 form).
// Everything else -- tiles, roles, software pipeline, epilogue -- is the fp32 kernel's.
// X16 (AR = 1 only): x itself is stored as bf16 (the 16-bit configurations' dz, ds_bn_bwd_apply_bf16): a channel pair is one
// 4-byte load and already the rounded operand -- the same bits as rounding the fp32 tensor on load, half the pixel bytes
template <int NB, bool BNS, bool EDGE, int AR = 0, bool Y16 = false, bool X16 = false>      // EDGE: H or W is not a multiple of four (partial last tile row / column); Y16: the BatchNorm-sums activation is stored as bf16
__global__ __launch_bounds__(256, 1) void conv_wino4_kernel(const Wino4Params p) {
    static_assert(!X16 || AR == 1, "16-bit x storage goes with the bf16 matrix cores");
    constexpr unsigned EB = X16 ? 2u : 4u;          // bytes per element of x
    // K loop: V[2][36][32 tiles][16 ci] = 144 KB; epilogue: M[36][32 co][32 tiles] = 144 KB
    __shared__ __attribute__((aligned(128))) float smem[36 * 32 * 32];
    __shared__ __attribute__((aligned(16))) float red[2 * 32 * 32];      // statistics exchange: [sum | sum of squares][tile][channel]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
#ifdef DS_W4_PROF
    unsigned long long prof_t[12];
    const unsigned long long prof_rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, the same on every XCD
    W4_STAMP(0);
#endif
    // 1-D XCD-aware launch as conv_wino.hip: the channel blocks of a tile group run back to back on one XCD
    const int id = blockIdx.x;
    const int lin = (id & 7) * (int)(gridDim.x >> 3) + (id >> 3);
    const int group = lin / p.ncol;
    int cblk = lin - group * p.ncol;
    if (group >= p.groups) return;                          // (uniform) surplus workgroup
    // split K (small batches: fewer workgroups than CUs, so a launch lasts as long as ONE workgroup's Cin / 16 K steps): slice
    // `split` of the reduction channels, its partial output into slab `split`; wino4_splitk_reduce_kernel adds the slabs
    int split = 0;
    if (p.ksplit > 1) {
        split = cblk % p.ksplit;
        cblk /= p.ksplit;
    }
    const int kst_tot = p.Cin >> 4;                         // K steps of the whole reduction (strides of U)
    const int k0 = split * p.kchunk;
    const int co0 = cblk * 32 * NB;
    const int m0 = group * 32;
    const int tpi = p.TH * p.TW;

    // ---- transform role: tile lt, channels 2 cp, 2 cp + 1 of the 16-channel K step --------------------------------
    // The SRD starts (W + 1) pixels in front of x, so the patch origin (4 th - 1, 4 tw - 1) has a non-negative offset
    // for every tile and the pixel (py, px) of the patch is a wave-uniform, non-negative scalar offset from it.
    const int lt = tid >> 3, cp = tid & 7;
    const int64_t shift = (int64_t)(p.W + 1) * p.ldx;
    const unsigned xk0 = (unsigned)k0 * 16u * EB;          // this slice's first channel (bytes); the range still ends at the tensor's end
    const __amdgpu_buffer_rsrc_t srd_x = w4srd(reinterpret_cast<const char *>(p.x) - shift * EB + xk0, p.x_bytes + (unsigned)(shift * EB) - xk0);
    unsigned rowoff[6];         // patch row py: the tile's base offset, or out of range (row outside the image / no tile)
    bool cv[6];                 // patch column px inside the image
    {
        const int m = m0 + lt;
        const bool tv = m < p.Mt;
        const int n = (tv ? m : 0) / tpi;
        const int r = (tv ? m : 0) - n * tpi;
        const int th = r / p.TW, tw = r - th * p.TW;
        const unsigned vbase = (unsigned)(((n * p.H + 4 * th) * p.W + 4 * tw) * p.ldx + 2 * cp) * EB;
        const unsigned rowstep = (unsigned)(p.W * p.ldx) * EB;
        // the patch ROW's offset rides in the lane's vector offset (six registers that exist anyway), so the scalar offset of
        // a request is only (patch column, channel step): six values per K step instead of thirty-six multiply-add pairs --
        // with one wave per SIMD every scalar instruction is an issue slot too (~150 s_mul / s_add per K step before)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            rowoff[k] = (tv && (unsigned)(4 * th - 1 + k) < (unsigned)p.H) ? vbase + (unsigned)k * rowstep : kOOB;
            cv[k] = (unsigned)(4 * tw - 1 + k) < (unsigned)p.W;
        }
    }
    const int pixstep = p.ldx * (int)EB;

    // ---- matrix role: B fragment offsets (column li of channel block nb, channels 4 kh .. of an 8-channel half step) ----
    const int ksteps = p.ksplit > 1 ? (kst_tot - k0 < p.kchunk ? kst_tot - k0 : p.kchunk) : kst_tot;      // of THIS workgroup
    const int nhalf = 2 * kst_tot;
    unsigned boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = co0 + 32 * nb + li;
        boff[nb] = col < p.Cout ? (AR ? (unsigned)(col * 32 + 16 * kh) : (unsigned)(col * 8 + 4 * kh) * 4u) : kOOB;
    }
    const int ustep = p.Cout * 32;                                 // bytes between half steps (AR: pieces) of one position
    const int upos = nhalf * ustep;                                // bytes between positions
    const unsigned uk0 = (unsigned)k0 * 2u * (unsigned)ustep;      // this slice's first half step
    const __amdgpu_buffer_rsrc_t srd_u = w4srd(reinterpret_cast<const char *>(p.u) + uk0, p.u_bytes - uk0);

    // NB = 2: 18 accumulators but 256 accumulation registers -- left to the compiler, two accumulators share 16 of them
    // and are swapped through the vector registers around their MFMAs (192 moves and two pipeline drains per K step:
    // the MFMA-only loop measured 5.5-6.4 us per step against 3.84 of matrix cycles).  Position 8's two accumulators are
    // therefore pinned to the ARCHITECTURAL vector registers (mfma_v: the gfx90a+ MFMA takes C / D in either file).
    f32x16 acc[9][NB], accv[NB];      // not initialised: the first K step's first MFMA per accumulator takes C = 0 (k_step FIRST)
    if constexpr (AR != 0 && !DS_W4H_FIRST) {
#pragma unroll
        for (int pi = 0; pi < 9; ++pi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[pi][nb][e] = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) accv[nb][e] = 0.f;
    }
    f32x2 raw[36];
    // weight-fragment ring: group g (half step g / 9, position g % 9) uses slot g % RING and is requested RING groups ahead.
    // NB = 2 has registers for six slots (3072 matrix cycles of lead); NB = 1 runs its 18 groups in half the time, so six
    // slots would be 1536 cycles -- less than a loaded L2 round trip -- and it has the registers for a whole K step.
    constexpr int RING = NB == 1 ? DS_W4_RING1 : 6;
    static_assert(18 % RING == 0, "the slot of a group must not depend on the K step");
    f32x4 b[RING][NB];
    auto load_pixel = [&](int q, int c0, const unsigned *ro) {
        const int py = q / 6, px = q - py * 6;
        if ((DS_W4_ABL & 2) && c0 >= 32) return;
        // (patch columns 1 .. 4 are image columns 4 tw .. 4 tw + 3: inside the image unless the map has a partial last tile)
        const unsigned vo = (cv[px] || (!EDGE && px >= 1 && px <= 4)) ? ro[py] : kOOB;
        if constexpr (X16) raw[q] = unpk_bf16(__builtin_amdgcn_raw_buffer_load_b32(srd_x, vo, px * pixstep + c0 * 2, 0));
        else raw[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(srd_x, vo, px * pixstep + c0 * 4, 0));
    };
    // AR: a slot holds one POSITION of a K step (hi and lo piece per channel block), requested RP positions ahead
    constexpr int RP = NB == 1 ? 9 : DS_W4H_RP2;
    f32x4 bh[AR ? RP : 1][NB], bl[AR ? RP : 1][NB];
    auto load_b_h = [&](int slot, int pi, int ks) {
        const int so = ((wave * 9 + pi) * kst_tot + ks) * 2 * ustep;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bh[slot][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_u, boff[nb], so, 0));
            bl[slot][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_u, boff[nb], so + ustep, 0));
        }
    };
    auto load_b = [&](int slot, int pi, int hs) {
        if ((DS_W4_ABL & 1) && hs > 0) return;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            b[slot][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_u, boff[nb], (wave * 9 + pi) * upos + hs * ustep, 0));
    };

    // ---- the input transform V = B^T d B of this thread's (tile, channel pair), in two halves of six chunks each so it
    // can be spread over the MFMA groups of the previous K step:
    //   column chunk px: B^T applied down patch column px (raw[.][px] -> t[.][px]; the pixels of that column are dead)
    //   row chunk i:     B^T applied along row i of t -> the six positions (i, 0..5), written position-major to LDS
    f32x2 t[36];
    auto col_chunk = [&](int px) {
        if (DS_W4_ABL & 4) return;
        if constexpr (AR == 1 && !X16) {          // the convolution's operand is bf16(x) (X16: stored that way)
#pragma unroll
            for (int k = 0; k < 6; ++k) raw[6 * k + px] = unpk_bf16(pk_bf16(raw[6 * k + px]));
        }
        in1d(raw[px], raw[6 + px], raw[12 + px], raw[18 + px], raw[24 + px], raw[30 + px], t[px], t[6 + px], t[12 + px],
             t[18 + px], t[24 + px], t[30 + px]);
    };
    auto row_chunk = [&](int i, float *Vw) {
        if (DS_W4_ABL & 4) {
            if (!(DS_W4_ABL & 8))
#pragma unroll
                for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2 *>(Vw + (i * 6 + j) * 512 + tid * 2) = t[i * 6 + j];
            return;
        }
        f32x2 v[6];
        in1d(t[i * 6], t[i * 6 + 1], t[i * 6 + 2], t[i * 6 + 3], t[i * 6 + 4], t[i * 6 + 5], v[0], v[1], v[2], v[3], v[4], v[5]);
        if constexpr (AR) {               // two bf16 pieces per value: [position][piece][tile][16 ci]
            unsigned *Vu = reinterpret_cast<unsigned *>(Vw);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const unsigned h = pk_bf16(v[j]);
                Vu[(i * 6 + j) * 512 + tid] = h;
                Vu[(i * 6 + j) * 512 + 256 + tid] = pk_bf16(psub(v[j], unpk_bf16(h)));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (!(DS_W4_ABL & 8)) *reinterpret_cast<f32x2 *>(Vw + (i * 6 + j) * 512 + tid * 2) = v[j];
            else asm volatile("" :: "v"(v[j]));
    };

    // prologue: pixels and transform of K step 0, pixels of K step 1, the first six weight fragments
#pragma unroll
    for (int q = 0; q < 36; ++q) load_pixel(q, 0, rowoff);
    if constexpr (AR) {
#pragma unroll
        for (int g = 0; g < RP; ++g) load_b_h(g, g, 0);
    } else {
#pragma unroll
        for (int g = 0; g < RING; ++g) load_b(g, g % 9, g / 9);
    }
#pragma unroll
    for (int px = 0; px < 6; ++px) col_chunk(px);
    if (ksteps > 1) {
#pragma unroll
        for (int q = 0; q < 36; ++q) load_pixel(q, 16, rowoff);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) row_chunk(i, smem);
    W4_STAMP(1);
    lds_barrier();
    W4_STAMP(2);

    // One K step.  LAST = false: every step but the last -- the transform of step ks + 1 and the pixel requests of step
    // ks + 2 are UNCONDITIONAL (past the last step the requests carry out-of-range offsets and return zeros nobody
    // uses): a condition around them would keep all 72 pixel registers alive across the whole loop next to the 72
    // registers of the half-transformed patch, which the kernel does not have.  LAST = true: MFMAs only.
    auto k_step = [&](int ks, auto last_tag, auto first_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;      // the accumulators are not initialised: C = 0 in their first MFMA
        const float *Vr = smem + (ks & 1) * (36 * 512);
        float *Vw = smem + ((ks + 1) & 1) * (36 * 512);
        const int c2 = (ks + 2) * 16;
        // row offsets of the requests for step ks + 2: out of range past the last step (the scalar offset is not part of
        // the SRD's range check, so a request there could leave the tensor).  The asm keeps the 36 per-pixel offsets
        // from being hoisted out of the loop into 36 registers the kernel does not have: one v_cndmask each.
        unsigned ro[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ro[k] = (ks + 2 < ksteps) ? rowoff[k] : kOOB;
            asm volatile("" : "+v"(ro[k]));
        }
        // ---- two 8-channel half steps x nine positions = 18 groups: A fragment from LDS, 4 NB MFMAs.  Between the
        // groups, in the shadow of the matrix pipe: the transform of K step ks + 1 (column chunks in groups 0-5, row
        // chunks in groups 6-11: VALU issue still adds to the MFMA time, but load, LDS and dependency latencies no longer
        // do), from group 6 on three pixels per group for K step ks + 2, column by column, into the registers the column
        // chunks freed (used twelve groups later), and the weights six groups ahead -----------------------------------
        const float *Va = Vr + (wave * 9) * 512 + li * 16 + kh * 4;
        // NB = 1: two groups at a time, their MFMAs alternating -- four back-to-back MFMAs on ONE accumulator wait for
        // each other's results
        constexpr int GP = NB == 1 ? 2 : 1;
        f32x4 av[GP], avn[GP];
#pragma unroll
        for (int u = 0; u < GP; ++u) av[u] = avn[u] = *reinterpret_cast<const f32x4 *>(Va + u * 512);
#pragma unroll
        for (int g0 = 0; g0 < 18; g0 += GP) {
#pragma unroll
            for (int u = 0; u < GP; ++u)
                if (g0 + GP + u < 18) avn[u] = *reinterpret_cast<const f32x4 *>(Va + ((g0 + GP + u) % 9) * 512 + ((g0 + GP + u) / 9) * 8);
            if constexpr (!LAST) {
#pragma unroll
                for (int g = g0; g < g0 + GP; ++g) {
                    if (g < 6) col_chunk(g);
                    else if (g < 12) row_chunk(g - 6, Vw);
                    // the pixels of K step ks + 2, column by column into the registers the column chunks freed
                    if (DS_W4_PIX == 0 && g >= 6) {          // three per group over groups 6-17 (lead: 6-11 groups)
#pragma unroll
                        for (int k = 3 * (g - 6); k < 3 * (g - 6) + 3; ++k) load_pixel((k % 6) * 6 + k / 6, c2, ro);
                    }
                    if (DS_W4_PIX == 1 && g < 6) {           // column g right behind its column chunk (lead: 18 groups)
#pragma unroll
                        for (int k = 6 * g; k < 6 * g + 6; ++k) load_pixel((k % 6) * 6 + k / 6, c2, ro);
                    }
                    if (DS_W4_PIX == 2 && g < 9) {           // four per group over groups 0-8 (column k / 6 <= g is free)
#pragma unroll
                        for (int k = 4 * g; k < 4 * g + 4; ++k) load_pixel((k % 6) * 6 + k / 6, c2, ro);
                    }
                    if (DS_W4_PIX == 3 && g < 12) {          // three per group over groups 0-11
#pragma unroll
                        for (int k = 3 * g; k < 3 * g + 3; ++k) load_pixel((k % 6) * 6 + k / 6, c2, ro);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < GP; ++u)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int g = g0 + u, pi = g % 9;
                        if (FIRST && g < 9 && j == 0) {
                            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            if (NB == 2 && pi == 8) mfma_v0(accv[nb], av[u][j], b[g % RING][nb][j]);
                            else acc[pi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], b[g % RING][nb][j], zero16, 0, 0, 0);
                        } else if (NB == 2 && pi == 8) mfma_v(accv[nb], av[u][j], b[g % RING][nb][j]);
                        else acc[pi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], b[g % RING][nb][j], acc[pi][nb], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GP; ++u) {
                const int g = g0 + u;
                if (g + RING < 18) load_b(g % RING, (g + RING) % 9, 2 * ks + (g + RING) / 9);
                else if (!LAST) load_b(g % RING, (g + RING - 18) % 9, 2 * ks + 2 + (g + RING - 18) / 9);
            }
#pragma unroll
            for (int u = 0; u < GP; ++u) av[u] = avn[u];
        }
    };
    // AR: one K step = nine groups (a position each: both pieces of V, 3 NB MFMAs); the work between the groups is the fp32
    // step's, two of its eighteen slices per group.
    auto k_step_h = [&](int ks, auto last_tag, auto first_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value && DS_W4H_FIRST;
        const float *Vr = smem + (ks & 1) * (36 * 512);
        float *Vw = smem + ((ks + 1) & 1) * (36 * 512);
        const int c2 = (ks + 2) * 16;
        unsigned ro[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ro[k] = (ks + 2 < ksteps) ? rowoff[k] : kOOB;
            asm volatile("" : "+v"(ro[k]));
        }
        const float *Va = Vr + (wave * 9) * 512 + li * 8 + kh * 4;          // tile li, channels 8 kh .. 8 kh + 7 (16 bytes)
        f32x4 avh = *reinterpret_cast<const f32x4 *>(Va), avl = *reinterpret_cast<const f32x4 *>(Va + 256), nvh = avh, nvl = avl;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if (g + 1 < 9) {
                nvh = *reinterpret_cast<const f32x4 *>(Va + (g + 1) * 512);
                nvl = *reinterpret_cast<const f32x4 *>(Va + (g + 1) * 512 + 256);
            }
            if constexpr (!LAST) {
#pragma unroll
                for (int G = 2 * g; G < 2 * g + 2; ++G) {
                    if (G < 6) col_chunk(G);
                    else if (G < 12) row_chunk(G - 6, Vw);
                    if (G >= 6) {
#pragma unroll
                        for (int k = 3 * (G - 6); k < 3 * (G - 6) + 3; ++k) load_pixel((k % 6) * 6 + k / 6, c2, ro);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)          // small terms first: Vlo Uhi, Vhi Ulo, Vhi Uhi
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f32x4 va = s3 == 0 ? avl : avh, vb = s3 == 1 ? bl[g % RP][nb] : bh[g % RP][nb];
                    if (NB == 2 && g == 8) {
                        if (FIRST && s3 == 0) mfma_hv0(accv[nb], va, vb);
                        else mfma_hv(accv[nb], va, vb);
                    } else if (FIRST && s3 == 0) {
                        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), zero16, 0, 0, 0);
                    } else {
                        acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc[g][nb], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            if (g + RP < 9) load_b_h(g % RP, g + RP, ks);
            else if (!LAST) load_b_h(g % RP, g + RP - 9, (DS_W4H_UNILOOP && ks + 1 >= ksteps) ? ks : ks + 1);      // (the scalar offset is not range-checked)
            avh = nvh;
            avl = nvl;
        }
    };
    auto step = [&](int ks, auto last_tag, auto first_tag) {
        if constexpr (AR) k_step_h(ks, last_tag, first_tag);
        else k_step(ks, last_tag, first_tag);
    };
    if (AR != 0 && DS_W4H_UNILOOP && ksteps > 1) {
        // AR, NB = 2: every step after the first runs the loop body, the last one too (its transform and requests work on zeros:
        // the offsets past the reduction are out of range).  With a separate last step behind a loop that may run zero times the
        // register allocator parked all 256 accumulation registers in scratch on the edge around the loop (1 KB per lane,
        // 300 KB of scratch writes per workgroup): Conv2d_2c's input gradient 474 -> 406 us without it.  NB = 1 has the
        // registers and keeps the cheaper last step.
        step(0, std::false_type{}, std::true_type{});
        lds_barrier();
        int ks = 1;
        do {
            step(ks, std::false_type{}, std::false_type{});
            lds_barrier();
        } while (++ks < ksteps);
    } else if (ksteps > 1) {
        step(0, std::false_type{}, std::true_type{});
        lds_barrier();          // V of step ks + 1 is complete, V of step ks is free
        for (int ks = 1; ks + 1 < ksteps; ++ks) {
            step(ks, std::false_type{}, std::false_type{});
            lds_barrier();
        }
        step(__builtin_amdgcn_readfirstlane(ksteps - 1), std::true_type{}, std::false_type{});      // (uniform: no waterfall loops around its loads)
    } else {
        step(0, std::true_type{}, std::true_type{});
    }
    W4_STAMP(3);

    // ---- output transform Y = A^T M A through LDS, one 32-channel block at a time ---------------------------------------
    // The waves park their accumulators as M[xi][tile][co] (dword writes, a wave's 32 channel lanes side by side); then
    // thread (tile et = tid / 8, channel quad eq = tid % 8) gathers the 36 positions of its tile for FOUR channels with
    // one ds_read_b128 each (lane-linear), transforms them, and stores every output pixel as 16 bytes: 16 store
    // instructions per thread and block where a thread-per-channel layout needs 64 (the dword stores were the bulk of the
    // epilogue's time).
    const int et = tid >> 3, eq = tid & 7;
    float neg1 = -1.f;
    asm volatile("" : "+s"(neg1));          // opaque -1 (see out1d)
    const __amdgpu_buffer_rsrc_t srd_z = w4srd(p.z + (int64_t)split * p.zslab, p.z_bytes);
    const __amdgpu_buffer_rsrc_t srd_y = w4srd(BNS ? p.y : p.z, Y16 ? p.z_bytes / 2 : p.z_bytes);
    const int orow = p.W * p.ldz * 4, opix = p.ldz * 4;
    int tbase, hrem, wrem;      // pixel index of the tile's top-left output (or -1); EDGE: rows / columns inside the image
    {
        const int m = m0 + et;
        const int n = (m < p.Mt ? m : 0) / tpi;
        const int rr = (m < p.Mt ? m : 0) - n * tpi;
        const int th = rr / p.TW, tw = rr - th * p.TW;
        tbase = m < p.Mt ? (n * p.H + 4 * th) * p.W + 4 * tw : -1;
        hrem = p.H - 4 * th;
        wrem = p.W - 4 * tw;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        lds_barrier();            // the K loop's last fragment reads / the previous block's gathers are done
        // accumulator element e: tile row (e & 3) + 8 (e >> 2) + 4 kh, channel column li
#pragma unroll
        for (int pi = 0; pi < 9; ++pi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const f32x16 &c = (NB == 2 && pi == 8) ? accv[nb] : acc[pi][nb];
                smem[(wave * 9 + pi) * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + li] = c[e];
            }
        W4_STAMP(4 + 3 * nb);     // parked
        lds_barrier();
        const int col = co0 + 32 * nb + 4 * eq;                 // first of this thread's four channels (Cout % 4 == 0)
        const bool colok = col < p.Cout;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (!BNS && p.pivot && colok) pv = *reinterpret_cast<const f32x4 *>(p.pivot + col);
        const unsigned vo = (tbase >= 0 && colok) ? (unsigned)(tbase * p.ldz + col) * 4u : kOOB;
        // DS_EPI_BNSUMS: the consumer's activations at the sixteen store offsets, ALL requested here -- in front of the
        // gathers and of every store of this pass.  Requested row by row inside the store loop, each row's reads sat behind
        // the previous row's stores (one in-order memory counter): a round trip per output row.
        f32x4 yall[BNS ? 4 : 1][4];
        if constexpr (BNS) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if constexpr (Y16) {          // four bf16 = 8 bytes at half the fp32 offsets
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                            srd_y, (!EDGE || (rr < hrem && k < wrem)) ? (vo == kOOB ? kOOB : vo >> 1) : kOOB, (rr * orow + k * opix) >> 1, 0));
                        yall[rr][k] = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                                            __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
                    } else {
                        yall[rr][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            srd_y, (!EDGE || (rr < hrem && k < wrem)) ? vo : kOOB, rr * orow + k * opix, 0));
                    }
        }
        const float *Mq = smem + tid * 4;
        f32x4 P[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            out1d(neg1, *reinterpret_cast<const f32x4 *>(Mq + (0 * 6 + j) * 1024), *reinterpret_cast<const f32x4 *>(Mq + (1 * 6 + j) * 1024),
                  *reinterpret_cast<const f32x4 *>(Mq + (2 * 6 + j) * 1024), *reinterpret_cast<const f32x4 *>(Mq + (3 * 6 + j) * 1024),
                  *reinterpret_cast<const f32x4 *>(Mq + (4 * 6 + j) * 1024), *reinterpret_cast<const f32x4 *>(Mq + (5 * 6 + j) * 1024),
                  P[0][j], P[1][j], P[2][j], P[3][j]);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            unsigned vof[4];            // store offsets of this output row's four pixels
#pragma unroll
            for (int k = 0; k < 4; ++k) vof[k] = (!EDGE || (rr < hrem && k < wrem)) ? vo : kOOB;
            const f32x4 *yv = yall[rr];
            f32x4 y[4];
            out1d(neg1, P[rr][0], P[rr][1], P[rr][2], P[rr][3], P[rr][4], P[rr][5], y[0], y[1], y[2], y[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // the pixel's offset goes into the VECTOR offset, the scalar offset stays the constant 0: with an SGPR there
                // the compiler assumes that a 16-byte buffer store has no "VALU overwrites the store data" hazard (true on
                // gfx900) and schedules the writes of y's registers right behind the store -- on gfx950 the store then
                // carried the NEW value in lanes 4-7 of every 8 (measured: one channel of one pixel per tile)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[k]), srd_z, vof[k] + (unsigned)(rr * orow + k * opix), 0, 2 /* nt */);
                if constexpr (BNS) {
                    f32x4 g;
#pragma unroll
                    for (int c = 0; c < 4; ++c) g[c] = yv[k][c] > 0.f ? y[k][c] : 0.f;      // (an out-of-range offset reads y = 0)
                    s += g;
                    q = qfma(g, yv[k], q);
                } else {
                    // pixels outside the image (EDGE) and tiles / channels past the end do not count: on full maps the
                    // condition is the thread's own (vo), applied once behind the loop
                    f32x4 uu = qfma(q4(neg1), pv, y[k]);
                    if (EDGE && vof[k] == kOOB) uu = q4(0.f);
                    s += uu;
                    q = qfma(uu, uu, q);
                }
            }
        }
        if (!BNS && !EDGE && vo == kOOB) s = q = f32x4{0.f, 0.f, 0.f, 0.f};
        W4_STAMP(5 + 3 * nb);     // gathered, transformed, stores issued
        if (BNS || (p.flags & DS_EPI_STATS)) {
            // the 32 tiles of a channel: every thread leaves its four channels' partial sums in LDS, then 64 threads (channel,
            // sum | sum of squares) add the 32 tiles in index order.  (Was: three rounds of eight ds_bpermute shuffles, a
            // four-wave combine and a barrier -- 1 us per channel block, measured by scripts/wino4_phase_prof.py.)
            *reinterpret_cast<f32x4 *>(red + et * 32 + 4 * eq) = s;
            *reinterpret_cast<f32x4 *>(red + 1024 + et * 32 + 4 * eq) = q;
            lds_barrier();
            if (tid < 64) {
                const int ch = tid & 31, which = tid >> 5;
                float acc2 = 0.f;
#pragma unroll
                for (int tt = 0; tt < 32; ++tt) acc2 += red[which * 1024 + tt * 32 + ch];
                if (co0 + 32 * nb + ch < p.Cout)
                    p.stats[((int64_t)which * p.Cout + co0 + 32 * nb + ch) * p.groups + group] = acc2;
            }
        }
        W4_STAMP(6 + 3 * nb);
    }
#ifdef DS_W4_PROF
    asm volatile("s_waitcnt vmcnt(0)");
    W4_STAMP(10);                 // stores drained
    if (tid == 0 && p.prof) {
#pragma unroll
        for (int i = 0; i < 11; ++i) p.prof[(int64_t)blockIdx.x * 16 + i] = (i >= 4 + 3 * NB && i < 10) ? 0ull : prof_t[i];
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.prof[(int64_t)blockIdx.x * 16 + 11] = xcc & 15u;
        p.prof[(int64_t)blockIdx.x * 16 + 12] = prof_rt0;
        p.prof[(int64_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// U = G g G^T (6 x 6) for every (ci, co) pair of the TF HWIO filter w [3][3][Cin][Cout], stored for the kernel's
// B loads as U[xi][r / 8][o][r % 8] with (r, o) = (reduction channel, output channel):
//   dgrad == 0: g = w[:, :, ci, co], (r, o) = (ci, co)                               (forward)
//   dgrad == 1: g = w[2 - kh, 2 - kw, ci, co], (r, o) = (co, ci)                     (Conv2DBackpropInput)
//   HB (ds_wino4_transform_weights_bf16x2): g is rounded to bf16 first (the 16-bit configurations' operand), U is stored as
//   two bf16 pieces (hi + lo) in the K-loop order of the AR kernel: [xi][r / 16][piece][o][r % 16].
template <bool HB>
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float *w, float *u, int Cin, int Cout, int dgrad) {
    const int64_t total = (int64_t)Cin * Cout;
    const int R = dgrad ? Cout : Cin, O = dgrad ? Cin : Cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i / Cout), co = (int)(i - (int64_t)ci * Cout);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
            {
                g[a][b] = w[((int64_t)((dgrad ? 2 - a : a) * 3 + (dgrad ? 2 - b : b)) * Cin + ci) * Cout + co];
                if (HB) g[a][b] = (float)(__bf16)g[a][b];
            }
        // G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
        auto g1d = [](float g0, float g1, float g2, float *t) {
            const float e = (g0 + g2) * (-1.f / 6.f), o = g1 * (1.f / 6.f);
            const float f = fmaf(g0, 1.f / 24.f, g2 * (1.f / 6.f)), h = g1 * (1.f / 12.f);
            t[0] = g0 * 0.25f;
            t[1] = e - o;
            t[2] = e + o;
            t[3] = f + h;
            t[4] = f - h;
            t[5] = g2;
        };
        float t[6][3];          // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float c[6];
            g1d(g[0][b], g[1][b], g[2][b], c);
#pragma unroll
            for (int a = 0; a < 6; ++a) t[a][b] = c[a];
        }
        const int r = dgrad ? co : ci, o = dgrad ? ci : co;
        const int64_t base = ((int64_t)(r >> 3) * O + o) * 8 + (r & 7);
        const int64_t plane = (int64_t)(R >> 3) * O * 8;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float c[6];
            g1d(t[a][0], t[a][1], t[a][2], c);
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) {
                if (HB) {
                    __bf16 *uh = reinterpret_cast<__bf16 *>(u);
                    const int64_t at = ((((int64_t)(a * 6 + bb) * (R >> 4) + (r >> 4)) * 2) * O + o) * 16 + (r & 15);
                    const __bf16 hi = (__bf16)c[bb];
                    uh[at] = hi;
                    uh[at + (int64_t)O * 16] = (__bf16)(c[bb] - (float)hi);
                } else {
                    u[(a * 6 + bb) * plane + base] = c[bb];
                }
            }
        }
    }
}

// Which kernel for a shape?  Launch-time model fitted to profiles/r03_wino4_layers.txt (B = 256 and B = 32): a launch
// takes ceil(workgroups / CUs) rounds of one workgroup's duration,
//     F(4x4), NB = 2: 10 us + 5.3 us per 16-channel K step      NB = 1: 6 us + 3.1 us per K step
//     F(2x2) (conv_wino.hip): 8.7 us + 2.35 us per 8-channel K step, 128 tiles of 2x2 x 32 channels per workgroup
// so the 14 x 14 and 7 x 7 maps (128 / 32 tile groups only) go to whichever fills the rounds best.
// (round 5, after the C = 0 first step and the issue-slot diet: 10 + 5.3 / step at NB = 2, 6 + 3.1 / step at NB = 1; with the
// round-3 constants 13.5 + 5.4 and 8.5 + 3.2 the step was 0.08 ms slower, three interleaved pairs)
#ifndef W4_FIX2
#define W4_FIX2 10.0
#define W4_STEP2 5.3
#define W4_FIX1 6.0
#define W4_STEP1 3.1
#endif
// (bf16x2 arithmetic: the K step is the transform's, not the matrix pipe's)
#ifndef W4H_STEP2
#define W4H_STEP2 2.6
#define W4H_STEP1 1.8
#endif
struct W4Choice {
    int nb;             // 1, 2: F(4x4) with that channel-block count
    double us4, us2;    // expected launch time of F(4x4) with nb / of F(2x2)
};
int g_w4_forced_nb = -1;      // ds_debug_conv_wino4_set_nb (tests, tuning); -1: not set yet -> DS_WINO4_NB or automatic
W4Choice w4_choose(int N, int H, int W, int Cin, int Cout, int ar = 0) {
    if (g_w4_forced_nb < 0) {
        const char *e = ds::tune_env("DS_WINO4_NB");
        g_w4_forced_nb = e ? atoi(e) : 0;
    }
    const int forced = g_w4_forced_nb;
    const double cus = 256.0;
    const int64_t mt4 = (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4), g4 = (mt4 + 31) / 32;
    const int64_t mt2 = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2), g2 = (mt2 + 127) / 128;
    W4Choice c;
    double t[3];
    for (int nb = 1; nb <= 2; ++nb) {
        const int64_t wgs = g4 * ((Cout + 32 * nb - 1) / (32 * nb));
        const double stp = ar ? (nb == 2 ? W4H_STEP2 : W4H_STEP1) : (nb == 2 ? W4_STEP2 : W4_STEP1);
        t[nb] = ceil(wgs / cus) * ((Cin / 16) * stp + (nb == 2 ? W4_FIX2 : W4_FIX1));
    }
    c.us2 = ceil(g2 * ((Cout + 31) / 32) / cus) * (8.7 + 2.35 * (Cin / 8));
    c.nb = (forced == 1 || forced == 2) ? forced : (t[2] <= t[1] ? 2 : 1);
    c.us4 = t[c.nb];
    return c;
}

// ---- split K: the second launch ------------------------------------------------------------------------------------------
// z[row][c] = sum over the slices' slabs, in slice order (deterministic), + the epilogue the conv launch could not run on
// partial sums: DS_EPI_STATS (column sums of z - pivot and its square) or DS_EPI_BNSUMS (sum g, sum g*y with g = z (y > 0)).
// Thread = (channel quad, row group) as bn_bwd_reduce_kernel; workgroup b takes rows [b * rpb, (b + 1) * rpb); partials
// [2][Cout][gridDim.x].  MODE 0: plain, 1: STATS, 2: BNSUMS (Y16: y in bf16 storage).
template <int MODE, bool Y16>
__global__ __launch_bounds__(256) void wino4_splitk_reduce_kernel(const float *slab, int S, int64_t zslab, float *z, int ldz,
                                                                  int64_t M, int C, const float *pivot, const void *yv,
                                                                  float *partials, int rpb) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [RG][C4][8]
    const int C4 = C >> 2;
    const int RG = 256 / C4 > 0 ? 256 / C4 : 1;
    const int tid = threadIdx.x;
    const int cg = tid % C4, rg = tid / C4;
    const bool active = tid < RG * C4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    int64_t r1 = r0 + rpb;
    if (r1 > M) r1 = M;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int c = cg * 4;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1 && pivot) pv = *reinterpret_cast<const f32x4 *>(pivot + c);
        for (int64_t row = r0 + rg; row < r1; row += RG) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(slab + row * C + c);
            for (int k = 1; k < S; ++k) v += *reinterpret_cast<const f32x4 *>(slab + k * zslab + row * C + c);
            *reinterpret_cast<f32x4 *>(z + row * ldz + c) = v;
            if constexpr (MODE == 1) {
                const f32x4 u = v - pv;
                s += u;
                q += u * u;
            } else if constexpr (MODE == 2) {
                f32x4 y;
                if constexpr (Y16) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    y = __builtin_convertvector(*reinterpret_cast<const bf16x4 *>(reinterpret_cast<const __bf16 *>(yv) + row * ldz + c), f32x4);
                } else {
                    y = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(yv) + row * ldz + c);
                }
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = y[j] > 0.f ? v[j] : 0.f;
                s += g;
                q += g * y;
            }
        }
        if constexpr (MODE != 0) {
            float *o = sh + ((int64_t)rg * C4 + cg) * 8;
            *reinterpret_cast<f32x4 *>(o) = s;
            *reinterpret_cast<f32x4 *>(o + 4) = q;
        }
    }
    if constexpr (MODE != 0) {
        __syncthreads();
        if (tid < C4) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += sh[((int64_t)g * C4 + tid) * 8 + j];
            const int P = gridDim.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                partials[(int64_t)(tid * 4 + j) * P + blockIdx.x] = a[j];
                partials[((int64_t)C + tid * 4 + j) * P + blockIdx.x] = a[4 + j];
            }
        }
    }
}

// rows per workgroup of the reduce launch: about 512 workgroups, at least 8 rows (a thread then walks 2-4 rows of S slabs: with
// 64 rows per workgroup the 7x7 maps of 32 samples gave 25 workgroups and the reduce launch took longer than the conv it followed)
inline int splitk_rpb(int64_t M) {
    int64_t r = (M + 511) / 512;
    r = (r + 7) / 8 * 8;
    return (int)(r < 8 ? 8 : r);
}

// Split K?  Only where the unsplit launch is ONE partial round of workgroups (fewer than half the CUs), i.e. lasts as long as
// one workgroup's Cin / 16 K steps whatever the batch: S slices cut that to ceil(ksteps / S) steps + a second launch
// (~9 us with its boundary).  us per K step / fixed part as w4_choose (NB = 1: such launches never choose NB = 2).
int w4_splitk_choose(int N, int H, int W, int Cin, int Cout) {
    static int forced = -1;
    if (forced < 0) {
        const char *e = ds::tune_env("DS_WINO4_SPLITK");
        forced = e ? atoi(e) : 0;
    }
    const int ksteps = Cin / 16;
    if (Cout > 1024) return 1;
    if (forced == 1) return 1;
    if (forced > 1) return forced < ksteps ? forced : (ksteps > 1 ? ksteps : 1);
    // (up to 32 samples per launch only: in the step the other branch chains run beside this launch, and from 64 samples on the
    // slices' extra prologues / epilogues and the reduce launch take more from them than the shorter latency gives back --
    // B = 64 5.16 -> 5.20 ms, B = 128 7.92 -> 7.95, B = 32 3.88 -> 3.76, profiles/r06_notes.md)
    if (N > 32) return 1;
    const W4Choice c = w4_choose(N, H, W, Cin, Cout);
    if (c.nb != 1) return 1;
    const int64_t mt4 = (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4), g4 = (mt4 + 31) / 32;
    const int64_t wgs = g4 * ((Cout + 31) / 32);
    if (wgs * 2 > 256 || ksteps < 6) return 1;
    int S = (int)(256 / wgs);
    if (S > 4) S = 4;
    if (S > ksteps / 3) S = ksteps / 3;
    if (S < 2) return 1;
    const int chunk = (ksteps + S - 1) / S;
    S = (ksteps + chunk - 1) / chunk;                  // no empty slice
    const double t1 = ksteps * W4_STEP1 + W4_FIX1, ts = chunk * W4_STEP1 + W4_FIX1 + 9.0;
    return (S >= 2 && ts < 0.85 * t1) ? S : 1;
}

}  // namespace

#ifdef DS_W4_PROF
static unsigned long long *g_w4_prof = nullptr;
extern "C" void ds_debug_w4_set_prof(unsigned long long *buf) { g_w4_prof = buf; }
#endif

// Debug aid (process-global, never called by the product path): pin the channel blocks per workgroup of ds_conv_wino4
// (1, 2; 0 = the launch-time model) so that tests reach both instantiations at small sizes.
#ifdef DS_TUNING
extern "C" int ds_debug_conv_wino4_set_nb(int nb) {
    DS_REQUIRE(nb >= 0 && nb <= 2, "ds_debug_conv_wino4_set_nb: 0 = automatic, 1, 2");
    g_w4_forced_nb = nb;
    return DS_OK;
}
#endif

extern "C" int ds_conv_wino4_supported(int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    return (H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0 && Cout % 4 == 0) ? 1 : 0;
}

extern "C" int ds_conv_wino4_prefer(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (!ds_conv_wino4_supported(H, W, Cin, Cout) || N <= 0) return 0;
    const W4Choice c = w4_choose(N, H, W, Cin, Cout);
    return c.us4 < c.us2 ? c.nb : 0;
}

extern "C" int ds_wino4_transform_weights(const float *w, float *u, int32_t Cin, int32_t Cout, int32_t dgrad, void *stream) {
    DS_REQUIRE(w && u && Cin > 0 && Cout > 0, "ds_wino4_transform_weights: bad argument");
    DS_REQUIRE((dgrad ? Cout : Cin) % 8 == 0, "ds_wino4_transform_weights: the reduction channels must be a multiple of 8");
    hipLaunchKernelGGL(wino4_weights_kernel<false>, dim3(ds::stream_grid((int64_t)Cin * Cout, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, u, Cin, Cout, dgrad);
    return ds::check_launch("ds_wino4_transform_weights");
}

extern "C" int ds_wino4_transform_weights_bf16x2(const float *w, void *u2, int32_t Cin, int32_t Cout, int32_t dgrad, void *stream) {
    DS_REQUIRE(w && u2 && Cin > 0 && Cout > 0, "ds_wino4_transform_weights_bf16x2: bad argument");
    DS_REQUIRE((dgrad ? Cout : Cin) % 16 == 0, "ds_wino4_transform_weights_bf16x2: the reduction channels must be a multiple of 16");
    hipLaunchKernelGGL(wino4_weights_kernel<true>, dim3(ds::stream_grid((int64_t)Cin * Cout, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, (float *)u2, Cin, Cout, dgrad);
    return ds::check_launch("ds_wino4_transform_weights_bf16x2");
}

extern "C" int ds_conv_wino4_partials(int32_t N, int32_t H, int32_t W) {
    const int64_t mt = (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4);
    return (int)((mt + 31) / 32);
}

namespace {
int w4_launch(int ar, bool y16, const float *x, const float *u, float *z, float *stats, const float *pivot, const float *ymask,
              int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz, int32_t flags, void *stream,
              bool x16 = false, int ksplit = 1, float *slabs = nullptr) {
    DS_REQUIRE(x && u && z && N > 0, "ds_conv_wino4: bad argument");
    DS_REQUIRE(ksplit == 1 || (ar == 0 && slabs && ksplit >= 2 && ksplit <= Cin / 16), "ds_conv_wino4_splitk: 2 .. Cin / 16 slices, fp32, with a workspace");
    DS_REQUIRE(ds_conv_wino4_supported(H, W, Cin, Cout) && ldx >= Cin && ldx % 2 == 0 && ldz >= Cout && ldz % 4 == 0 && ((((uintptr_t)u) | ((uintptr_t)z)) & 15) == 0 &&
                   (((uintptr_t)x) & (x16 ? 3 : 7)) == 0 && (!x16 || ar == 1) && (!(flags & DS_EPI_BNSUMS) || (((uintptr_t)ymask) & 15) == 0) &&
                   (!(flags & DS_EPI_STATS) || !pivot || (((uintptr_t)pivot) & 15) == 0),
               "ds_conv_wino4: needs Cin %% 16 == 0, Cout %% 4 == 0, even ldx, ldz %% 4 == 0, 8-byte aligned x, 16-byte aligned u / z / y / pivot");
    DS_REQUIRE((flags & ~(DS_EPI_STATS | DS_EPI_BNSUMS)) == 0 && (!(flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) || stats),
               "ds_conv_wino4: only DS_EPI_STATS / DS_EPI_BNSUMS are supported (with a partials buffer)");
    DS_REQUIRE(!(flags & DS_EPI_BNSUMS) || (ymask && !(flags & DS_EPI_STATS)),
               "ds_conv_wino4: DS_EPI_BNSUMS needs y (pixel stride ldz) and excludes DS_EPI_STATS");
    Wino4Params p;
    p.x = x; p.u = u; p.z = z; p.stats = stats; p.pivot = (flags & DS_EPI_STATS) ? pivot : nullptr;
    p.y = (flags & DS_EPI_BNSUMS) ? ymask : nullptr;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.ldz = ldz;
    p.TH = (H + 3) / 4; p.TW = (W + 3) / 4;
    const int64_t mt = (int64_t)N * p.TH * p.TW;
    const int64_t xb = ((int64_t)N * H * W - 1) * ldx + Cin + (int64_t)(W + 1) * ldx, ub = (int64_t)36 * Cin * Cout;
    DS_REQUIRE(mt < (1ll << 30) && xb * 4 < (1ll << 31) && ub * 4 < (1ll << 31), "ds_conv_wino4: operand larger than 2 GiB");
    p.Mt = (int)mt;
    p.x_bytes = (unsigned)((((int64_t)N * H * W - 1) * ldx + Cin) * (x16 ? 2 : 4));
    p.u_bytes = (unsigned)(ub * 4);
    const int64_t zb = ((int64_t)N * H * W - 1) * ldz + Cout;
    DS_REQUIRE(zb * 4 < (1ll << 31), "ds_conv_wino4: output larger than 2 GiB");
    p.z_bytes = (unsigned)(zb * 4);
    p.flags = flags;
#ifdef DS_W4_PROF
    p.prof = g_w4_prof;
#endif
    p.groups = (int)((mt + 31) / 32);
    const int nb = ksplit > 1 ? 1 : w4_choose(N, H, W, Cin, Cout, ar).nb;
    p.ncol = (Cout + 32 * nb - 1) / (32 * nb);
    p.ksplit = 1; p.kchunk = Cin / 16; p.zslab = 0;
    const int64_t Mpix = (int64_t)N * H * W;
    if (ksplit > 1) {
        // the conv launch writes the slices' partial outputs, dense [ksplit][N H W][Cout], and runs no epilogue
        p.kchunk = (Cin / 16 + ksplit - 1) / ksplit;
        p.ksplit = (Cin / 16 + p.kchunk - 1) / p.kchunk;
        p.ncol *= p.ksplit;
        p.zslab = Mpix * Cout;
        DS_REQUIRE(p.zslab * 4 < (1ll << 31), "ds_conv_wino4_splitk: a slab larger than 2 GiB");
        p.z = slabs; p.ldz = Cout; p.z_bytes = (unsigned)(p.zslab * 4);
        p.flags = 0; p.stats = nullptr; p.pivot = nullptr; p.y = nullptr;
    }
    const dim3 grid((unsigned)(((int64_t)p.groups * p.ncol + 7) / 8 * 8));
    const bool bns = ksplit == 1 && (flags & DS_EPI_BNSUMS) != 0;
    hipStream_t st = (hipStream_t)stream;
    const bool edge = (H % 4) != 0 || (W % 4) != 0;
#define DS_W4_LAUNCH(NBV, BNSV, EDGEV)                                                                          \
    do {                                                                                                        \
        if (ar && x16 && BNSV && y16) hipLaunchKernelGGL((conv_wino4_kernel<NBV, BNSV, EDGEV, 1, BNSV, true>), grid, dim3(256), 0, st, p); \
        else if (ar && x16) hipLaunchKernelGGL((conv_wino4_kernel<NBV, BNSV, EDGEV, 1, false, true>), grid, dim3(256), 0, st, p); \
        else if (ar && BNSV && y16) hipLaunchKernelGGL((conv_wino4_kernel<NBV, BNSV, EDGEV, 1, BNSV>), grid, dim3(256), 0, st, p); \
        else if (ar) hipLaunchKernelGGL((conv_wino4_kernel<NBV, BNSV, EDGEV, 1>), grid, dim3(256), 0, st, p);   \
        else hipLaunchKernelGGL((conv_wino4_kernel<NBV, BNSV, EDGEV, 0>), grid, dim3(256), 0, st, p);           \
    } while (0)
    if (nb == 2) {
        if (bns) { if (edge) DS_W4_LAUNCH(2, true, true); else DS_W4_LAUNCH(2, true, false); }
        else { if (edge) DS_W4_LAUNCH(2, false, true); else DS_W4_LAUNCH(2, false, false); }
    } else {
        if (bns) { if (edge) DS_W4_LAUNCH(1, true, true); else DS_W4_LAUNCH(1, true, false); }
        else { if (edge) DS_W4_LAUNCH(1, false, true); else DS_W4_LAUNCH(1, false, false); }
    }
#undef DS_W4_LAUNCH
    if (ksplit > 1) {
        const int rpb = splitk_rpb(Mpix), P = (int)((Mpix + rpb - 1) / rpb);
        const int C4 = Cout / 4, RG = 256 / C4 > 0 ? 256 / C4 : 1;
        const size_t shm = (size_t)RG * C4 * 8 * sizeof(float);
        if (flags & DS_EPI_BNSUMS) {
            if (y16) hipLaunchKernelGGL((wino4_splitk_reduce_kernel<2, true>), dim3(P), dim3(256), shm, st, slabs, p.ksplit, (int64_t)p.zslab, z, ldz, Mpix, Cout, nullptr, (const void *)ymask, stats, rpb);
            else hipLaunchKernelGGL((wino4_splitk_reduce_kernel<2, false>), dim3(P), dim3(256), shm, st, slabs, p.ksplit, (int64_t)p.zslab, z, ldz, Mpix, Cout, nullptr, (const void *)ymask, stats, rpb);
        } else if (flags & DS_EPI_STATS) {
            hipLaunchKernelGGL((wino4_splitk_reduce_kernel<1, false>), dim3(P), dim3(256), shm, st, slabs, p.ksplit, (int64_t)p.zslab, z, ldz, Mpix, Cout, pivot, nullptr, stats, rpb);
        } else {
            hipLaunchKernelGGL((wino4_splitk_reduce_kernel<0, false>), dim3(P), dim3(256), 0, st, slabs, p.ksplit, (int64_t)p.zslab, z, ldz, Mpix, Cout, nullptr, nullptr, nullptr, rpb);
        }
        return ds::check_launch("ds_conv_wino4_splitk");
    }
    return ds::check_launch(ar ? "ds_conv_wino4_bf16x2" : "ds_conv_wino4");
}
}  // namespace

extern "C" int ds_conv_wino4_splitk_choose(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (!ds_conv_wino4_supported(H, W, Cin, Cout) || N <= 0) return 1;
    return w4_splitk_choose(N, H, W, Cin, Cout);
}

extern "C" int ds_conv_wino4_splitk_partials(int32_t N, int32_t H, int32_t W) {
    const int64_t M = (int64_t)N * H * W;
    const int rpb = splitk_rpb(M);
    return (int)((M + rpb - 1) / rpb);
}

extern "C" size_t ds_conv_wino4_splitk_workspace(int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t splits) {
    return splits > 1 ? (size_t)splits * N * H * W * Cout * sizeof(float) : 0;
}

extern "C" int ds_conv_wino4_splitk(const float *x, const float *u, float *z, float *stats, const float *pivot, const float *ymask,
                                    int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout,
                                    int32_t ldz, int32_t flags, int32_t splits, void *ws, size_t ws_bytes, void *stream) {
    DS_REQUIRE(splits >= 2 && ws && ws_bytes >= ds_conv_wino4_splitk_workspace(N, H, W, Cout, splits) && (((uintptr_t)ws) & 15) == 0,
               "ds_conv_wino4_splitk: needs >= 2 slices and a 16-byte aligned workspace of ds_conv_wino4_splitk_workspace bytes");
    DS_REQUIRE(Cout <= 1024 && (y_dtype == DS_DTYPE_F32 || y_dtype == DS_DTYPE_BF16), "ds_conv_wino4_splitk: Cout <= 1024, y in fp32 / bf16");
    return w4_launch(0, y_dtype == DS_DTYPE_BF16, x, u, z, stats, pivot, ymask, N, H, W, Cin, ldx, Cout, ldz, flags, stream, false,
                     splits, (float *)ws);
}

extern "C" int ds_conv_wino4(const float *x, const float *u, float *z, float *stats, const float *pivot, const float *ymask,
                             int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz,
                             int32_t flags, void *stream) {
    return w4_launch(0, false, x, u, z, stats, pivot, ymask, N, H, W, Cin, ldx, Cout, ldz, flags, stream);
}

// The same convolution for the 16-bit configurations: bf16-rounded operands, the Winograd-domain products from two bf16
// pieces per operand on the bf16 matrix cores (conv_wino4_kernel<.., AR = 1>); u2 from ds_wino4_transform_weights_bf16x2.
extern "C" int ds_conv_wino4_bf16x2(const float *x, const void *u2, float *z, float *stats, const float *pivot, const void *ymask,
                                    int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout,
                                    int32_t ldz, int32_t flags, void *stream) {
    DS_REQUIRE(y_dtype == DS_DTYPE_F32 || y_dtype == DS_DTYPE_BF16, "ds_conv_wino4_bf16x2: y_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16");
    return w4_launch(1, y_dtype == DS_DTYPE_BF16, x, (const float *)u2, z, stats, pivot, (const float *)ymask, N, H, W, Cin, ldx, Cout, ldz,
                     flags, stream);
}

// ... reading x from 16-bit storage (ldx in bf16 elements): the 3x3 input gradients of the 16-bit configurations from the bf16 dz
// that ds_bn_bwd_apply_bf16 wrote; the same bits as ds_conv_wino4_bf16x2 on the fp32 tensor those values were rounded from
extern "C" int ds_conv_wino4_bf16x2_x16(const void *x16, const void *u2, float *z, float *stats, const float *pivot, const void *ymask,
                                        int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout,
                                        int32_t ldz, int32_t flags, void *stream) {
    DS_REQUIRE(y_dtype == DS_DTYPE_F32 || y_dtype == DS_DTYPE_BF16, "ds_conv_wino4_bf16x2_x16: y_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16");
    return w4_launch(1, y_dtype == DS_DTYPE_BF16, (const float *)x16, (const float *)u2, z, stats, pivot, (const float *)ymask, N, H, W, Cin,
                     ldx, Cout, ldz, flags, stream, true);
}
