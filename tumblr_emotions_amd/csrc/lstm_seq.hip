// BasicLSTMCell under tf.nn.dynamic_rnn for the WHOLE sequence in one launch per direction.
//
// Replaces the per-step pair (recurrent GEMM launch + cell launch) of image_text_model/im_text_rnn_model.py:89-92
// (text_model/text_embedding.py:79-82): at T = 32 that was 31 + 32 dependent launches forward and as many
// backward, each a 256 x 2048 x 512 GEMM that cannot fill the chip.  Here:
//   * the recurrent weights Wh [H, 4H] never move: workgroup (cg, rg) owns hidden units [16 cg, 16 cg + 16) for
//     the batch rows of row group rg (RB = 32 rows, or 16 where the batch leaves CUs idle: pick_rb) and keeps ITS
//     slice of Wh in registers as MFMA B fragments for all T steps (64 columns x H for the forward gates, 16 rows x
//     4H for the backward product), split over the waves along K;
//   * the cell state c (forward) / the carried gradients dc, dh (backward) of those RB x 16 cells live in
//     registers for the whole sequence; gates, masking (t >= seq_len copies the state through) and the
//     last-valid-step capture (h[T] is gather_nd(outputs, seq_len - 1)) are fused into the step;
//   * per step the only exchange is h_t (forward) / dgates_t (backward) among the H/16 workgroups of ONE row
//     group -- batch rows are independent -- through the placement-independent protocol of
//     cdna_hip_programming.md Guideline 16: payload written write-through (sc1 stores), every storing wave
//     drains (s_waitcnt vmcnt(0)), one lane bumps the row group's arrival counter with a relaxed agent-scope
//     atomic; consumers poll that one word relaxed from one lane, then read the payload with sc1 loads (served
//     past the non-coherent L1, so no acquire fence is needed).  Every spin is bounded: on a timeout an error
//     word is set and the kernel still terminates.
//   * round 4: the payload travels through a two-slot ring in MFMA-FRAGMENT order (X[t & 1][row group][k / 4][row][4]:
//     a wave's A load is 1 KB of consecutive bytes; out of the [B, 4H] tensor it touched sixteen half-used lines and
//     the backward step was bound by exactly that), and where the MEASURED placement (HW_REG_XCC_ID, group_on_one_xcd)
//     puts a whole row group on one XCD the ring is written with plain stores that stay in that XCD's L2.  h / dgates
//     themselves are written behind the publish for the kernels that follow.
//   * the input projection x_t Wx + b stays hoisted in one big GEMM in front (ds_conv_igemm), the two weight
//     gradients in two big wgrad GEMMs behind (ds_conv_wgrad), as before.
// Numerics: fp32 MFMA (v_mfma_f32_32x32x2_f32 forward, v_mfma_f32_16x16x4_f32 backward), the K reduction split
// in NW fixed slices summed in a fixed order: deterministic; matches the step-wise path to rounding.
#include <stdlib.h>
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kSC1 = 16;                       // buffer-instruction aux bit: sc1 (system-coherent level 1)
constexpr unsigned kSpinLimit = 1u << 22;      // ~ seconds; a healthy step waits microseconds

struct SeqParams {
    float *gates;              // [T, B, 4H]  fwd: in x_t Wx + b, out activations (i, j, f, o); bwd: activations (in)
    const float *wh;           // [H, 4H] row-major with row stride ldw (rows [D, D+H) of the TF kernel)
    int ldw;
    float *h;                  // [T+1, B, H]  h[0] = initial state (zeros in the reference)
    float *c;                  // [T+1, B, H]
    const float *dh_last;      // bwd: d(loss)/d(h[T])  [B, ld_dh]
    int ld_dh;
    float *dgates;             // bwd out [T, B, 4H]
    const int64_t *seq_len;
    int T, B, H;
    float forget_bias;
    unsigned *sync;            // this direction's [row groups] arrival counters
    unsigned *xcc;             // this direction's [row groups][64] placement words (group_on_one_xcd), zeroed per launch
    float *xchg;               // two-slot exchange ring in fragment order: fwd 2 x B' x H floats, bwd 2 x B' x 4H (B' = B rounded up to 32)
    unsigned *err;             // this direction's error word (sticky: launches never clear it)
    int nrg;
    int ncg, nrgw;             // unit blocks (H / 16); workgroup rows (ceil(nrg / R))
    int xcd_map;               // 1: 1-D XCD-local launch (see wg_coords)
    int strided;               // row groups of a workgroup strided over the workgroup rows (skip_masked launches)
    int skip_masked;           // 1: a row group stops exchanging after its longest row's last step (DS_LSTM_SKIP_MASKED)
    unsigned long long *prof;  // tuning aid (ds_debug_lstm_seq_set_profile): workgroup (0,0) stamps its phases, [T][8]
};

#define DS_STAMP(slot)                                                                  \
    do {                                                                                \
        if (p.prof && tid == 0 && cg == 0 && yb == 0)                                       \
            p.prof[(int64_t)t * 8 + (slot)] = __builtin_amdgcn_s_memtime();             \
    } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd_of(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// Which (unit block cg, workgroup row yb) is this workgroup?  The H / 16 workgroups of a row group exchange h_t / dgates_t
// every step: 64 KB per workgroup and step at H = 512, 16 MB per step over the whole launch at B = 256.  With the plain
// 2-D grid a row group's workgroups land on all eight XCDs (workgroup id % 8, observed placement) and that traffic crosses
// the fabric -- it was the 6.5 us "wait" of a 15.5 us step.  xcd_map: a 1-D grid whose id % 8 picks the XCD and whose
// id / 8 walks the unit blocks of the row groups assigned to that XCD (yb = xcd, xcd + 8, ...), so one row group's
// exchange stays inside one XCD's L2.  Only used when a row group fits an XCD's 32 CUs at one workgroup per CU
// (H <= 512); a different hardware placement only costs speed -- the hand-off protocol is placement independent.
__device__ __forceinline__ bool wg_coords(const SeqParams &p, int &cg, int &yb) {
    if (p.xcd_map) {
        const int lin = blockIdx.x, s = lin >> 3;
        cg = s % p.ncg;
        yb = (lin & 7) + 8 * (s / p.ncg);
        return yb < p.nrgw;
    }
    cg = blockIdx.x;
    yb = blockIdx.y;
    return true;
}

// one lane waits until the row group's counter reaches `target`; bounded
__device__ __forceinline__ void wait_counter(unsigned *cnt, unsigned target, unsigned *err) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// write-through store of 1 or 2 floats (the hand-off payload)
template <int N>
__device__ __forceinline__ void store_sc1(float *p, const float *v) {
    if (N == 2) {
        unsigned long long x = ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v[0]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Do all `ncg` workgroups of a row group run on ONE XCD?  Then the per-step payload can stay in that XCD's L2: plain
// stores keep their lines there (an sc1 store writes through AND drops the line: the readers then fetch at the cross-XCD
// rate, MI355X_MICROARCH.md hand-off table -- 128 KB per workgroup and step took 8.8 of the backward step's 10.8 us),
// `s_waitcnt vmcnt(0)` = acknowledged by the L2, and the readers' sc1 loads bypass only their L1.  The launch geometry
// aims at that placement (wg_coords) but nothing guarantees it, so it is MEASURED: every workgroup publishes its XCC id
// and reads the others' (write-through words, bounded spin).  All members read the same table, so they take the same
// decision; a row group that is spread over XCDs keeps the write-through payload.  Called by all threads.
__device__ __forceinline__ bool group_on_one_xcd(unsigned *tab, int cg, int ncg, unsigned *err) {
    __shared__ int verdict;
    unsigned me;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(me));
    me = (me & 15u) + 1u;
    const int tid = threadIdx.x;
    if (tid == 0) __hip_atomic_store(tab + cg, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        bool same = true;
        if (tid < ncg) {
            unsigned v = 0, spins = 0;
            while ((v = __hip_atomic_load(tab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            same = v == me;
        }
        const bool all_same = __all(same);
        if (tid == 0) verdict = all_same ? 1 : 0;
    }
    __syncthreads();
    return verdict != 0;
}

// ================================================================================================
// forward: H = 8 * NCH * NW; workgroup tile 32 rows x 64 gate columns (16 units x i,j,f,o), K = H split over NW waves
// ================================================================================================
// RB = rows of a row group.  32: v_mfma_f32_32x32x2_f32, two 32-column blocks.  16 (small batches, H <= 512): v_mfma_f32_16x16x4_f32,
// four 16-column blocks -- the step of a 32-row group is bound by ONE CU's matrix rate (32 x 64 x H x 2 flops: 3.4 us at
// H = 512, scripts/lstm_phase_prof.py: 6.1 of an 8.6 us step are fragment loads + MFMAs); with 16-row groups a batch that
// leaves CUs idle spreads over twice as many of them and every workgroup's loads and MFMAs halve.
template <int NCH, int NW, int R, int RB = 32>      // R row groups per workgroup (the `rows` argument)
__global__ __launch_bounds__(64 * NW, NW / 4) void lstm_seq_fwd_kernel(const SeqParams p) {
    constexpr int H = 8 * NCH * NW, KQ = 8 * NCH;
    constexpr int UPT = RB * 16 / (64 * NW);       // hidden units per thread in the cell phase (RB x 16 cells / threads)
    static_assert(UPT >= 1, "16-row groups need four waves");
    constexpr int LDR = 68;                        // padded row of the reduction buffer
    __shared__ __attribute__((aligned(16))) float red[NW * RB * LDR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = RB == 32 ? lane & 31 : lane & 15, kh = RB == 32 ? lane >> 5 : lane >> 4;     // RB = 16: kh = k quarter
    int cg, yb;
    if (!wg_coords(p, cg, yb)) return;                  // (uniform) surplus workgroup of the XCD-local launch
    // Row groups of this workgroup: rgb + rr * rgs.  Contiguous (yb * R + rr) normally; on length-sorted batches with the masked
    // steps skipped (skip_masked) strided over the workgroup rows, so every workgroup walks long AND short row groups and all of
    // them finish together (contiguous: the first workgroup row holds the longest R groups and sets the launch's duration)
    const bool strided = R > 1 && p.strided;
    const int rg0 = strided ? yb : yb * R, rgs = strided ? p.nrgw : 1;
    const int u0 = cg * 16;
    const int ncg = p.ncg;
    const int B = p.B, T = p.T;
    unsigned *err = p.err;

    // ---- this wave's slice of Wh as MFMA B fragments: column c64 = 16 g + u  <->  Wh column g H + u0 + u ------
    // RB = 32: bfr[cb][q][j] = Wh[wave KQ + 8 q + 4 kh + j][column 32 cb + li]      (q < NCH, two column blocks)
    // RB = 16: bfr[cb][q][j] = Wh[wave KQ + 16 q + 4 kh + j][column 16 cb + li]     (q < NCH / 2, four column blocks)
    constexpr int NCB = RB == 32 ? 2 : 4, NQ = RB == 32 ? NCH : NCH / 2;
    float bfr[NCB][NQ][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int c64 = (RB == 32 ? 32 : 16) * cb + li;
        const int col = (c64 >> 4) * H + u0 + (c64 & 15);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[cb][q][j] = p.wh[(int64_t)(wave * KQ + (RB == 32 ? 8 : 16) * q + 4 * kh + j) * p.ldw + col];
    }

    // ---- cell ownership: thread -> (row, UPT consecutive units), for each of the R row groups ----------------
    const int crow = tid / (16 / UPT), cu = (tid % (16 / UPT)) * UPT;
    int grow[R], arow[R];
    bool valid[R];
    int64_t sl[R];
    float cst[R][UPT], hst[R][UPT];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int r0 = (rg0 + rr * rgs) * RB;
        grow[rr] = r0 + crow;
        valid[rr] = grow[rr] < B;
        sl[rr] = valid[rr] ? p.seq_len[grow[rr]] : 0;
#pragma unroll
        for (int e = 0; e < UPT; ++e) {
            cst[rr][e] = valid[rr] ? p.c[(int64_t)grow[rr] * H + u0 + cu + e] : 0.f;
            hst[rr][e] = valid[rr] ? p.h[(int64_t)grow[rr] * H + u0 + cu + e] : 0.f;
        }
        arow[rr] = (r0 + li < B) ? r0 + li : B - 1;        // rows past the batch read a valid row, results unused
    }

    const __amdgpu_buffer_rsrc_t srd_h = srd_of(p.h, (unsigned)((int64_t)(T + 1) * B * H * 4));
    bool l2_local = false;                              // (uniform over the row group)
    if (R == 1 && p.xcd_map && ncg <= 64) l2_local = group_on_one_xcd(p.xcc + rg0 * 64, cg, ncg, err);
    // l2_local: h_t travels through a two-slot ring in fragment order, X[t & 1][row group][k / 4][row][4] (see the backward
    // kernel): a wave's A load reads 1 KB of consecutive bytes instead of 16 / 32 row pieces.  h itself is written behind
    // the publish (the head, the weight gradient and the caller read it); step 0 reads the initial state from h[0].
    const __amdgpu_buffer_rsrc_t srd_x = srd_of(p.xchg, (unsigned)((int64_t)2 * p.nrg * RB * H * 4));
    constexpr unsigned kSlot = (unsigned)(H / 4) * RB * 16u;           // bytes of one row group's slot
    constexpr unsigned kQuads = RB == 32 ? 2u : 4u;                    // K quads per A chunk (8 / 16 channels)

    // DS_LSTM_SKIP_MASKED: past the LONGEST row of a row group every step only copies the state through (dynamic_rnn's
    // masking): no recurrent GEMM, no exchange -- the workgroups of that row group (they all see the same lengths)
    // write the carried h / c and go on.  With the batch sorted by length (ds_seq_sort_desc) a row group's rows have
    // nearly the same length, so the launch does ~mean(length) / T of the steps; per-row results are unchanged to the bit.
    __shared__ int tg_s[R > 8 ? R : 8];
    if (tid < R) tg_s[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
        if (valid[rr]) atomicMax(&tg_s[rr], (int)(sl[rr] < (int64_t)T ? sl[rr] : (int64_t)T));
    __syncthreads();
    int tg[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) tg[rr] = p.skip_masked ? tg_s[rr] : T;

    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            if (rg0 + rr * rgs >= p.nrg) continue;                    // (uniform) a last workgroup with fewer row groups
            if (t >= tg[rr]) {                                  // (uniform) every row of the group is past its length
                if (valid[rr]) {
                    const int64_t o = ((int64_t)(t + 1) * B + grow[rr]) * H + u0 + cu;
#pragma unroll
                    for (int e = 0; e < UPT; ++e) {
                        p.c[o + e] = cst[rr][e];
                        p.h[o + e] = hst[rr][e];
                    }
                }
                continue;
            }
            unsigned *cnt = p.sync + rg0 + rr * rgs;
            // pre-activations of this thread's cells (hoisted x_t Wx + b): independent of the other workgroups
            float gp[4][UPT];
            if (valid[rr]) {
                const float *g = p.gates + ((int64_t)t * B + grow[rr]) * 4 * H + u0 + cu;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < UPT; ++e) gp[k][e] = g[k * H + e];
            }
            if (rr == 0) DS_STAMP(0);
            if (t > 0) {                                        // h[t] of the whole row group must have landed
                if (tid == 0) wait_counter(cnt, (unsigned)t * ncg, err);
                __syncthreads();
            }
            if (rr == 0) DS_STAMP(1);
            // ---- A fragments straight from h[t] (sc1: past the L1, which other CUs' stores never refresh) --------
            f32x4 a[NQ];
            // (uniform) step 0 reads the initial state from h[0]; row groups spread over XCDs keep the [B, H] exchange (their
            // write-through ring stores would be 8-byte pieces 512 bytes apart: forward launch 0.91 -> 1.08 ms beside the
            // image tower; the backward launch, which reads four times the bytes, gains even so)
            const bool ring = l2_local && t > 0;
            const unsigned abase = ring ? ((unsigned)(t & 1) * (unsigned)p.nrg + (unsigned)(rg0 + rr * rgs)) * kSlot +
                                              (unsigned)(((wave * (KQ / 4) + kh) * RB + li) * 16)
                                        : (unsigned)((((int64_t)t * B + arow[rr]) * H + wave * KQ + 4 * kh) * 4);
            const unsigned qstep = ring ? kQuads * RB * 16u : (RB == 32 ? 32u : 64u);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                a[q] = __builtin_bit_cast(f32x4, ring ? __builtin_amdgcn_raw_buffer_load_b128(srd_x, abase + qstep * q, 0, kSC1)
                                                      : __builtin_amdgcn_raw_buffer_load_b128(srd_h, abase + qstep * q, 0, kSC1));
            // ---- K slices of the NW waves -> LDS, summed in wave order by the cell threads ------------------------
            if constexpr (RB == 32) {
                f32x16 acc[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], bfr[cb][q][j], acc[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * LDR + 32 * cb + li] = acc[cb][r];
            } else {
                f32x4 acc[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], bfr[cb][q][j], acc[cb], 0, 0, 0);
                // C layout of the 16x16 MFMA: lane (li, kh) holds rows 4 kh .. 4 kh + 3 of column li
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kh + r) * LDR + 16 * cb + li] = acc[cb][r];
            }
            if (rr == 0) DS_STAMP(2);
            __syncthreads();
            if (rr == 0) DS_STAMP(3);
            float act[4][UPT], hn[UPT];
            if (valid[rr]) {
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < UPT; ++e) gp[k][e] += red[(w * RB + crow) * LDR + 16 * k + cu + e];
                const bool live = (int64_t)t < sl[rr];
#pragma unroll
                for (int e = 0; e < UPT; ++e) {
                    const float si = sigm(gp[0][e]), tj = tanhf(gp[1][e]);
                    const float sf = sigm(gp[2][e] + p.forget_bias), so = sigm(gp[3][e]);
                    const float c_new = cst[rr][e] * sf + si * tj;
                    const float h_new = tanhf(c_new) * so;
                    act[0][e] = si; act[1][e] = tj; act[2][e] = sf; act[3][e] = so;
                    cst[rr][e] = live ? c_new : cst[rr][e];      // dynamic_rnn copies the state through past seq_len
                    hst[rr][e] = hn[e] = live ? h_new : hst[rr][e];
                }
                // the hand-off payload goes first ...
                if (l2_local) {
                    float *xs = p.xchg + ((int64_t)(((t + 1) & 1) * p.nrg + rg0 + rr * rgs) * (H / 4) * RB) * 4 +
                                (((u0 + cu) >> 2) * RB + crow) * 4 + ((u0 + cu) & 3);
#pragma unroll
                    for (int e = 0; e < UPT; ++e) xs[e] = hn[e];
                } else {
                    store_sc1<UPT>(p.h + ((int64_t)(t + 1) * B + grow[rr]) * H + u0 + cu, hn);
                }
            }
            if (rr == 0) DS_STAMP(4);
            // ---- publish: every storing wave drains, then one lane counts this workgroup in ----------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (rr == 0) DS_STAMP(5);
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rr == 0) DS_STAMP(6);
            if (valid[rr]) {                                    // ... what only later kernels read stays off the critical path
                float *g = p.gates + ((int64_t)t * B + grow[rr]) * 4 * H + u0 + cu;
                const int64_t o = ((int64_t)(t + 1) * B + grow[rr]) * H + u0 + cu;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < UPT; ++e) g[k * H + e] = act[k][e];
#pragma unroll
                for (int e = 0; e < UPT; ++e) p.c[o + e] = cst[rr][e];
                if (l2_local) {
#pragma unroll
                    for (int e = 0; e < UPT; ++e) p.h[o + e] = hn[e];
                }
            }
        }
    }
}

// ================================================================================================
// backward: d(h_t)[rows, own 16 units] = carried part + dgates_{t+1}[rows, :] * Wh[own units, :]^T  (K = 4H)
// ================================================================================================
template <int NQ, int NW, int R, int RB = 32>       // 4H = 16 * NQ * NW; R row groups of RB = 32 / 16 rows per workgroup
__global__ __launch_bounds__(64 * NW, NW / 4) void lstm_seq_bwd_kernel(const SeqParams p) {
    constexpr int H4 = 16 * NQ * NW, H = H4 / 4, KQ = 16 * NQ;
    constexpr int UPT = RB * 16 / (64 * NW);
    static_assert(UPT >= 1, "16-row groups need four waves");
    constexpr int NRB = RB / 16;                            // 16-row blocks of the 16x16x4 MFMA
    constexpr int GQ = NQ < 8 ? NQ : (NW == 8 ? 4 : 8);     // A chunks in flight per register group
    constexpr int LDR = 20;
    __shared__ __attribute__((aligned(16))) float red[NW * RB * LDR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kb = lane >> 4;
    int cg, yb;
    if (!wg_coords(p, cg, yb)) return;
    // Row groups of this workgroup: rgb + rr * rgs.  Contiguous (yb * R + rr) normally; on length-sorted batches with the masked
    // steps skipped (skip_masked) strided over the workgroup rows, so every workgroup walks long AND short row groups and all of
    // them finish together (contiguous: the first workgroup row holds the longest R groups and sets the launch's duration)
    const bool strided = R > 1 && p.strided;
    const int rg0 = strided ? yb : yb * R, rgs = strided ? p.nrgw : 1;
    const int u0 = cg * 16;
    const int ncg = p.ncg;
    const int B = p.B, T = p.T;
    unsigned *err = p.err;

    // ---- Wh[u0 + n, this wave's K range] as B fragments of the 16x16x4 MFMA (k-contiguous rows) ----------------
    f32x4 bfr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        bfr[q] = *reinterpret_cast<const f32x4 *>(p.wh + (int64_t)(u0 + li) * p.ldw + wave * KQ + 16 * q + 4 * kb);

    const int crow = tid / (16 / UPT), cu = (tid % (16 / UPT)) * UPT;
    int grow[R];
    bool valid[R];
    int64_t sl[R];
    float dcs[R][UPT], dhc[R][UPT];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int r0 = (rg0 + rr * rgs) * RB;
        grow[rr] = r0 + crow;
        valid[rr] = grow[rr] < B;
        sl[rr] = valid[rr] ? p.seq_len[grow[rr]] : 0;
#pragma unroll
        for (int e = 0; e < UPT; ++e) {
            dcs[rr][e] = 0.f;
            dhc[rr][e] = valid[rr] ? p.dh_last[(int64_t)grow[rr] * p.ld_dh + u0 + cu + e] : 0.f;      // gradient of h[T]
        }
    }

    bool l2_local = false;                              // (uniform over the row group)
    if (R == 1 && p.xcd_map && ncg <= 64) l2_local = group_on_one_xcd(p.xcc + rg0 * 64, cg, ncg, err);
    // l2_local: the exchange goes through a two-slot ring in FRAGMENT order, X[t & 1][row group][k / 4][row][4]: lane
    // (li, kb)'s 16-byte A chunk of K quad 4 q + kb sits next to its neighbours', so a wave's load instruction reads 1 KB of
    // consecutive bytes -- out of the [B, 4H] tensor it touched sixteen half-used lines of sixteen rows, and the 128 KB a
    // workgroup reads per step took 9 of the step's 11 us.  Slot parity is safe: a workgroup writes X[t - 1] only after
    // every member has published step t, i.e. after they have all read X[t + 1].  dgates itself is still written for the
    // weight-gradient GEMMs, behind the publish.
    const __amdgpu_buffer_rsrc_t srd_x = srd_of(p.xchg, (unsigned)((int64_t)2 * p.nrg * RB * H4 * 4));
    constexpr unsigned kSlot = (unsigned)(H4 / 4) * RB * 16u;          // bytes of one row group's slot

    // DS_LSTM_SKIP_MASKED (see the forward kernel): steps past the row group's longest row have dgates = 0 and carry the
    // gradient of h through unchanged -- zeros are written, nothing is exchanged; the group's first real step (t = tg - 1)
    // plays the role of t = T - 1: no recurrent term yet, no wait
    __shared__ int tg_s[R > 8 ? R : 8];
    if (tid < R) tg_s[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
        if (valid[rr]) atomicMax(&tg_s[rr], (int)(sl[rr] < (int64_t)T ? sl[rr] : (int64_t)T));
    __syncthreads();
    int tg[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) tg[rr] = p.skip_masked ? tg_s[rr] : T;

    for (int t = T - 1; t >= 0; --t) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            if (rg0 + rr * rgs >= p.nrg) continue;
            if (t >= tg[rr]) {                                  // (uniform) masked for every row of the group
                if (valid[rr]) {
                    const int64_t gz = ((int64_t)t * B + grow[rr]) * H4 + u0 + cu;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < UPT; ++e) p.dgates[gz + k * H + e] = 0.f;
                }
                continue;
            }
            unsigned *cnt = p.sync + rg0 + rr * rgs;
            float rec[UPT];
#pragma unroll
            for (int e = 0; e < UPT; ++e) rec[e] = 0.f;
            // this thread's saved activations and cell states (written by the forward launch): requested before the
            // hand-off wait, they arrive under it
            const int64_t gi = ((int64_t)t * B + grow[rr]) * H4 + u0 + cu;
            const int64_t ci = ((int64_t)(t + 1) * B + grow[rr]) * H + u0 + cu;      // c_t = c[t+1], c_{t-1} = c[t]
            float a_si[UPT], a_tj[UPT], a_sf[UPT], a_so[UPT], a_ct[UPT], a_cp[UPT];
#pragma unroll
            for (int e = 0; e < UPT; ++e) {
                a_si[e] = valid[rr] ? p.gates[gi + e] : 0.f;
                a_tj[e] = valid[rr] ? p.gates[gi + H + e] : 0.f;
                a_sf[e] = valid[rr] ? p.gates[gi + 2 * H + e] : 0.f;
                a_so[e] = valid[rr] ? p.gates[gi + 3 * H + e] : 0.f;
                a_ct[e] = valid[rr] ? p.c[ci + e] : 0.f;
                a_cp[e] = valid[rr] ? p.c[ci - (int64_t)B * H + e] : 0.f;
            }
            if (rr == 0) DS_STAMP(0);
            if (t < tg[rr] - 1) {
                if (tid == 0) wait_counter(cnt, (unsigned)(tg[rr] - 1 - t) * ncg, err);
                __syncthreads();
                if (rr == 0) DS_STAMP(1);
                f32x4 acc[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                unsigned abase[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    abase[rb] = ((unsigned)((t + 1) & 1) * (unsigned)p.nrg + (unsigned)(rg0 + rr * rgs)) * kSlot +
                                (unsigned)(((wave * (KQ / 4) + kb) * RB + 16 * rb + li) * 16);
                constexpr unsigned qstep = 4u * RB * 16u;                    // bytes between 16-channel chunks
                f32x4 a[2][NRB][GQ];                            // [buffer][row block][chunk]
                auto load_group = [&](int buf, int g0) {
#pragma unroll
                    for (int q = 0; q < GQ; ++q)
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb)
                            a[buf][rb][q] = __builtin_bit_cast(
                                f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, abase[rb] + qstep * (g0 + q), 0, kSC1));
                };
                load_group(0, 0);
#pragma unroll
                for (int g = 0; g < NQ / GQ; ++g) {
                    if (g + 1 < NQ / GQ) load_group((g + 1) & 1, (g + 1) * GQ);
#pragma unroll
                    for (int q = 0; q < GQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int rb = 0; rb < NRB; ++rb)
                                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g & 1][rb][q][j], bfr[g * GQ + q][j], acc[rb],
                                                                               0, 0, 0);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(wave * RB + 16 * rb + 4 * kb + r) * LDR + li] = acc[rb][r];
                if (rr == 0) DS_STAMP(2);
                __syncthreads();
                if (rr == 0) DS_STAMP(3);
                if (valid[rr]) {
#pragma unroll
                    for (int w = 0; w < NW; ++w)
#pragma unroll
                        for (int e = 0; e < UPT; ++e) rec[e] += red[(w * RB + crow) * LDR + cu + e];
                }
            }
            float dg[4][UPT];
            if (valid[rr]) {
                const bool live = (int64_t)t < sl[rr];
#pragma unroll
                for (int e = 0; e < UPT; ++e) {
                    const float dhv = dhc[rr][e] + rec[e];
                    if (live) {
                        const float si = a_si[e], tj = a_tj[e];
                        const float sf = a_sf[e], so = a_so[e];
                        const float tc = tanhf(a_ct[e]);
                        const float dct = dcs[rr][e] + dhv * so * (1.f - tc * tc);
                        dg[0][e] = dct * tj * si * (1.f - si);
                        dg[1][e] = dct * si * (1.f - tj * tj);
                        dg[2][e] = dct * a_cp[e] * sf * (1.f - sf);
                        dg[3][e] = dhv * tc * so * (1.f - so);
                        dcs[rr][e] = dct * sf;
                        dhc[rr][e] = 0.f;
                    } else {            // past seq_len the state was copied through: so is its gradient
                        dg[0][e] = dg[1][e] = dg[2][e] = dg[3][e] = 0.f;
                        dhc[rr][e] = dhv;
                    }
                }
                // the hand-off payload, in fragment order: plain stores where the row group shares an L2, else write-through
                float *xs = p.xchg + ((int64_t)((t & 1) * p.nrg + rg0 + rr * rgs) * (H4 / 4) * RB) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col = k * H + u0 + cu;                    // (UPT = 2: cu is even, one 8-byte piece)
                    float *q = xs + ((col >> 2) * RB + crow) * 4 + (col & 3);
                    if (l2_local) {
#pragma unroll
                        for (int e = 0; e < UPT; ++e) q[e] = dg[k][e];
                    } else {
                        store_sc1<UPT>(q, dg[k]);
                    }
                }
            }
            if (rr == 0) DS_STAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (rr == 0) DS_STAMP(5);
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rr == 0) DS_STAMP(6);
            if (valid[rr]) {                    // ... and dgates for the kernels behind this one, off the hand-off's path
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < UPT; ++e) p.dgates[gi + k * H + e] = dg[k][e];
            }
        }
    }
}

typedef void (*SeqFn)(const SeqParams);
unsigned long long *g_prof = nullptr, *g_prof_bwd = nullptr;

struct SeqCfg {
    SeqFn fwd, bwd;
    int nw;
};

// Row groups per workgroup (the `rows` argument of ds_lstm_seq_fwd / _bwd).  1: every (unit block, row group) pair is a workgroup -- the
// shortest sequence time, but at B = 256, H = 512 that is 256 workgroups of 256-register waves which spend ~40 % of a
// step waiting for their row group: nothing that needs a whole SIMD (the Winograd conv of the image tower) can run
// beside them.  R > 1: a workgroup walks R row groups per time step with the same register-resident Wh slice; the
// hand-off wait of one row group is covered by the work on the others and the launch occupies 1/R of the CUs.

template <int R, int RB>
bool seq_cfg_r(int H, SeqCfg *c) {
    switch (H) {
        case 32:            // (one 8-channel K chunk per wave: no 16-channel chunk for the 16x16x4 form)
            if constexpr (RB == 32) {
                *c = {lstm_seq_fwd_kernel<1, 4, R, 32>, lstm_seq_bwd_kernel<2, 4, R, 32>, 4};
                return true;
            }
            return false;
        case 64: *c = {lstm_seq_fwd_kernel<2, 4, R, RB>, lstm_seq_bwd_kernel<4, 4, R, RB>, 4}; return true;
        case 128: *c = {lstm_seq_fwd_kernel<4, 4, R, RB>, lstm_seq_bwd_kernel<8, 4, R, RB>, 4}; return true;
        case 256: *c = {lstm_seq_fwd_kernel<8, 4, R, RB>, lstm_seq_bwd_kernel<16, 4, R, RB>, 4}; return true;
        case 512: *c = {lstm_seq_fwd_kernel<16, 4, R, RB>, lstm_seq_bwd_kernel<32, 4, R, RB>, 4}; return true;
        case 1024:
            if constexpr (RB == 32) {
                *c = {lstm_seq_fwd_kernel<16, 8, R, 32>, lstm_seq_bwd_kernel<32, 8, R, 32>, 8};
                return true;
            }
            return false;
        default: return false;
    }
}

bool seq_cfg(int H, SeqCfg *c, int rows = 1, int rb = 32) {
    if (rb == 16) return H >= 64 && seq_cfg_r<1, 16>(H, c);        // (16-row groups: rows = 1 only; H = 32 has NQ / NCH too small)
    switch (rows) {
        case 2: return seq_cfg_r<2, 32>(H, c);
        case 4: return seq_cfg_r<4, 32>(H, c);
        case 8: return seq_cfg_r<8, 32>(H, c);
        default: return seq_cfg_r<1, 32>(H, c);
    }
}

// the R in use for a batch of nrg row groups: never more than there are row groups
int rows_for(int rows, int nrg) {
    int r = rows;
    while (r > 1 && r > nrg) r >>= 1;
    return r;
}

// compute units of the current device (init-once property cache; 256 when no device can be queried, e.g. in a
// build container).  Every workgroup of a row group must be resident at once and the kernels ask for one workgroup
// per CU (exclusive_lds), so a launch needs H / 16 CUs.
int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
        (void)hipGetLastError();
    }
    return cus;
}

// One workgroup per CU: two co-resident workgroups would share the CU's four matrix pipes and finish their step
// late, and the whole row group waits for its slowest member (measured: ~3.5 us of a 15 us step was this skew).
// Residency is limited through the LDS request: static reduction buffer + this dynamic pad > half of 160 KiB.
// DS_LSTM_SHARED_CU=1 switches the pad off (A/B aid).
size_t exclusive_lds(int nw, bool fwd, int rb = 32) {
    static int shared_cu = -1;
    if (shared_cu < 0) {
        const char *e = ds::tune_env("DS_LSTM_SHARED_CU");
        shared_cu = e ? atoi(e) : 0;
    }
    if (shared_cu) return 0;
    const size_t stat = (size_t)nw * rb * (fwd ? 68 : 20) * 4;
    const size_t want = 84 * 1024;
    return stat >= want ? 0 : want - stat;
}

// Rows per row group: 16 where the batch leaves CUs idle (rows = 1, H = 64 ... 512, and all ceil(B / 16) * H / 16 workgroups
// resident at one per CU), else 32.  DS_LSTM_RB = 16 / 32 forces it (A/B aid; 16 only where the kernels exist).
int pick_rb(int B, int H, int rows) {
    static int forced = -1;
    if (forced < 0) {
        const char *e = ds::tune_env("DS_LSTM_RB");
        forced = e ? atoi(e) : 0;
    }
    const bool can16 = rows == 1 && H >= 64 && H <= 512;
    if (!can16 || forced == 32) return 32;
    if (forced == 16) return 16;
    return ((B + 15) / 16) * (H / 16) <= device_cus() ? 16 : 32;
}

// workspace words: per direction a block of G = ceil(B / 16) arrival counters (a launch with 32-row groups uses the first
// half) and G x 64 placement words (group_on_one_xcd); behind the two blocks the forward and the backward error word
inline int ws_groups(int B) { return (B + 15) / 16; }
// words of one direction's block: G arrival counters + G x 64 placement words (re-zeroed by every launch of that direction)
inline size_t dir_words(int B) { return (size_t)ws_groups(B) * 65; }
inline size_t ctl_bytes(int B) { return (2 * dir_words(B) + 2 + 63) / 64 * 256; }          // control words, 256-byte aligned
// the backward launch's exchange ring (two slots of dgates_t in fragment order; rows rounded up to 32)
inline size_t ring_bytes(int B, int H) { return (size_t)2 * ((B + 31) / 32 * 32) * 4 * H * sizeof(float); }
// ... and the forward launch's (h_t), behind it
inline size_t ring_bytes_fwd(int B, int H) { return (size_t)2 * ((B + 31) / 32 * 32) * H * sizeof(float); }

// launch geometry: XCD-local 1-D grid when a row group's H / 16 workgroups fit one XCD (32 CUs, one workgroup each);
// DS_LSTM_XCD=0 switches back to the 2-D grid (A/B aid)
dim3 seq_grid(SeqParams &p, int H, int rows) {
    static int use = -1;
    if (use < 0) {
        const char *e = ds::tune_env("DS_LSTM_XCD");
        use = e ? atoi(e) : 1;
    }
    p.ncg = H / 16;
    p.nrgw = (p.nrg + rows - 1) / rows;
    // rows > 1 is the setting for running BESIDE another kernel that needs whole CUs (the image tower's Winograd conv):
    // there the row groups stay spread over all XCDs -- packed onto one or two XCDs they would leave those XCDs no CU
    // for the other kernel's workgroups (joint step 17.8 -> 19.0 ms, measured)
    // Residency: in the 1-D grid the ids interleave up to 8 row groups (id & 7 = row group, id >> 3 = unit block), so
    // the first workgroups the dispatcher places are unit blocks 0.. of ALL of them; a device with fewer than
    // min(nrgw, 8) * ncg CUs (one workgroup per CU, exclusive_lds) would never hold a whole row group and every
    // workgroup would spin on peers that cannot start.  The 2-D grid dispatches row group 0 completely first and needs
    // only ncg CUs (ds_lstm_seq_supported), so smaller / partitioned devices fall back to it.
    const int groups_in_flight = p.nrgw < 8 ? p.nrgw : 8;
    p.xcd_map = (use && p.ncg <= 32 && rows == 1 && device_cus() >= groups_in_flight * p.ncg) ? 1 : 0;
    if (p.xcd_map) return dim3((unsigned)(8 * p.ncg * ((p.nrgw + 7) / 8)));
    return dim3((unsigned)p.ncg, (unsigned)p.nrgw);
}

int common_checks(const char *who, const void *a, const void *b, const void *c, int T, int B, int H, int ldw, int rows,
                  void *ws, size_t ws_bytes) {
    DS_REQUIRE(a && b && c && ws, "%s: null argument", who);
    // (eight row groups per workgroup: dropped in round 5 as never the measured choice, back in round 6 -- with the batch sorted by
    // length and the masked steps skipped a workgroup's eight groups thin out as the steps go, and ONE workgroup row on H / 16 CUs
    // costs the image tower beside it less than two: joint step 13.30 -> 13.24 ms)
    DS_REQUIRE(rows == 1 || rows == 2 || rows == 4 || rows == 8, "%s: rows (row groups per workgroup) must be 1, 2, 4 or 8", who);
    DS_REQUIRE(T > 0 && B > 0 && ds_lstm_seq_supported(B, H),
               "%s: unsupported size (H must be 32 ... 1024, a power of two, and H / 16 workgroups must fit the device's CUs)", who);
    DS_REQUIRE(ldw >= 4 * H && ldw % 4 == 0 && (((uintptr_t)b) & 15) == 0, "%s: Wh must be 16-byte aligned with ldw %% 4 == 0", who);
    DS_REQUIRE((int64_t)(T + 1) * B * 4 * H * 4 < (1ll << 31), "%s: sequence buffers above 2 GiB (split the batch)", who);
    DS_REQUIRE(ws_bytes >= ds_lstm_seq_workspace(B, H), "%s: workspace too small", who);
    return DS_OK;
}

}  // namespace

extern "C" int ds_lstm_seq_supported(int32_t B, int32_t H) {
    SeqCfg c;
    // the H / 16 workgroups of a row group spin on each other: all of them must be resident, one per CU
    return B > 0 && seq_cfg(H, &c) && H / 16 <= device_cus() ? 1 : 0;
}

// workspace: [forward block][backward block][forward error word][backward error word]; a block = G arrival counters
// (G = ceil(B / 16): one per 16-row group; launches with 32-row groups use the first half) + G x 64 placement words.  A
// launch re-zeroes ITS block only; the error words are sticky until ds_lstm_seq_status reports them (the workspace is
// zero-initialised once by the caller).
extern "C" size_t ds_lstm_seq_workspace(int32_t B, int32_t H) {
    (void)H;
    return ctl_bytes(B) + ring_bytes(B, H) + ring_bytes_fwd(B, H);
}

extern "C" int ds_lstm_seq_fwd(float *gates, const float *wh, int32_t ldw, float *h, float *c, const int64_t *seq_len,
                               int32_t T, int32_t B, int32_t H, float forget_bias, int32_t rows_arg, void *ws,
                               size_t ws_bytes, void *stream) {
    const int skip = (rows_arg & DS_LSTM_SKIP_MASKED) ? 1 : 0;
    rows_arg &= ~DS_LSTM_SKIP_MASKED;
    if (int e = common_checks("ds_lstm_seq_fwd", gates, wh, h, T, B, H, ldw, rows_arg, ws, ws_bytes)) return e;
    DS_REQUIRE(c && seq_len, "ds_lstm_seq_fwd: null argument");
    SeqCfg cfg;
    const int rows = rows_for(rows_arg, (B + 31) / 32);
    const int rb = pick_rb(B, H, rows);
    seq_cfg(H, &cfg, rows, rb);
    SeqParams p = {};
    p.gates = gates; p.wh = wh; p.ldw = ldw; p.h = h; p.c = c; p.seq_len = seq_len;
    p.T = T; p.B = B; p.H = H; p.forget_bias = forget_bias;
    p.skip_masked = skip;
    p.strided = skip && !ds::tune_env("DS_LSTM_CONTIG");          // (DS_LSTM_CONTIG: A/B aid, tuning library only)
    p.nrg = (B + rb - 1) / rb;
    p.sync = (unsigned *)ws;
    p.xcc = p.sync + ws_groups(B);
    p.err = (unsigned *)ws + 2 * dir_words(B);
    p.xchg = reinterpret_cast<float *>((char *)ws + ctl_bytes(B) + ring_bytes(B, H));
    p.prof = g_prof;
    // every polled word is re-initialised by a memset node in front of the launch (Guideline 16); the error words
    // are left alone
    if (hipMemsetAsync(p.sync, 0, dir_words(B) * sizeof(unsigned), (hipStream_t)stream) != hipSuccess)
        return ds::check_launch("ds_lstm_seq_fwd(memset)");
    const dim3 grid = seq_grid(p, H, rows);
    hipLaunchKernelGGL(cfg.fwd, grid, dim3(64 * cfg.nw), exclusive_lds(cfg.nw, true, rb), (hipStream_t)stream, p);
    return ds::check_launch("ds_lstm_seq_fwd");
}

extern "C" int ds_lstm_seq_bwd(const float *acts, const float *wh, int32_t ldw, const float *c, const float *dh_last,
                               int32_t ld_dh, const int64_t *seq_len, int32_t T, int32_t B, int32_t H, float *dgates,
                               int32_t rows_arg, void *ws, size_t ws_bytes, void *stream) {
    const int skip = (rows_arg & DS_LSTM_SKIP_MASKED) ? 1 : 0;
    rows_arg &= ~DS_LSTM_SKIP_MASKED;
    if (int e = common_checks("ds_lstm_seq_bwd", acts, wh, c, T, B, H, ldw, rows_arg, ws, ws_bytes)) return e;
    DS_REQUIRE(dh_last && seq_len && dgates && ld_dh >= H, "ds_lstm_seq_bwd: bad argument");
    SeqCfg cfg;
    const int rows = rows_for(rows_arg, (B + 31) / 32);
    const int rb = pick_rb(B, H, rows);
    seq_cfg(H, &cfg, rows, rb);
    SeqParams p = {};
    p.gates = const_cast<float *>(acts); p.wh = wh; p.ldw = ldw; p.c = const_cast<float *>(c);
    p.dh_last = dh_last; p.ld_dh = ld_dh; p.dgates = dgates; p.seq_len = seq_len;
    p.T = T; p.B = B; p.H = H;
    p.skip_masked = skip;
    p.strided = skip && !ds::tune_env("DS_LSTM_CONTIG");          // (DS_LSTM_CONTIG: A/B aid, tuning library only)
    p.nrg = (B + rb - 1) / rb;
    p.sync = (unsigned *)ws + dir_words(B);
    p.xcc = p.sync + ws_groups(B);
    p.err = (unsigned *)ws + 2 * dir_words(B) + 1;
    p.xchg = reinterpret_cast<float *>((char *)ws + ctl_bytes(B));
    p.prof = g_prof_bwd;
    if (hipMemsetAsync(p.sync, 0, dir_words(B) * sizeof(unsigned), (hipStream_t)stream) != hipSuccess)
        return ds::check_launch("ds_lstm_seq_bwd(memset)");
    const dim3 grid = seq_grid(p, H, rows);
    hipLaunchKernelGGL(cfg.bwd, grid, dim3(64 * cfg.nw), exclusive_lds(cfg.nw, false, rb), (hipStream_t)stream, p);
    return ds::check_launch("ds_lstm_seq_bwd");
}

extern "C" int ds_lstm_seq_status(void *ws, int32_t B) {
    // host-side read of the two error words of FINISHED launches (the caller synchronised): 0 = ok, bit 0 = a
    // hand-off wait of a forward launch timed out (a workgroup of the row group never became resident), bit 1 = the
    // same in a backward launch; the results of the launches since the last call are invalid.  The words are sticky
    // across launches and CLEARED by this call once reported, so a later, healthy step is not blamed for an old one.
    unsigned v[2] = {0, 0};
    unsigned *err = (unsigned *)ws + 2 * dir_words(B);
    if (hipMemcpy(v, err, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return DS_ERR_LAUNCH;
    if ((v[0] | v[1]) && hipMemset(err, 0, sizeof(v)) != hipSuccess) return DS_ERR_LAUNCH;
    return (int)((v[0] ? 1u : 0u) | (v[1] ? 2u : 0u));
}

// Debug aid (never called by the product path; process-global, not re-entrant): device buffer of T*8 uint64 in which
// workgroup (0,0) of the NEXT ds_lstm_seq_fwd launches stamps s_memtime at its phase boundaries; NULL switches it off.
#ifdef DS_TUNING
extern "C" int ds_debug_lstm_seq_set_profile(void *buf) {
    g_prof = (unsigned long long *)buf;
    return DS_OK;
}
#endif

// the same for the NEXT ds_lstm_seq_bwd launches (stamp row t of step t; the walk goes from T - 1 down to 0)
#ifdef DS_TUNING
extern "C" int ds_debug_lstm_seq_set_profile_bwd(void *buf) {
    g_prof_bwd = (unsigned long long *)buf;
    return DS_OK;
}
#endif
