// fp8 convolution path (BASELINE configs[4]: "fp16 joint with fp8 (CDNA4) MFMA conv path"): 1x1 and 3x3 convs, forward
// and Conv2DBackpropInput, on v_mfma_f32_32x32x16_fp8_fp8 / _bf8_fp8 with per-tensor power-of-two scales and fp32
// accumulation.  Call sites replaced: slim.conv2d of image_model/inception_v1.py:71-250 and its input gradient (implied
// by create_train_op, image_text_model/im_text_rnn_model.py:135).  NOT the fp32 parity path: its own label, its own
// documented tolerance (tests/test_kernels_gpu.py, tests/test_model_gpu.py).
//
//   * formats (gfx950 = OCP): weights and forward activations e4m3 (max 448), gradients dz e5m2 (max 57344);
//   * scales: s = 2^floor(log2(FMAX / amax)) per tensor -- a power of two, so scaling itself never rounds.  The
//     weight scale is fixed when the filter is converted (ds_weights_to_fp8: amax pass + conversion into the
//     kernel's K-loop order [chunk x tap][column][16 k], 16 bytes per column and iteration); the activation scale
//     is derived IN the kernel from a device word holding max|x| (ds_absmax, or any producer that tracks it), so
//     no value crosses to the host;
//   * structure = conv_bf16d_kernel (conv_igemm.hip): register-direct A (lane (i, kh) loads channels 8 kh .. 8 kh + 7
//     of its pixel as two float4, scales, saturates and converts them with v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32:
//     that IS its A fragment), weights by LDS-DMA, a wave owns 32 pixels x NB*32 columns;
//   * epilogue: z = acc / (s_a * s_w) in fp32, BatchNorm column statistics about the pivot (DS_EPI_STATS).
#include <stdlib.h>
#include "ds_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

namespace {

constexpr unsigned kOOB = 0x80000000u;
constexpr float kMaxE4M3 = 448.f, kMaxE5M2 = 57344.f;

struct Fp8Params {
    ds_conv_desc d;
    const float *x;
    const unsigned char *w;      // [it][ncols][16] fp8
    const float *x_amax;         // device word: max |x|
    const float *wscale;         // float[4]: amax, s_w, 1 / s_w, -
    float *z;
    const float *mask;           // DS_EPI_BNSUMS: the consumer layer's activation (fp32 or bf16 storage, d.mask_dtype)
    float *stats;
    const float *pivot;
    int M, row_tiles, col_tiles, ncols;
    unsigned x_bytes, w_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// 2^floor(log2(fmax / amax)): the largest power of two that keeps amax * s <= fmax (1 for an all-zero tensor)
__host__ __device__ __forceinline__ float pow2_scale(float amax, float fmax) {
    if (!(amax > 0.f)) return 1.f;
    const float r = fmax / amax;
    union { float f; unsigned u; } c;
    c.f = r;
    int e = (int)((c.u >> 23) & 0xff) - 127;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    c.u = (unsigned)(e + 127) << 23;
    return c.f;
}

// SAT: saturate in software (the conversion instruction's own overflow behaviour depends on a mode bit).  The A
// operand needs none: its scale comes from a max|x| that bounds every element (the tensor's own ds_absmax, or its
// producer's record), so |x s| <= FMAX by construction -- and every VALU slot of the fragment build is a slot the
// matrix pipe waits for.
template <bool E5M2, bool SAT = true>
__device__ __forceinline__ int cvt_pk(float a, float b, int old, bool hi) {
    if (SAT) {
        constexpr float m = E5M2 ? kMaxE5M2 : kMaxE4M3;
        a = __builtin_fminf(__builtin_fmaxf(a, -m), m);
        b = __builtin_fminf(__builtin_fmaxf(b, -m), m);
    }
    if (E5M2) return hi ? __builtin_amdgcn_cvt_pk_bf8_f32(a, b, old, true) : __builtin_amdgcn_cvt_pk_bf8_f32(a, b, old, false);
    return hi ? __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, false);
}

template <int NB, int KS, bool E5M2, bool XB>      // KS x KS taps (1 or 3); E5M2: the A operand is a gradient (e5m2); XB: x stored as bf16
__global__ __launch_bounds__(256, 2) void conv_fp8d_kernel(const Fp8Params p) {
    constexpr int BN = NB * 32, SI = 4, TAPS = KS * KS;
    constexpr int BSZ = SI * BN * 16;                          // bytes per B buffer
    constexpr int DJ = (BSZ + 4095) / 4096;                    // 16-byte DMA slots per thread and step
    __shared__ __attribute__((aligned(128))) unsigned char smem[2 * DJ * 4096 + 1024];
    const ds_conv_desc &d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    // 1-D XCD-aware launch, as conv_igemm.hip's TileId: column tiles of a row tile run back to back on one XCD
    const int id = blockIdx.x;
    const int lin = (id & 7) * (int)(gridDim.x >> 3) + (id >> 3);
    const int trow = lin / p.col_tiles, tcol = lin - trow * p.col_tiles;
    const int n0 = tcol * BN;
    const bool item = trow < p.row_tiles;
    const __amdgpu_buffer_rsrc_t srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.w, p.w_bytes);
    const int chunks = (d.Cin + 15) >> 4;
    const int iters = chunks * TAPS;
    const int nsteps = (iters + SI - 1) / SI;
    const float sa = pow2_scale(ds::amax_read(p.x_amax), E5M2 ? kMaxE5M2 : kMaxE4M3);
    const float inv = p.wscale[2] / sa;                        // 1 / (s_a s_w): exact, both are powers of two

    const int m = trow * 128 + wave * 32 + li;
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
    // B DMA slots: 16-byte slot sl of a step = (local iteration sl / BN, column sl % BN); the source is linear in w
    unsigned uoff[DJ];
    int uit[DJ];
#pragma unroll
    for (int i = 0; i < DJ; ++i) {
        const int sl = i * 256 + tid;
        const int itl = sl / BN, nn = sl - itl * BN;
        uit[i] = itl;
        uoff[i] = (itl < SI && n0 + nn < p.ncols) ? (unsigned)((itl * p.ncols + n0 + nn) * 16) : kOOB;
    }
    const unsigned it_bytes = (unsigned)p.ncols * 16u;
    auto dma_b = [&](int buf, int step, int i) {
        unsigned o = uoff[i] + (unsigned)(step * SI) * it_bytes;
        if (step * SI + uit[i] >= iters || uoff[i] == 0x80000000u) o = 0x80000000u;     // past the reduction: zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr)(smem + buf * (DJ * 4096) + wave * 1024 + i * 4096), 16, o, 0, 0, 0);
    };
    auto load_a = [&](int it, f32x4 &lo, f32x4 &hi) {
        const int chunk = it / TAPS, tap = it - chunk * TAPS;
        unsigned vo = voff[0];
#pragma unroll
        for (int k = 1; k < TAPS; ++k) vo = tap == k ? voff[k] : vo;
        if (XB) {           // eight bf16 channels in one 16-byte load; widened below
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
    for (int i = 0; i < DJ; ++i) dma_b(0, 0, i);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const bool more = st + 1 < nsteps;
        const unsigned char *b_s = smem + (st & 1) * (DJ * 4096) + li * 16 + kh * 8;
        f32x4 nlo[SI], nhi[SI];
#pragma unroll
        for (int j = 0; j < SI; ++j) {
            // A fragment: 8 consecutive channels, scaled by the power of two, saturated, rounded to fp8 (RNE)
            float a8[8];
            if (XB) {       // bf16 -> fp32 is a 16-bit shift
                const u32x4 raw = __builtin_bit_cast(u32x4, alo[j]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a8[2 * q] = __builtin_bit_cast(float, raw[q] << 16);
                    a8[2 * q + 1] = __builtin_bit_cast(float, raw[q] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { a8[q] = alo[j][q]; a8[4 + q] = ahi[j][q]; }
            }
            int w0 = cvt_pk<E5M2, false>(a8[0] * sa, a8[1] * sa, 0, false);
            w0 = cvt_pk<E5M2, false>(a8[2] * sa, a8[3] * sa, w0, true);
            int w1 = cvt_pk<E5M2, false>(a8[4] * sa, a8[5] * sa, 0, false);
            w1 = cvt_pk<E5M2, false>(a8[6] * sa, a8[7] * sa, w1, true);
            const long af = (long)(((unsigned long long)(unsigned)w1 << 32) | (unsigned)w0);
            if (more) {
                load_a((st + 1) * SI + j, nlo[j], nhi[j]);
#pragma unroll
                for (int i = j; i < DJ; i += SI) dma_b((st + 1) & 1, st + 1, i);
            }
            long bfr[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) bfr[b] = *reinterpret_cast<const long *>(b_s + (j * BN + b * 32) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (E5M2) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(af, bfr[b], acc[b], 0, 0, 0);
                else acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(af, bfr[b], acc[b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < SI; ++j) { alo[j] = nlo[j]; ahi[j] = nhi[j]; }
        __syncthreads();
    }

    // ---- epilogue: unscale, store, BatchNorm column statistics (two phases as gemm_wide_kernel: reads and sums for
    // every column block first, all stores last -- a read behind a store waits for it) ----------------------------------
    const int flags = d.flags;
    float *red = reinterpret_cast<float *>(smem + 2 * DJ * 4096);
    const int mrow0 = trow * 128 + wave * 32;
    // (as gemm_wide_kernel) z / accumulate / activation accesses through buffer descriptors: 32-bit lane offset + row offset
    // in the vector offset, hardware range check for rows past M and columns past Cout (kOOB)
    const bool m16 = d.mask_dtype == DS_DTYPE_BF16;      // (uniform) BatchNorm-sums activation in bf16 storage
    const bool z16 = d.z_dtype == DS_DTYPE_BF16;         // (uniform) z stored as bf16(z - pivot): as conv_bf16d_kernel
    const unsigned zeb = z16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t srd_z = make_srd(p.z, (unsigned)(((int64_t)(p.M - 1) * d.ldz + d.Cout) * zeb));
    const __amdgpu_buffer_rsrc_t srd_m = make_srd((flags & DS_EPI_BNSUMS) ? p.mask : (const void *)p.z,
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
        float zv[16];                                    // DS_EPI_ACCUM: the previous values
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2)
            zv[r2] = (flags & DS_EPI_ACCUM) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_z, vz + roff(r2, rz), 0, 0)) : 0.f;
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) acc[b][r2] = acc[b][r2] * inv + zv[r2];
        if (flags & DS_EPI_BNSUMS) {
            // (as gemm_wide_kernel) dgrad whose result dy feeds a BatchNorm + ReLU backward: column sums of g = dy (y > 0)
            // and g * y; y = the consumer layer's activation in fp32 or bf16 storage
            float yv[16];
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                if (m16) yv[r2] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(srd_m, vm + roff(r2, rm), 0, 0) << 16);
                else yv[r2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_m, vm + roff(r2, rm), 0, 0));
            }
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const float u = yv[r2] > 0.f ? acc[b][r2] : 0.f;          // (out of range: y = 0)
                s += u;
                q += u * yv[r2];
            }
        } else if (flags & DS_EPI_STATS) {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const int row = mrow0 + (r2 & 3) + 8 * (r2 >> 2) + 4 * kh;
                if (row < p.M && colok) {
                    const float u = acc[b][r2] - pv;
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
        if (z16) {          // centred about the statistics pivot and rounded to nearest even (see conv_bf16d_kernel)
            const float pvz = (p.pivot && colok) ? p.pivot[col] : 0.f;
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const __bf16 hv = (__bf16)(acc[b][r2] - pvz);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), srd_z, vz + roff(r2, rz), 0, 2 /* nt */);
            }
        } else {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const float val = acc[b][r2];          // (bit_cast of a vector-element lvalue reads element 0: copy first)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_z, vz + roff(r2, rz), 0, 2 /* nt */);
            }
        }
        if ((flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) && tid < 32 && item && n0 + 32 * b + tid < d.Cout) {
            p.stats[(int64_t)(n0 + 32 * b + tid) * p.row_tiles + trow] = pss[b];
            p.stats[((int64_t)d.Cout + n0 + 32 * b + tid) * p.row_tiles + trow] = pqq[b];
        }
    }
}

// max |x| into out[0] (as float bits; non-negative floats order like unsigned integers, so atomicMax is exact and
// order independent).  out[0] must be zero on entry (the entry point clears it).
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const unsigned *x, int64_t n2, unsigned *out) {
    // n2 pairs of bf16: |v| of a bf16 is its 15 low bits; as fp32 bits that is (bits << 16), and non-negative floats
    // order like unsigned integers
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const unsigned v = x[i];
        const unsigned a = (v & 0x7fffu) << 16, b = v & 0x7fff0000u;
        m = max(m, max(a, b));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) ds::atomic_max_nonneg(reinterpret_cast<float *>(out), __uint_as_float(m));
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *x, int64_t n, unsigned *out) {
    float m = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) ds::atomic_max_nonneg(reinterpret_cast<float *>(out), m);
}

// wq[it][column][16] e4m3, it = chunk * taps + tap, from the TF HWIO filter w [taps][Cin][Cout] scaled by
// s_w = pow2_scale(amax): layout rules as weights_to_bf16_kernel (conv_igemm.hip).  wscale[0] = amax on entry.
__global__ __launch_bounds__(256) void weights_to_fp8_kernel(const float *w, unsigned char *wq, const float *wamax,
                                                            float *wscale, int Cin, int Cout, int taps, int dgrad) {
    const int K = dgrad ? Cout : Cin, Ncol = dgrad ? Cin : Cout;
    const int chunks = (K + 15) >> 4, ncols = (Ncol + 31) / 32 * 32;
    const float wmax = ds::amax_read(wamax);       // every lane (wave shuffles inside)
    const float sw = pow2_scale(wmax, kMaxE4M3);
    const int64_t total = (int64_t)chunks * taps * ncols * 8;            // pairs of consecutive k
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i & 7) * 2;
        const int64_t rest = i >> 3;
        const int col = (int)(rest % ncols);
        const int it = (int)(rest / ncols);
        const int chunk = it / taps, tap = it - chunk * taps;
        float v[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = chunk * 16 + j + e;
            if (col < Ncol && k < K) {
                const int ci = dgrad ? col : k, co = dgrad ? k : col, tp = dgrad ? taps - 1 - tap : tap;
                v[e] = w[((int64_t)tp * Cin + ci) * Cout + co] * sw;
            }
        }
        const int pk = cvt_pk<false>(v[0], v[1], 0, false);
        reinterpret_cast<unsigned short *>(wq)[i] = (unsigned short)(pk & 0xffff);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        wscale[0] = wmax;
        wscale[1] = sw;
        wscale[2] = 1.f / sw;
    }
}

int fp8_nb(int Cout) {
    int best = 8, best_cost = 1 << 30;
    for (int nb = 8; nb >= 1; --nb) {
        const int tiles = (Cout + 32 * nb - 1) / (32 * nb);
        const int cost = tiles * (4 + nb);
        if (cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}

bool fp8_ok(const ds_conv_desc *d) {
    return (d->KH == d->KW) && (d->KH == 1 || d->KH == 3) && d->fold_cin == 0 && d->Cin % 8 == 0 && d->ldx % 4 == 0 &&
           !(d->flags & ~(DS_EPI_STATS | DS_EPI_ACCUM | DS_EPI_BNSUMS)) &&
           !((d->flags & DS_EPI_BNSUMS) && (d->flags & DS_EPI_STATS)) && d->splits <= 1 &&
           // the epilogue addresses z (and the BatchNorm-sums activation) through 32-bit buffer offsets
           (((int64_t)d->N * d->OH * d->OW - 1) * d->ldz + d->Cout) * 4 < (1ll << 31) &&
           (!(d->flags & DS_EPI_BNSUMS) || (((int64_t)d->N * d->OH * d->OW - 1) * d->ldmask + d->Cout) * 4 < (1ll << 31));
}

int64_t fp8_M(const ds_conv_desc *d) { return (int64_t)d->N * d->OH * d->OW; }

template <int NB>
void launch_fp8(const ds_conv_desc *d, int a_format, dim3 grid, hipStream_t st, const Fp8Params &p) {
    const bool xb = d->x_dtype == DS_DTYPE_BF16;      // forward activations only (checked by the caller)
    if (d->KH == 1) {
        if (a_format == DS_FP8_E5M2) hipLaunchKernelGGL((conv_fp8d_kernel<NB, 1, true, false>), grid, dim3(256), 0, st, p);
        else if (xb) hipLaunchKernelGGL((conv_fp8d_kernel<NB, 1, false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_fp8d_kernel<NB, 1, false, false>), grid, dim3(256), 0, st, p);
    } else {
        if (a_format == DS_FP8_E5M2) hipLaunchKernelGGL((conv_fp8d_kernel<NB, 3, true, false>), grid, dim3(256), 0, st, p);
        else if (xb) hipLaunchKernelGGL((conv_fp8d_kernel<NB, 3, false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_fp8d_kernel<NB, 3, false, false>), grid, dim3(256), 0, st, p);
    }
}

}  // namespace

extern "C" int ds_absmax(const void *x, int64_t n, int32_t x_dtype, float *amax, void *stream) {
    DS_REQUIRE(x && amax && n > 0 && (((uintptr_t)x) & 15) == 0, "ds_absmax: bad argument (x must be 16-byte aligned)");
    DS_REQUIRE(x_dtype == DS_DTYPE_F32 || (x_dtype == DS_DTYPE_BF16 && n % 2 == 0),
               "ds_absmax: x_dtype must be DS_DTYPE_F32 or DS_DTYPE_BF16 (an even count)");
    if (hipMemsetAsync(amax, 0, DS_AMAX_FLOATS * sizeof(float), (hipStream_t)stream) != hipSuccess) return ds::check_launch("ds_absmax(memset)");
    if (x_dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL(absmax_bf16_kernel, dim3(ds::stream_grid(n / 2, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned *)x, n / 2, (unsigned *)amax);
    else
        hipLaunchKernelGGL(absmax_kernel, dim3(ds::stream_grid(n / 4 + 1, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)x, n, (unsigned *)amax);
    return ds::check_launch("ds_absmax");
}

extern "C" size_t ds_weights_fp8_bytes(int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad) {
    const int K = dgrad ? Cout : Cin, Ncol = dgrad ? Cin : Cout;
    return (size_t)((K + 15) / 16) * taps * ((Ncol + 31) / 32 * 32) * 16;
}

extern "C" int ds_weights_to_fp8(const float *w, void *wq, float *wscale, int32_t Cin, int32_t Cout, int32_t taps,
                                 int32_t dgrad, void *stream) {
    DS_REQUIRE(w && wq && wscale && Cin > 0 && Cout > 0 && taps > 0, "ds_weights_to_fp8: bad argument");
    const int64_t n = (int64_t)taps * Cin * Cout;
    float *wamax = wscale + 4;                 // the scale record is followed by the amax record of the filter
    if (int e = ds_absmax(w, n, DS_DTYPE_F32, wamax, stream)) return e;
    const int64_t pairs = (int64_t)ds_weights_fp8_bytes(Cin, Cout, taps, dgrad) / 2;
    hipLaunchKernelGGL(weights_to_fp8_kernel, dim3(ds::stream_grid(pairs, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned char *)wq, wamax, wscale, Cin, Cout, taps, dgrad);
    return ds::check_launch("ds_weights_to_fp8");
}

extern "C" int ds_conv_fp8_supported(const ds_conv_desc *d) { return d && fp8_ok(d) ? 1 : 0; }

extern "C" int ds_conv_fp8_partials(const ds_conv_desc *d) { return (int)((fp8_M(d) + 127) / 128); }

extern "C" int ds_conv_fp8(const ds_conv_desc *d, const void *x, const float *x_amax, int32_t a_format, const void *wq,
                           const float *wscale, float *z, const void *mask, float *stats, const float *pivot, void *stream) {
    DS_REQUIRE(d && x && x_amax && wq && wscale && z, "ds_conv_fp8: null argument");
    DS_REQUIRE(!(d->flags & DS_EPI_BNSUMS) || (mask && stats && d->ldmask >= d->Cout &&
                                               (d->mask_dtype == DS_DTYPE_F32 || d->mask_dtype == DS_DTYPE_BF16)),
               "ds_conv_fp8: DS_EPI_BNSUMS needs mask (row stride ldmask, mask_dtype) and a partials buffer");
    DS_REQUIRE(!d->mask_rstd && !d->norm_rstd && !d->bnb, "ds_conv_fp8: the on-load transforms belong to the fp32 kernels");
    DS_REQUIRE(d->x_dtype == DS_DTYPE_F32 || (d->x_dtype == DS_DTYPE_BF16 && a_format == DS_FP8_E4M3 && d->ldx % 8 == 0),
               "ds_conv_fp8: x_dtype DS_DTYPE_BF16 is for forward activations (e4m3) with ldx %% 8 == 0");
    DS_REQUIRE(fp8_ok(d), "ds_conv_fp8: needs a 1x1 or 3x3 conv, Cin %% 8 == 0, ldx %% 4 == 0, flags within DS_EPI_STATS, or "
                          "DS_EPI_ACCUM | DS_EPI_BNSUMS for a dgrad");
    DS_REQUIRE(a_format == DS_FP8_E4M3 || a_format == DS_FP8_E5M2, "ds_conv_fp8: a_format must be DS_FP8_E4M3 or DS_FP8_E5M2");
    DS_REQUIRE(d->z_dtype == DS_DTYPE_F32 || (d->z_dtype == DS_DTYPE_BF16 && !(d->flags & (DS_EPI_ACCUM | DS_EPI_BNSUMS))),
               "ds_conv_fp8: the output in bf16 storage excludes the accumulate / BatchNorm-sums epilogues");
    DS_REQUIRE(((((uintptr_t)x | (uintptr_t)wq) & 15) == 0) && fp8_M(d) < (1ll << 31), "ds_conv_fp8: operands must be 16-byte aligned");
    DS_REQUIRE(!(d->flags & DS_EPI_STATS) || stats, "ds_conv_fp8: DS_EPI_STATS without stats buffer");
    Fp8Params p = {};
    p.d = *d;
    p.x = (const float *)x; p.w = (const unsigned char *)wq; p.x_amax = x_amax; p.wscale = wscale; p.z = z; p.stats = stats;
    p.mask = (const float *)mask;
    p.pivot = (d->flags & DS_EPI_STATS) ? pivot : nullptr;
    p.M = (int)fp8_M(d);
    const int taps = d->KH * d->KW;
    const int64_t x_elems = ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin;
    p.ncols = (d->Cout + 31) / 32 * 32;
    const int64_t wq_bytes = (int64_t)((d->Cin + 15) / 16) * taps * p.ncols * 16;
    DS_REQUIRE(x_elems * 4 < (1ll << 31) && wq_bytes < (1ll << 31), "ds_conv_fp8: operand larger than 2 GiB");
    p.x_bytes = (unsigned)(x_elems * (d->x_dtype == DS_DTYPE_BF16 ? 2 : 4));
    p.w_bytes = (unsigned)wq_bytes;
    const int nb = fp8_nb(d->Cout);
    p.row_tiles = (int)((fp8_M(d) + 127) / 128);
    p.col_tiles = (d->Cout + 32 * nb - 1) / (32 * nb);
    const dim3 grid((unsigned)(((int64_t)p.row_tiles * p.col_tiles + 7) / 8 * 8));
    hipStream_t st = (hipStream_t)stream;
    switch (nb) {
        case 1: launch_fp8<1>(d, a_format, grid, st, p); break;
        case 2: launch_fp8<2>(d, a_format, grid, st, p); break;
        case 3: launch_fp8<3>(d, a_format, grid, st, p); break;
        case 4: launch_fp8<4>(d, a_format, grid, st, p); break;
        case 5: launch_fp8<5>(d, a_format, grid, st, p); break;
        case 6: launch_fp8<6>(d, a_format, grid, st, p); break;
        case 7: launch_fp8<7>(d, a_format, grid, st, p); break;
        default: launch_fp8<8>(d, a_format, grid, st, p); break;
    }
    return ds::check_launch("ds_conv_fp8");
}
