// ds_conv_plan / ds_conv_prepare_weights / ds_conv_run: ONE conv-layer interface over the kernel families.
//
// A caller (the Python engine, or a C / C++ host) describes a slim.conv2d layer in TensorFlow's terms -- batch, map,
// filter [k][k][Cin][Cout] in HWIO, stride, SAME padding -- says whether it wants the forward conv or
// Conv2DBackpropInput and in which arithmetic, and gets back a plan: which kernel family this library runs for that
// shape (implicit GEMM / wide 1x1 / LDS-DMA, fused Winograd F(2x2) or F(4x4), the packed-RGB stem kernel, the
// register-direct bf16 / fp8 / f32x3 kernels), how many BatchNorm partials it writes, and which prepared form of the
// filter it reads.  The selection rules that used to live in the Python engine (isinstance chains over five plan
// classes) are here, next to the launch-time model they consult.  Host code only: no kernel in this file.
#include <stdlib.h>
#include <string.h>

#include "ds_kernels.h"

namespace ds {
void set_error(const char *fmt, ...);
}

#define PLAN_REQUIRE(cond, ...)         \
    do {                                \
        if (!(cond)) {                  \
            ds::set_error(__VA_ARGS__); \
            return DS_ERR_ARG;          \
        }                               \
    } while (0)

namespace {

// TF SAME geometry: out = ceil(n / s), the extra padding goes bottom / right
inline void same_pad(int n, int k, int s, int *out, int *before) {
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n;
    if (total < 0) total = 0;
    *before = total / 2;
}

inline bool is_wino(const ds_conv_layer_plan *p) { return p->family == DS_FAM_WINO2 || p->family == DS_FAM_WINO4 || p->family == DS_FAM_WINO4H; }

// partial count of the statistics epilogue of the chosen launch
int plan_partials(const ds_conv_layer_plan *p) {
    const ds_conv_desc &d = p->d;
    if (!(d.flags & (DS_EPI_STATS | DS_EPI_BNSUMS)) && p->family != DS_FAM_STEM && p->family != DS_FAM_STEM_POOL) return 0;
    switch (p->family) {
    case DS_FAM_WINO2: return ds_conv_wino_partials(d.N, d.H, d.W);
    case DS_FAM_WINO4:
        if (p->splitk > 1) return ds_conv_wino4_splitk_partials(d.N, d.H, d.W);      // (the reduce launch's row blocks)
        return ds_conv_wino4_partials(d.N, d.H, d.W);
    case DS_FAM_WINO4H: return ds_conv_wino4_partials(d.N, d.H, d.W);
    case DS_FAM_STEM: return d.dtype == DS_DTYPE_BF16 ? ds_conv_stem_bf16_partials(d.N, d.OH, d.OW) : ds_conv_stem_partials(d.N, d.OH, d.OW);
    case DS_FAM_STEM_POOL: return ds_conv_stem_pool_partials(d.N, d.OH, d.OW);
    case DS_FAM_BF16D: return ds_conv_bf16_partials(&d);
    case DS_FAM_FP8D: return ds_conv_fp8_partials(&d);
    case DS_FAM_F32X3: return ds_conv_f32x3_partials(&d);
    default: {
        ds_conv_desc t = d;
        t.partials = 0;
        return ds_conv_igemm_partials(&t);
    }
    }
}

}  // namespace

extern "C" int ds_conv_plan(ds_conv_layer_plan *out, int32_t role, int32_t arith, uint32_t options, int32_t N, int32_t H,
                            int32_t W, int32_t w_cin, int32_t w_cout, int32_t k, int32_t stride, int32_t ldx, int32_t ldz,
                            int32_t flags) {
    PLAN_REQUIRE(out != nullptr, "ds_conv_plan: null plan");
    PLAN_REQUIRE(role == DS_CONV_FWD || role == DS_CONV_DGRAD, "ds_conv_plan: role %d", role);
    PLAN_REQUIRE(arith >= DS_ARITH_F32 && arith <= DS_ARITH_F32X3, "ds_conv_plan: arithmetic %d", arith);
    PLAN_REQUIRE(N > 0 && H > 0 && W > 0 && w_cin > 0 && w_cout > 0 && k > 0 && stride > 0, "ds_conv_plan: geometry");
    PLAN_REQUIRE(role == DS_CONV_FWD || stride == 1, "ds_conv_plan: Conv2DBackpropInput is built for stride-1 SAME convs");
    const bool stem = (options & DS_PLAN_PACKED_RGB) != 0;      // Conv2d_1a_7x7: x is the packed [N, H, W, 3] batch
    PLAN_REQUIRE(!stem || (role == DS_CONV_FWD && k == 7 && (w_cin == 3 || w_cin == 4)),
                 "ds_conv_plan: DS_PLAN_PACKED_RGB is the 7x7 stem (filter stored with 3 or 4 input channels)");
    memset(out, 0, sizeof(*out));
    out->role = role;
    out->arith = arith;
    out->k = k;
    out->w_cin = w_cin;
    out->w_cout = w_cout;
    const bool dgrad = role == DS_CONV_DGRAD;
    const int cin = dgrad ? w_cout : w_cin;       // reduction channels of THIS launch
    const int cout = dgrad ? w_cin : w_cout;      // its output columns
    ds_conv_desc &d = out->d;
    d.N = N; d.H = H; d.W = W;
    d.stride = stride;
    d.Cout = cout;
    d.ldx = ldx; d.ldz = ldz;
    d.flags = flags;
    d.splits = 1;
    d.dtype = arith == DS_ARITH_F32 || arith == DS_ARITH_F32X3 ? DS_DTYPE_F32 : DS_DTYPE_BF16;
    if (stem) {
        // generic form of the stem: KW folded into the channel axis of a zero-padded 4-channel copy of the batch
        PLAN_REQUIRE(w_cin == 4, "ds_conv_plan: the stem filter is stored zero-padded to 4 input channels");
        d.Cin = 7 * 4; d.KH = 7; d.KW = 1; d.fold_cin = 4; d.ldx = 4;
        same_pad(H, 7, stride, &d.OH, &d.pad_t);
        same_pad(W, 7, stride, &d.OW, &d.pad_l);
        d.w_tap_stride = 28 * (int64_t)cout; d.w_n_stride = 1; d.w_k_stride = cout;
        out->alg_flops = 2.0 * N * d.OH * d.OW * cout * 147.0;
    } else {
        d.Cin = cin; d.KH = k; d.KW = k;
        same_pad(H, k, stride, &d.OH, &d.pad_t);
        same_pad(W, k, stride, &d.OW, &d.pad_l);
        d.w_tap_stride = (int64_t)w_cin * w_cout;
        d.w_n_stride = dgrad ? w_cout : 1;
        d.w_k_stride = dgrad ? 1 : w_cout;
        d.flip = dgrad ? 1 : 0;
        out->alg_flops = 2.0 * N * d.OH * d.OW * cout * (double)k * k * cin;
    }

    // ---- which kernel family -----------------------------------------------------------------------------------
    const bool f32 = arith == DS_ARITH_F32 || arith == DS_ARITH_F32X3;
    int fam = DS_FAM_IGEMM;
    if (stem) {
        // (the 16-bit configurations run the same kernel on the bf16 matrix cores: ds_conv_stem_bf16)
        if (cout == 64 && !(options & DS_PLAN_NO_STEM_DIRECT)) fam = DS_FAM_STEM;
        // ... with MaxPool_2a inside the kernel when the caller asks for it (a frozen stem whose only consumer is the pool)
        if (fam == DS_FAM_STEM && (options & DS_PLAN_STEM_POOL) && arith != DS_ARITH_F32X3 && stride == 2 &&
            ds_conv_stem_pool_supported(H, W))
            fam = DS_FAM_STEM_POOL;
    } else if (f32) {
        // 3x3 stride-1 layers: fused Winograd where it beats the implicit GEMM (profiles/r02_wino_layers.txt: every
        // 56x56 / 28x28 / 14x14 layer; on the 7x7 maps only the wide ones), F(4x4) where the launch-time model says so
        const bool wino = !(options & DS_PLAN_NO_WINO) && k == 3 && stride == 1 && cin % 8 == 0 && (H >= 14 || cout >= 128);
        if (wino)
            fam = (!(options & DS_PLAN_NO_WINO4) && ds_conv_wino4_prefer(N, H, W, cin, cout)) ? DS_FAM_WINO4 : DS_FAM_WINO2;
        else if (arith == DS_ARITH_F32X3 && !dgrad && k == 1 && stride == 1 && cin % 8 == 0 && cin <= 1024)
            fam = DS_FAM_F32X3;      // fp32 products from three bf16 pieces: the forward 1x1 convs (opt-in, own label)
    } else if (k == 1 || k == 3) {
        // ds_conv_fp8 on the layers where it beats the bf16 kernels (profiles/r04_fp8_layers_b128.txt, every conv shape of
        // the tower at cfg5's per-GPU batch): both kernels are fetch / conversion-bound, so fp8 is within +-5 % of bf16 on
        // most shapes, 3x SLOWER on the 16 / 24 / 32-channel 3x3 layers, 5-30 % slower into 64 columns (Conv2d_2c's dgrad:
        // 375 against 283 us) -- and 1.09-1.5x FASTER on the forward 3x3 layers with >= 96 input channels on 14 x 14 and
        // larger maps.  Only those run in fp8; the rest takes the bf16 rules.  In-box, ms/step at B = 256: bf16 13.02,
        // fp8 by this rule 13.07, by the wider rule (>= 64 channels into >= 96 columns, DS_PLAN_FP8_WIDE_RULE) 13.46, everywhere 13.8.
        const bool wins = (options & DS_PLAN_FP8_WIDE_RULE) ? (cin >= 64 && cout >= 96)      // A/B: the wider round-4 rule
                                                            : (!dgrad && k == 3 && cin >= 96 && H >= 14);      // the 1.09-1.5x layers only
        const bool fp8 = arith == DS_ARITH_FP8 && stride == 1 && cin % 8 == 0 && ((options & DS_PLAN_FP8_EVERYWHERE) || wins);
        if (fp8) {
            fam = DS_FAM_FP8D;
        } else if (!(options & DS_PLAN_NO_BF16_DIRECT)) {
            // register-direct bf16 where it beats the staged kernel (profiles/r02_bf16_layers.txt): forward from 48 output
            // columns up (always under 16-bit activation storage, which only it reads), dgrad for the 1x1 layers and
            // from 160 columns up
            if (!dgrad && cin % 8 == 0 && (cout >= 48 || (options & DS_PLAN_ACT16))) fam = DS_FAM_BF16D;
            if (dgrad && cin % 8 == 0 && (k == 1 || cout >= 160)) fam = DS_FAM_BF16D;
        }
        // the 3x3 input gradients (dz is fp32 in every configuration): F(4x4, 3x3) of the bf16-rounded operands on the bf16
        // matrix cores (ds_conv_wino4_bf16x2) against the LDS-staged and the register-direct bf16 kernels, us per launch at
        // B = 256 with the BatchNorm-sums epilogue (scripts/wino4h_dgrad_bench.py, profiles/r05_notes.md): all nineteen layers
        // 1756 against 2516 (direct), Conv2d_2c's 496 against 854; the two exceptions are the 14 x 14 layers with >= 288
        // reduction channels (direct: 130 / 141 against 144 / 155)
        if (!fp8 && dgrad && k == 3 && stride == 1 && !(options & DS_PLAN_NO_WINO4H) && ds_conv_wino4_supported(H, W, cin, cout)) {
            if (!(H == 14 && W == 14 && cin >= 288)) fam = DS_FAM_WINO4H;
            else if (!(options & DS_PLAN_NO_BF16_DIRECT) && cin % 8 == 0) fam = DS_FAM_BF16D;      // (the staged kernel: 141 / 155, no sums epilogue)
        }
        if (fam == DS_FAM_BF16D && !ds_conv_bf16_supported(&d)) fam = DS_FAM_IGEMM;
        if (fam == DS_FAM_FP8D && !ds_conv_fp8_supported(&d)) fam = DS_FAM_IGEMM;
    }
    out->family = fam;
    // small per-GPU batches: the reduction of a fused-Winograd launch split over several workgroups per output block
    out->splitk = 1;
    if (fam == DS_FAM_WINO4 && !(options & DS_PLAN_NO_SPLITK)) {
        out->splitk = ds_conv_wino4_splitk_choose(N, H, W, cin, cout);
        out->ws_bytes = (int64_t)ds_conv_wino4_splitk_workspace(N, H, W, cout, out->splitk);
    }
    out->a_format = dgrad ? DS_FP8_E5M2 : DS_FP8_E4M3;
    out->x16_ok = (fam == DS_FAM_BF16D || fam == DS_FAM_FP8D || fam == DS_FAM_WINO4H) ? 1 : 0;
    const int taps = k * k;
    switch (fam) {
    case DS_FAM_WINO2: out->w_bytes = 4LL * 16 * cin * cout; break;
    case DS_FAM_WINO4:
    case DS_FAM_WINO4H: out->w_bytes = 4LL * 36 * cin * cout; break;
    case DS_FAM_BF16D: out->w_bytes = (int64_t)ds_weights_bf16_bytes(w_cin, w_cout, taps, dgrad); break;
    case DS_FAM_FP8D:
        out->w_bytes = (int64_t)ds_weights_fp8_bytes(w_cin, w_cout, taps, dgrad);
        out->wscale_floats = 4 + DS_AMAX_FLOATS;
        break;
    case DS_FAM_F32X3: out->w_bytes = (int64_t)ds_weights_f32x3_bytes(w_cin, w_cout, taps, dgrad); break;
    default: out->w_bytes = 0;          // reads the HWIO filter in place
    }
    if (is_wino(out)) PLAN_REQUIRE(!(flags & ~(DS_EPI_STATS | DS_EPI_BNSUMS)), "ds_conv_plan: Winograd epilogues are STATS / BNSUMS");
    out->partials = plan_partials(out);
    if (fam == DS_FAM_IGEMM) d.partials = out->partials;
    return DS_OK;
}

extern "C" int ds_conv_plan_set_flags(ds_conv_layer_plan *p, int32_t flags) {
    PLAN_REQUIRE(p != nullptr, "ds_conv_plan_set_flags: null plan");
    p->d.flags = flags;
    p->d.partials = 0;
    p->partials = plan_partials(p);
    if (p->family == DS_FAM_IGEMM) p->d.partials = p->partials;
    return p->partials;
}

extern "C" int ds_conv_plan_enable_bnsums(ds_conv_layer_plan *p, int32_t ldy) {
    if (p == nullptr || p->role != DS_CONV_DGRAD) return 0;
    if (is_wino(p)) {
        p->d.flags = (p->d.flags & ~DS_EPI_STATS) | DS_EPI_BNSUMS;      // (the two sum epilogues exclude each other)
    } else if (p->family == DS_FAM_IGEMM && p->k == 1 && p->d.dtype == DS_DTYPE_F32) {
        ds_conv_desc t = p->d;
        t.partials = 0;
        if (!ds_conv_igemm_bnsums_supported(&t)) return 0;
        p->d.flags |= DS_EPI_BNSUMS;
        p->d.ldmask = ldy;
    } else if (p->family == DS_FAM_BF16D || p->family == DS_FAM_FP8D) {
        p->d.flags |= DS_EPI_BNSUMS;          // register-direct bf16 / fp8 dgrads carry the wide kernel's epilogue
        p->d.ldmask = ldy;
    } else {
        return 0;      // implicit-GEMM fallbacks of other shapes (LDS-staged kernels): the separate reduce pass stays
    }
    p->d.partials = 0;
    p->partials = plan_partials(p);
    if (p->family == DS_FAM_IGEMM) p->d.partials = p->partials;
    return p->partials;
}

extern "C" int ds_conv_plan_norm_supported(const ds_conv_layer_plan *p) {
    if (p == nullptr) return 0;
    if (p->family == DS_FAM_IGEMM) {
        ds_conv_desc t = p->d;
        t.partials = 0;
        return ds_conv_igemm_norm_supported(&t);
    }
    return p->family == DS_FAM_F32X3 && p->k == 1 && p->d.Cin <= 1024 ? 1 : 0;
}

extern "C" int ds_conv_plan_bnb_supported(const ds_conv_layer_plan *p) {
    if (p == nullptr || p->role != DS_CONV_DGRAD || p->family != DS_FAM_IGEMM || p->k != 1) return 0;
    ds_conv_desc t = p->d;
    t.partials = 0;
    return ds_conv_igemm_bnb_supported(&t);
}

extern "C" int ds_conv_plan_enable_pool3(ds_conv_layer_plan *p, uint8_t *argmax) {
    if (p == nullptr || argmax == nullptr || p->role != DS_CONV_FWD || p->family != DS_FAM_IGEMM || p->k != 1) return 0;
    ds_conv_desc t = p->d;
    t.partials = 0;
    if (!ds_conv_igemm_pool3_supported(&t)) return 0;
    p->d.pool_argmax = argmax;
    p->d.partials = 0;
    p->partials = plan_partials(p);
    p->d.partials = p->partials;
    return 1;
}

extern "C" int ds_conv_plan_finalize_tickets(const ds_conv_layer_plan *p) {
    if (p == nullptr || p->role != DS_CONV_FWD || p->family != DS_FAM_IGEMM) return 0;
    ds_conv_desc t = p->d;
    t.flags |= DS_EPI_STATS;
    return ds_conv_igemm_finalize_tickets(&t);
}

extern "C" int ds_conv_prepare_weights(const ds_conv_layer_plan *p, const float *w_hwio, void *w_prepared, float *wscale,
                                       void *stream) {
    PLAN_REQUIRE(p != nullptr, "ds_conv_prepare_weights: null plan");
    if (p->w_bytes == 0) return DS_OK;          // the family reads the HWIO tensor in place
    PLAN_REQUIRE(w_hwio && w_prepared, "ds_conv_prepare_weights: null filter");
    const int dgrad = p->role == DS_CONV_DGRAD, taps = p->k * p->k;
    switch (p->family) {
    case DS_FAM_WINO2: return ds_wino_transform_weights(w_hwio, (float *)w_prepared, p->w_cin, p->w_cout, dgrad, stream);
    case DS_FAM_WINO4: return ds_wino4_transform_weights(w_hwio, (float *)w_prepared, p->w_cin, p->w_cout, dgrad, stream);
    case DS_FAM_WINO4H: return ds_wino4_transform_weights_bf16x2(w_hwio, w_prepared, p->w_cin, p->w_cout, dgrad, stream);
    case DS_FAM_BF16D: return ds_weights_to_bf16(w_hwio, w_prepared, p->w_cin, p->w_cout, taps, dgrad, stream);
    case DS_FAM_F32X3: return ds_weights_to_f32x3(w_hwio, w_prepared, p->w_cin, p->w_cout, taps, dgrad, stream);
    case DS_FAM_FP8D:
        PLAN_REQUIRE(wscale, "ds_conv_prepare_weights: the fp8 filter needs its scale record");
        return ds_weights_to_fp8(w_hwio, w_prepared, wscale, p->w_cin, p->w_cout, taps, dgrad, stream);
    default: break;
    }
    PLAN_REQUIRE(false, "ds_conv_prepare_weights: family %d", p->family);
}

extern "C" int ds_conv_run(const ds_conv_layer_plan *p, const void *x, const void *w, float *z, const ds_conv_io *io,
                           void *stream) {
    PLAN_REQUIRE(p != nullptr, "ds_conv_run: null plan");
    static const ds_conv_io none = {};
    if (io == nullptr) io = &none;
    const ds_conv_desc &d = p->d;
    switch (p->family) {
    case DS_FAM_IGEMM:
        if (io->fin && (d.flags & DS_EPI_STATS)) {      // ds_bn_finalize inside the launch
            ds_conv_desc t = d;
            t.fin = io->fin;
            return ds_conv_igemm(&t, (const float *)x, (const float *)w, z, io->bias, io->mask, io->stats, io->pivot, stream);
        }
        return ds_conv_igemm(&d, (const float *)x, (const float *)w, z, io->bias, io->mask, io->stats, io->pivot, stream);
    case DS_FAM_WINO2:
        return ds_conv_wino((const float *)x, (const float *)w, z, io->stats, io->pivot, io->mask, d.N, d.H, d.W, d.Cin,
                            d.ldx, d.Cout, d.ldz, d.flags, stream);
    case DS_FAM_WINO4:
        if (p->splitk > 1) {
            PLAN_REQUIRE(io->ws && io->ws_bytes >= (size_t)p->ws_bytes, "ds_conv_run: this plan needs io.ws of %lld bytes", (long long)p->ws_bytes);
            return ds_conv_wino4_splitk((const float *)x, (const float *)w, z, io->stats, io->pivot, io->mask,
                                        (d.flags & DS_EPI_BNSUMS) ? d.mask_dtype : DS_DTYPE_F32, d.N, d.H, d.W, d.Cin, d.ldx, d.Cout,
                                        d.ldz, d.flags, p->splitk, io->ws, io->ws_bytes, stream);
        }
        return ds_conv_wino4((const float *)x, (const float *)w, z, io->stats, io->pivot, io->mask, d.N, d.H, d.W, d.Cin,
                             d.ldx, d.Cout, d.ldz, d.flags, stream);
    case DS_FAM_WINO4H:
        if (d.x_dtype == DS_DTYPE_BF16)
            return ds_conv_wino4_bf16x2_x16(x, w, z, io->stats, io->pivot, io->mask, (d.flags & DS_EPI_BNSUMS) ? d.mask_dtype : DS_DTYPE_F32,
                                            d.N, d.H, d.W, d.Cin, d.ldx, d.Cout, d.ldz, d.flags, stream);
        return ds_conv_wino4_bf16x2((const float *)x, w, z, io->stats, io->pivot, io->mask,
                                    (d.flags & DS_EPI_BNSUMS) ? d.mask_dtype : DS_DTYPE_F32, d.N, d.H, d.W, d.Cin, d.ldx, d.Cout, d.ldz,
                                    d.flags, stream);
    case DS_FAM_STEM:
        return (d.dtype == DS_DTYPE_BF16 ? ds_conv_stem_bf16 : ds_conv_stem)(
            (const float *)x, (const float *)w, z, (d.flags & DS_EPI_STATS) ? io->stats : nullptr, io->pivot, d.N, d.H, d.W, p->w_cin,
            d.Cout, d.ldz, stream);
    case DS_FAM_STEM_POOL:
        return (d.dtype == DS_DTYPE_BF16 ? ds_conv_stem_pool_bf16 : ds_conv_stem_pool)(
            (const float *)x, (const float *)w, z, (d.flags & DS_EPI_STATS) ? io->stats : nullptr, io->pivot, d.N, d.H, d.W, p->w_cin,
            d.Cout, d.ldz, stream);
    case DS_FAM_BF16D: return ds_conv_bf16(&d, x, w, z, io->mask, io->stats, io->pivot, stream);
    case DS_FAM_F32X3: return ds_conv_f32x3(&d, (const float *)x, w, z, io->mask, io->stats, io->pivot, stream);
    case DS_FAM_FP8D:
        return ds_conv_fp8(&d, x, io->x_amax, p->a_format, w, io->wscale, z, io->mask, io->stats, io->pivot, stream);
    default: break;
    }
    PLAN_REQUIRE(false, "ds_conv_run: family %d", p->family);
}
