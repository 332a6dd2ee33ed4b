/*
 * ds_kernels.h -- C ABI of libds_kernels.so, the MI355X (gfx950) kernel library behind the
 * Deep Sentiment training path.
 *
 * The reference (anthonyhu/tumblr-emotions) has no FFI/plugin layer: its hot path sits
 * directly behind Python calls into TensorFlow 1.x.  Each entry point below
 * therefore names the TensorFlow op call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors); the library
 *     never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and returns immediately;
 *   - return 0 on success, negative DS_ERR_* otherwise; ds_last_error() gives the text;
 *   - activations NHWC fp32, conv weights HWIO fp32 (TensorFlow layout, never re-packed),
 *     matrices row-major, ids/labels/seq_len int64 (as the reference's TFRecord schema,
 *     datasets/convert_to_dataset.py:148-161).
 */
#ifndef DS_KERNELS_H
#define DS_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_ERR_ARG (-1)
#define DS_ERR_LAUNCH (-2)
#define DS_ERR_WORKSPACE (-3)

/* arithmetic type of the multiply in ds_conv_igemm (ds_conv_desc.dtype); storage and accumulation stay fp32 */
#define DS_DTYPE_F32 0     /* v_mfma_f32_32x32x2_f32: exact fp32 (the 1e-3 parity path)                         */
#define DS_DTYPE_BF16 1    /* v_mfma_f32_32x32x16_bf16: operands rounded to bf16 on the way into LDS, fp32      */
                           /* accumulate, fp32 z / BatchNorm statistics / master weights (BASELINE configs[4]) */

/* epilogue flags of ds_conv_igemm */
#define DS_EPI_BIAS 1      /* z += bias[col]                       (BiasAdd)                        */
#define DS_EPI_RELU 2      /* z = max(z, 0)                        (tf.nn.relu)                     */
#define DS_EPI_ACCUM 4     /* z += previous contents of z          (AddN of two gradient paths)     */
#define DS_EPI_STATS 8     /* emit per-column sum / sum-of-squares partials for BatchNorm           */
#define DS_EPI_MASK 16     /* z *= (mask[row*ldmask+col] > 0)      (ReluGrad)                       */
#define DS_EPI_BNSUMS 32   /* Conv2DBackpropInput whose result dy feeds a BatchNorm+ReLU backward: emit the  */
                           /* per-column partials of sum(g) and sum(g*y), g = dy*(y > 0), y = mask[row*ldmask */
                           /* + col] = the forward activation relu(bn(z)) of the layer that consumes dy.  They */
                           /* replace that layer's ds_bn_bwd_reduce pass over z and dy (ds_bn_bwd_finalize_segs,*/
                           /* kind 1).  Layout float[2][Cout][P] like DS_EPI_STATS; excludes STATS and MASK.     */

int ds_version(void);
const char *ds_last_error(void);

/* Implicit-GEMM convolution / matrix multiply on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   z[m, co] = sum_{tap, ci} x[pixel(m) + tap, ci] * w[tap', ci|co ...]
 * One kernel serves
 *   - Conv2D forward, SAME padding      image_model/inception_v1.py:63-250 (slim.conv2d)
 *   - Conv2DBackpropInput (dgrad)       implied by create_train_op, im_text_rnn_model.py:135
 *   - MatMul (+BiasAdd, +Relu)          im_text_rnn_model.py:99-105, text_embedding.py:86,
 *                                       BasicLSTMCell's [x,h]*kernel, im_text_rnn_model.py:89-90
 * Weights are read in place from the TF HWIO tensor through (tap, n, k) strides; `flip`
 * reverses the tap order (dgrad).  A plain GEMM is the case N=M, H=W=OH=OW=KH=KW=1.        */
typedef struct ds_conv_desc {
    int32_t N, H, W;          /* input images and spatial size                                      */
    int32_t Cin;              /* reduction channels per tap                                         */
    int32_t ldx;              /* input pixel stride in floats (>= Cin; x points at channel offset)  */
    int32_t KH, KW, stride;
    int32_t pad_t, pad_l;     /* TF SAME "before" pads (extra goes bottom/right)                    */
    int32_t OH, OW;
    int32_t Cout;             /* output channels                                                    */
    int32_t ldz;              /* output pixel stride in floats                                      */
    int64_t w_tap_stride;     /* floats between consecutive taps in w                               */
    int32_t w_n_stride;       /* floats between consecutive output channels in w                    */
    int32_t w_k_stride;       /* floats between consecutive reduction channels in w (one of the two is 1) */
    int32_t flip;             /* 1: use tap (KH*KW-1-tap)                                           */
    int32_t fold_cin;         /* >0: KW is folded into Cin (stem conv): real channels per pixel     */
    int32_t flags;            /* DS_EPI_*                                                           */
    int32_t ldmask;           /* row stride of the DS_EPI_MASK source                               */
    int32_t splits;           /* 0/1: none.  >1: split-K for small-M GEMMs (LSTM steps): split s     */
                              /* writes its partial sum to z + s*z_split_stride; the consumer adds    */
                              /* the slabs (flags must be 0)                                          */
    int64_t z_split_stride;   /* floats between output slabs                                          */
    int32_t tile_nt;          /* 0: automatic.  1..6: workgroup tile is 128 rows x 32*tile_nt columns  */
    int32_t grid_x;           /* 0: automatic.  >0: persistent workgroups per column tile (each walks  */
                              /* row tiles blockIdx.x, +grid_x, ...); also the stats partial count P   */
    int32_t dtype;            /* DS_DTYPE_F32 (0, default) or DS_DTYPE_BF16                             */
    int32_t x_dtype;          /* storage type of x for ds_conv_bf16 / ds_conv_fp8: DS_DTYPE_F32 (0) or            */
                              /* DS_DTYPE_BF16 (16-bit activation storage: ldx counts bf16 elements)              */
    int32_t partials;         /* 0: unchecked.  >0 with DS_EPI_STATS: the partial count P the caller sized and  */
                              /* finalises with (ds_conv_igemm_partials at plan time); a launch that would     */
                              /* write a different count fails with DS_ERR_ARG instead of corrupting the stats */
    /* BatchNorm + ReLU applied ON LOAD (wide 1x1 kernel only, ds_conv_igemm_norm_supported; all nullable):          */
    const float *norm_rstd;   /* x holds pre-BatchNorm conv outputs z: the kernel uses relu(x*norm_rstd[k] +          */
    const float *norm_shift;  /* norm_shift[k]) per reduction channel k (Cin floats each, Cin <= 1024).  Channels      */
                              /* that are already activations carry (1, 0): relu(y) = y.                               */
    const float *mask_rstd;   /* DS_EPI_BNSUMS: `mask` holds z of the consumer layer(s) instead of y; y is rebuilt as */
    const float *mask_shift;  /* relu(mask*mask_rstd[n] + mask_shift[n]) per output column n (Cout floats each)        */
    const struct ds_bn_bwd_on_load *bnb;   /* HOST pointer, nullable: BatchNorm + ReLU backward applied ON LOAD (below)  */
    int32_t mask_dtype;       /* DS_EPI_BNSUMS in ds_conv_bf16 / ds_conv_fp8: storage type of `mask` (DS_DTYPE_F32 or, under     */
                              /* 16-bit activation storage, DS_DTYPE_BF16: ldmask then counts bf16 elements)                    */
    int32_t z_dtype;          /* ds_conv_bf16 without accumulate / sums epilogue: DS_DTYPE_BF16 stores z - pivot ROUNDED to bf16 (RNE; pixel stride */
                              /* ldz in elements; pivot = the DS_EPI_STATS pivot, 0 without one) -- the 16-bit configurations'  */
                              /* frozen layers, whose BatchNorm passes then move 2 instead of 4 bytes per element of z.  The    */
                              /* statistics come from the fp32 accumulators; centring keeps xhat = (z - mean) rstd free of      */
                              /* cancellation (ds_bn_finalize_centered gives the consumers mean - pivot and the matching shift) */
    /* MaxPool 3x3 / 1 SAME applied ON LOAD (wide 1x1 kernel only, ds_conv_igemm_pool3_supported; nullable): an Inception   */
    /* block's Branch_3 = slim.max_pool2d(net, [3, 3], stride 1) -> slim.conv2d(., [1, 1]) (image_model/inception_v1.py:94-95  */
    /* ... :246-247) as ONE launch.  The reduction operand of pixel (h, w) is the maximum of x over its 3x3 neighbourhood     */
    /* (padded cells never win), formed as the loader reads the three rows; with norm_rstd / norm_shift the maximum is taken  */
    /* over the pre-BatchNorm values and normalised afterwards (rstd > 0: the same value).  pool_argmax [N*H*W][Cin] bytes    */
    /* receives the row-major-first winner of every window (kh * 3 + kw, what ds_maxpool_fwd records and ds_maxpool_bwd       */
    /* reads); the pooled tensor itself never reaches memory.  z is bit-identical to ds_maxpool_fwd + the plain launch; the   */
    /* statistics partials group other rows (32-pixel blocks of whole image rows), so their sums differ in rounding.          */
    uint8_t *pool_argmax;
    /* ds_bn_finalize INSIDE the conv launch (HOST pointer, nullable; DS_EPI_STATS launches of the wide 1x1 kernel with at most  */
    /* 256 partials per channel, ds_conv_igemm_finalize_supported): see ds_bn_finalize_in_launch below.                       */
    const struct ds_bn_finalize_in_launch *fin;
} ds_conv_desc;

/* slim.batch_norm's statistics epilogue without its own launch.  With small per-GPU batches a conv launch is one partial round
 * of workgroups and the ds_bn_finalize launch behind it costs a dependent-launch boundary (~3 us) for ~1 us of work, 39 times
 * per forward pass (BASELINE configs[3]: 32 samples per GPU).  With `fin` set, every workgroup publishes its partial
 * write-through and takes a ticket of its COLUMN TILE (one relaxed agent-scope increment, no fence); the last arriver of a
 * column tile reads that tile's P partials back and writes mean / rstd / shift (+ the moving averages) of its columns with
 * ds_bn_finalize's own summation tree: bit-identical to the separate launch.  ticket: one zero-initialised uint32 per column
 * tile (ds_conv_igemm_finalize_tickets), reset by the last arriver, NOT shared by launches that can run concurrently.     */
typedef struct ds_bn_finalize_in_launch {
    const float *beta;              /* Cout floats                                                                          */
    float *mean, *rstd, *shift;     /* out, Cout floats each; `mean` may be the launch's pivot                                */
    float *moving_mean, *moving_var; /* nullable: assign_moving_average with `decay`                                          */
    uint32_t *ticket;
    int64_t count;                  /* elements per channel (N * OH * OW; sync_bn callers finalize themselves)               */
    float eps, decay;
} ds_bn_finalize_in_launch;

/* Conv2DBackpropInput of a 1x1 conv + BatchNorm + ReLU layer WITHOUT the separate ds_bn_bwd_apply pass (wide 1x1 kernel
 * only, ds_conv_igemm_bnb_supported): x holds the layer's pre-BatchNorm output z (pixel stride ldx, Cin = the layer's
 * output channels), and the reduction operand is formed as it is loaded,
 *     dz = rstd (g - coef[0] - xhat coef[1]),  g = dy (z rstd + shift > 0),  xhat = (z - mean) rstd
 * -- slim.batch_norm's gradient (scale = False; slim/nets/inception_utils.py:48-70) with coef = the two column means that
 * ds_bn_bwd_finalize* produce; bit-identical to ds_bn_bwd_apply followed by the plain dgrad.  dy may live in up to three
 * channel ranges (the slices of a fused 1x1 layer's output gradient): ascending, boundaries multiples of 16.  z is left
 * untouched (a weight gradient that needs dz keeps the separate pass).                                               */
typedef struct ds_bn_bwd_on_load {
    const float *mean, *rstd, *shift;   /* Cin floats each                                                        */
    const float *coef;                  /* float[2][Cin]                                                          */
    int32_t nseg;                       /* 1..3                                                                   */
    int32_t c_end[3];                   /* range i covers channels [c_end[i-1], c_end[i]); c_end[nseg-1] == Cin   */
    int32_t ld[3];                      /* pixel stride of each range's dy (floats)                               */
    const float *dy[3];                 /* channel c_end[i-1] of pixel 0 of range i                               */
} ds_bn_bwd_on_load;

#ifdef DS_TUNING
/* DEBUG / A-B AIDS (ds_debug_*): process-global switches for tests and tuning scripts.  They exist in the TUNING build only
 * (-DDS_TUNING: libds_kernels_tuning.so, which also honours the DS_* environment knobs of the selection rules); the shipped
 * libds_kernels.so exports none of them, reads no environment variable and keeps no mutable global state besides its
 * init-once device caches (tests/test_abi_cpu.py checks its symbol table).  They are NOT re-entrant and none is needed for
 * correct results; per-layer choices go through ds_conv_desc.tile_nt / grid_x instead.
 * Pin the workgroup tile to (128*mt) x (32*nt) rows x columns; 0,0 = automatic.  */
int ds_debug_conv_set_tile(int mt, int nt);
/* 0 = automatic, 1 = register-staged LDS kernel (K-tile 16), 2 = register-direct (LDS-free)
 * kernel, 3 = LDS-DMA kernel (buffer_load ... lds, K-tile 32; falls back to 1 where it does not apply). */
int ds_debug_conv_set_path(int path);
/* the wide-tile register-direct kernel for plain 1x1 / GEMM shapes (flags within DS_EPI_STATS):
 * 0 = never, 1 = automatic (default), 2 = wherever the shape allows.                                      */
int ds_debug_conv_set_wide(int mode);
/* ds_conv_wino's kernel carries four ablation bits in `flags` (256 / 512 / 1024 / 2048: skip the pixel loads, the weight
 * DMAs, the output stores ... -- garbage results, timing only).  They are refused (DS_ERR_ARG) unless switched on here. */
int ds_debug_conv_wino_allow_ablation(int on);
/* ds_conv_wino4: pin the 32-channel blocks per workgroup (1, 2; 0 = the launch-time model, the default).  Both choices give
 * the same z up to summation order; the statistics partial count does not depend on it.                                  */
int ds_debug_conv_wino4_set_nb(int nb);
/* tuning: the most column blocks (of 32) a wave of ds_conv_bf16 may own (1 .. 8; default 8) */
int ds_debug_conv_bf16_set_max_nb(int nb);
#endif /* DS_TUNING */
/* Number of row-tile blocks (P) the launch for `d` will use = number of stats partials.   */
int ds_conv_igemm_partials(const ds_conv_desc *d);
/* 1 if a launch for `d` can carry DS_EPI_BNSUMS (plain 1x1 stride-1 shapes that run on the wide kernel).  */
int ds_conv_igemm_bnsums_supported(const ds_conv_desc *d);
/* ... and whether ds_conv_igemm would apply norm_rstd / norm_shift for this descriptor (same kernel, same rule).   */
int ds_conv_igemm_norm_supported(const ds_conv_desc *d);
/* ... and whether it takes ds_conv_desc.bnb (k-contiguous weights, i.e. a dgrad, on the wide kernel; Cin <= 1024).     */
int ds_conv_igemm_bnb_supported(const ds_conv_desc *d);
/* ... and whether it would form the 3x3 / 1 max pool of x on load for this shape (ds_conv_desc.pool_argmax: forward 1x1,
 * n-contiguous weights, W <= 32, Cin % 16 == 0, Cout <= 128 -- one column tile, so the pooling is done once).  Judged on
 * the shape alone: d->pool_argmax may still be null.                                                                  */
int ds_conv_igemm_pool3_supported(const ds_conv_desc *d);
/* ... and whether a DS_EPI_STATS launch for `d` would honour ds_conv_desc.fin (wide kernel, at most 256 partials); returns the
 * number of ticket words the launch needs (its column tiles), 0 = not supported (the caller keeps ds_bn_finalize).        */
int ds_conv_igemm_finalize_tickets(const ds_conv_desc *d);
/* stats (DS_EPI_STATS): float[2][Cout][P] partial column sums of (z - pivot) and (z - pivot)^2.
 * pivot (nullable = 0): float[Cout], any value near the column mean -- the build passes the previous step's
 * batch mean (the moving mean after a restore) -- so that the fp32 partial sums carry the spread of z rather
 * than its offset and the variance E[u^2] - E[u]^2 does not cancel for channels with |mean| >> std
 * (tf.nn.moments, which slim.batch_norm uses, takes the mean of squared differences).        */
int ds_conv_igemm(const ds_conv_desc *d, const float *x, const float *w, float *z, const float *bias,
                  const float *mask, float *stats, const float *pivot, void *stream);

/* DS_DTYPE_BF16, second generation: register-direct implicit GEMM on v_mfma_f32_32x32x16_bf16 for the 1x1 and 3x3
 * convs (forward and Conv2DBackpropInput), fp32 accumulation and storage.  wb = the filter converted once per
 * weight update by ds_weights_to_bf16 into the kernel's own order ([channel chunk x tap][column][16 k] bf16, zero
 * padded; ds_weights_bf16_bytes gives its size; dgrad = 1: flipped taps, channel roles swapped -- the conv then
 * runs over dz with Cin = Cout_w, Cout = Cin_w).  `d` as for ds_conv_igemm (geometry, ldx, ldz; the weight strides are
 * ignored); flags DS_EPI_STATS (forward), or -- a dgrad -- DS_EPI_ACCUM and DS_EPI_BNSUMS exactly as ds_conv_igemm's wide
 * kernel, with `mask` = the consumer layer's activation in fp32 or bf16 storage (d.mask_dtype, ldmask);
 * partials float[2][Cout][ds_conv_bf16_partials(d)].                                                          */
size_t ds_weights_bf16_bytes(int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad);
int ds_weights_to_bf16(const float *w, void *wb, int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad, void *stream);
int ds_conv_bf16_supported(const ds_conv_desc *d);
int ds_conv_bf16_partials(const ds_conv_desc *d);
int ds_conv_bf16(const ds_conv_desc *d, const void *x, const void *wb, float *z, const void *mask, float *stats,
                 const float *pivot, void *stream);

/* fp32 PRODUCTS on the bf16 matrix cores ("bf16x3").  Every fp32 operand is split into three bf16 pieces (8 + 8 + 8
 * mantissa bits) and a*b is accumulated in fp32 from the six piece products whose magnitude exceeds 2^-24 |ab| -- six
 * v_mfma_f32_32x32x16_bf16 instead of eight v_mfma_f32_32x32x2_f32 per 16 reduction channels, 2.67x fewer matrix
 * cycles.  Error against fp64 equals the fp32 MFMA's (scripts/microbench/mfma_x3.hip: 3.26e-7 vs 3.22e-7 relative rms); results
 * are NOT bit-identical to ds_conv_igemm.  Same descriptor as ds_conv_bf16 (1x1 / 3x3, Cin % 8 == 0, fp32 x); flags
 * DS_EPI_STATS, or -- for a dgrad -- DS_EPI_ACCUM and DS_EPI_BNSUMS (mask, ldmask, mask_rstd / mask_shift) exactly as
 * ds_conv_igemm's wide kernel; norm_rstd / norm_shift (1x1) as for ds_conv_igemm.  wb = ds_weights_to_f32x3(w): the weights'
 * three pieces in the kernel's K-loop order, ds_weights_f32x3_bytes bytes, remade whenever w changes.               */
size_t ds_weights_f32x3_bytes(int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad);
int ds_weights_to_f32x3(const float *w, void *wb, int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad, void *stream);
int ds_conv_f32x3_supported(const ds_conv_desc *d);
int ds_conv_f32x3_partials(const ds_conv_desc *d);
int ds_conv_f32x3(const ds_conv_desc *d, const float *x, const void *wb, float *z, const float *mask, float *stats,
                  const float *pivot, void *stream);

/* fp8 convolution path (BASELINE configs[4]: fp8 MFMA conv path on CDNA4), 1x1 and 3x3 convs, forward and
 * Conv2DBackpropInput (slim.conv2d, image_model/inception_v1.py:71-250): v_mfma_f32_32x32x16_fp8_fp8 / _bf8_fp8, OCP
 * formats (gfx950), fp32 accumulation, fp32 z.  Per-tensor POWER-OF-TWO scales s = 2^floor(log2(FMAX / amax)):
 *   weights      e4m3 (FMAX 448): ds_weights_to_fp8 takes amax, fixes s_w and converts the TF HWIO filter into the
 *                kernel's order [16-channel chunk x tap][column][16 k] (ds_weights_fp8_bytes; dgrad = 1: flipped taps,
 *                channel roles swapped); wscale = device float[4 + DS_AMAX_FLOATS]: {amax, s_w, 1 / s_w, -} written by
 *                the call, followed by scratch for the filter's amax record;
 *   activations  a_format DS_FP8_E4M3 (forward x) or DS_FP8_E5M2 (FMAX 57344: dgrad's dz); the scale is derived in the
 *                kernel from the device record x_amax (max |x|: ds_absmax, or a producer that tracks it): nothing
 *                crosses to the host.  Values are scaled, saturated to +-FMAX and rounded to nearest even.
 * z = acc / (s_a s_w); `d`, flags and `mask` as for ds_conv_bf16 (partials float[2][Cout][ds_conv_fp8_partials]).
 * Not the fp32 parity path: separately labelled, tolerance documented in tests/test_kernels_gpu.py / DESIGN.md.     */
#define DS_FP8_E4M3 0
#define DS_FP8_E5M2 1
/* A max|.| record ("amax", "x_amax" below and in ds_segments / ds_bn_bwd_apply / ds_maxpool_bn_relu_fwd) is
 * DS_AMAX_FLOATS device floats: 16 slots, one per 128-byte line (float 0, 32, 64, ...), that the producing kernels
 * raise by atomic max -- spread so that thousands of waves do not queue on one word; its value is the maximum of the 16
 * slots.  The owner zeroes a record before its first producer of a step (ds_absmax zeroes its own).            */
#define DS_AMAX_FLOATS 512
int ds_absmax(const void *x, int64_t n, int32_t x_dtype, float *amax, void *stream);
size_t ds_weights_fp8_bytes(int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad);
int ds_weights_to_fp8(const float *w, void *wq, float *wscale, int32_t Cin, int32_t Cout, int32_t taps, int32_t dgrad,
                      void *stream);
int ds_conv_fp8_supported(const ds_conv_desc *d);
int ds_conv_fp8_partials(const ds_conv_desc *d);
int ds_conv_fp8(const ds_conv_desc *d, const void *x, const float *x_amax, int32_t a_format, const void *wq,
                const float *wscale, float *z, const void *mask, float *stats, const float *pivot, void *stream);

/* Conv2d_1a_7x7 (inception_v1.py:63): 7x7 stride-2 SAME conv 3 -> 64 read from the PACKED RGB images
 * x [N, H, W, 3] (no 4-channel copy), w = HWIO [7][7][cin_store][64] (cin_store 3 or 4: the store keeps the stem
 * filter zero-padded to 4 input channels), z [N*OH*OW, ldz].  stats != NULL: BatchNorm column sums about `pivot`
 * as float[2][64][ds_conv_stem_partials(N, OH, OW)].                                                         */
int ds_conv_stem_partials(int32_t N, int32_t OH, int32_t OW);
int ds_conv_stem(const float *x, const float *w, float *z, float *stats, const float *pivot, int32_t N, int32_t H,
                 int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream);
/* Conv2d_1a_7x7 -> [BatchNorm -> ReLU ->] MaxPool_2a_3x3 (inception_v1.py:63-67) in ONE launch: the 3x3 / 2 SAME max pool is
 * taken inside the conv kernel (it commutes with relu(rstd * . + shift), rstd > 0), so the full-resolution conv output is
 * never written: zmax [N, OH/2, OW/2, 64] (pixel stride ldz, 16-byte aligned) = the window maxima of z; stats = the column
 * sums of the FULL conv map about `pivot`, float[2][64][ds_conv_stem_pool_partials(N, OH, OW)].  The pooled activation is
 * relu(rstd * zmax + shift); BatchNorm's backward sums of a frozen stem follow from zmax and the pooled gradient alone.
 * zmax holds exactly the maxima of ds_conv_stem's z (same MFMA sequence); the statistics group other pixels per partial.
 * Conv maps with even sizes and 16 .. 112 columns (ds_conv_stem_pool_supported(H, W) of the INPUT size).               */
int ds_conv_stem_pool_supported(int32_t H, int32_t W);
int ds_conv_stem_pool_partials(int32_t N, int32_t OH, int32_t OW);
int ds_conv_stem_pool(const float *x, const float *w, float *zmax, float *stats, const float *pivot, int32_t N, int32_t H,
                      int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream);
/* ... of ds_conv_stem_bf16 (the 16-bit configurations: operands rounded to bf16, bf16 MFMA; zmax and the sums stay fp32) */
int ds_conv_stem_pool_bf16(const float *x, const float *w, float *zmax, float *stats, const float *pivot, int32_t N, int32_t H,
                           int32_t W, int32_t cin_store, int32_t Cout, int32_t ldz, void *stream);
/* ds_conv_stem for the 16-bit configurations: x and w rounded to bf16 (RNE) as they are packed, v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (two kernel rows per three MFMAs: 11 instead of 84 per accumulator); same arguments and layout; stats as
 * float[2][64][ds_conv_stem_bf16_partials(N, OH, OW)]. */
int ds_conv_stem_bf16_partials(int32_t N, int32_t OH, int32_t OW);      /* its own partial count (one per workgroup) */
int ds_conv_stem_bf16(const float *x, const float *w, float *z, float *stats, const float *pivot, int32_t N, int32_t H, int32_t W,
                      int32_t cin_store, int32_t Cout, int32_t ldz, void *stream);

/* 3x3 stride-1 SAME convolution as fused Winograd F(2x2, 3x3) on fp32 MFMA: 2.25x fewer matrix passes than the
 * implicit GEMM for Conv2d_2c_3x3 and the Branch_1 / Branch_2 Conv2d_0b_3x3 of every Mixed block
 * (image_model/inception_v1.py:74-75, 86-247), forward and Conv2DBackpropInput alike.
 *   u      = G g G^T of the filter, [16][Cout][Cin] floats, made by ds_wino_transform_weights from the TF HWIO
 *            tensor w [3][3][Cin_w][Cout_w] whenever w changes.  dgrad = 0: forward (Cin = Cin_w, Cout = Cout_w);
 *            dgrad = 1: input gradient -- flipped taps, channel roles swapped (the conv then runs over dz with
 *            Cin = Cout_w, Cout = Cin_w).  ds_wino_transform_weights takes (Cin_w, Cout_w) either way.
 *   x, z   NHWC with pixel strides ldx / ldz; Cin % 8 == 0; flags: 0, DS_EPI_STATS (partials float[2][Cout][P],
 *          P = ds_conv_wino_partials, about `pivot` like ds_conv_igemm) or DS_EPI_BNSUMS (dgrad: partials of sum g,
 *          sum g*y with y = ymask[pixel*ldz + channel], the consumer layer's forward activation; same layout).
 * Results equal the direct kernels' to ~1e-6 relative (the transforms round differently), deterministically.   */
int ds_wino_transform_weights(const float *w, float *u, int32_t Cin, int32_t Cout, int32_t dgrad, void *stream);
int ds_conv_wino_partials(int32_t N, int32_t H, int32_t W);
int ds_conv_wino(const float *x, const float *u, float *z, float *stats, const float *pivot, const float *ymask,
                 int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz, int32_t flags,
                 void *stream);

/* The same convolution as fused Winograd F(4x4, 3x3): 4x fewer matrix passes than the implicit GEMM, 1.78x fewer than
 * ds_conv_wino; 1.3-1.6x faster than ds_conv_wino on the 56 x 56 / 28 x 28 layers (Conv2d_2c_3x3, Mixed_3b / 3c:
 * image_model/inception_v1.py:74-75, 86-115) and on those 14 x 14 / 7 x 7 layers whose workgroup count fills the chip
 * better this way.  Same arguments and flags as ds_conv_wino except
 *   u      = G g G^T (6 x 6) as float[36][Cin / 8][Cout][8] from ds_wino4_transform_weights (same dgrad convention);
 *   stats  partials float[2][Cout][P] with P = ds_conv_wino4_partials (one per 32 tiles of 4 x 4 outputs);
 *   needs  Cin % 16 == 0, even ldx, 8-byte aligned x (ds_conv_wino4_supported); any H, W (partial border tiles).
 * ds_conv_wino4_prefer: the launch-time model's advice for a shape -- 0: ds_conv_wino is expected to be faster, else
 * nonzero.  Results equal the direct kernels' to ~1e-5 relative (F(4x4) costs about one decimal digit against
 * F(2x2): rms 2.4e-6 vs 3.7e-7), deterministically.   */
int ds_conv_wino4_supported(int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int ds_conv_wino4_prefer(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int ds_wino4_transform_weights(const float *w, float *u, int32_t Cin, int32_t Cout, int32_t dgrad, void *stream);
int ds_conv_wino4_partials(int32_t N, int32_t H, int32_t W);
int ds_conv_wino4(const float *x, const float *u, float *z, float *stats, const float *pivot, const float *ymask,
                  int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz, int32_t flags,
                  void *stream);
/* The same with the reduction channels SPLIT over `splits` workgroups per output block (small per-GPU batches: with fewer
 * workgroups than CUs the launch lasts as long as ONE workgroup's Cin / 16 K steps -- 68 us for the 7x7 320 -> 160 input
 * gradient at 256 samples and at 32 alike).  Two launches: the conv kernel writes the slices' partial outputs into ws
 * (dense [splits][N H W][Cout], ds_conv_wino4_splitk_workspace bytes), a reduce launch adds them in slice order, writes z
 * (pixel stride ldz) and runs the DS_EPI_STATS / DS_EPI_BNSUMS epilogue on the sums (partials float[2][Cout][P] with
 * P = ds_conv_wino4_splitk_partials; y_dtype: storage of the BNSUMS activation).  ds_conv_wino4_splitk_choose: the slice count
 * the launch-time model picks (1 = use ds_conv_wino4).  z differs from ds_conv_wino4's by fp32 summation order only.   */
int ds_conv_wino4_splitk_choose(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int ds_conv_wino4_splitk_partials(int32_t N, int32_t H, int32_t W);
size_t ds_conv_wino4_splitk_workspace(int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t splits);
int ds_conv_wino4_splitk(const float *x, const float *u, float *z, float *stats, const float *pivot, const float *ymask,
                         int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz,
                         int32_t flags, int32_t splits, void *ws, size_t ws_bytes, void *stream);
/* ds_conv_wino4 for the 16-bit configurations (bf16 / fp8 labels): the convolution of the bf16-ROUNDED operands (x is rounded
 * as it is loaded, the filter before G g G^T) evaluated through F(4x4, 3x3) on the bf16 matrix cores, every Winograd-domain
 * value carried as two bf16 pieces (three v_mfma_f32_32x32x16_bf16 per product: ~2^-16 relative in the transform domain).
 * Same geometry, epilogues, partial count and NB pin as ds_conv_wino4; u2 (36 * Cin * Cout * 4 bytes, 16-byte aligned) from
 * ds_wino4_transform_weights_bf16x2 (reduction channels % 16 == 0).  Replaces the reference's tf.nn.conv2d 3x3 / its input
 * gradient (slim/nets inception_v1.py:74-75, :86-247) under the cfg5 label. */
int ds_wino4_transform_weights_bf16x2(const float *w, void *u2, int32_t Cin, int32_t Cout, int32_t dgrad, void *stream);
int ds_conv_wino4_bf16x2(const float *x, const void *u2, float *z, float *stats, const float *pivot, const void *ymask,
                         int32_t y_dtype /* of ymask: DS_DTYPE_F32 / DS_DTYPE_BF16 (16-bit activation storage) */, int32_t N,
                         int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz, int32_t flags, void *stream);
/* ... with x in 16-bit storage (ldx in bf16 elements; the bf16 dz of ds_bn_bwd_apply_bf16): same bits, half the pixel bytes */
int ds_conv_wino4_bf16x2_x16(const void *x16, const void *u2, float *z, float *stats, const float *pivot, const void *ymask,
                             int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Cout, int32_t ldz,
                             int32_t flags, void *stream);

/* ---- ONE conv-layer interface over the kernel families above (the product path's conv entry points) ---------------
 * slim.conv2d (image_model/inception_v1.py:63-250) and its Conv2DBackpropInput, described in TensorFlow's terms; the
 * LIBRARY picks the kernel family for the shape -- implicit GEMM / wide 1x1 / LDS-DMA (ds_conv_igemm), fused Winograd
 * F(2x2) / F(4x4) by the launch-time model (ds_conv_wino, ds_conv_wino4), the packed-RGB stem kernel (ds_conv_stem), the
 * register-direct bf16 / fp8 / f32x3 kernels -- so a C, C++ or Python host gets the measured configuration from three
 * calls.  The family-level entry points above stay exported for kernel tests and tuning scripts; the engine of this
 * build calls only these (tests/test_abi_cpu.py checks that).
 *   ds_conv_plan             fills a caller-owned plan (plain data, no allocation): role DS_CONV_FWD or DS_CONV_DGRAD
 *                            (stride-1 SAME convs), arithmetic DS_ARITH_*, N / H / W of the layer INPUT, the filter's own
 *                            channel counts and size k, pixel strides of x and z, epilogue flags (DS_EPI_*).
 *   ds_conv_plan_set_flags   changes the epilogue flags of a plan, returns its new partial count
 *   ds_conv_plan_enable_bnsums   dgrad whose result feeds a BatchNorm + ReLU backward: adds DS_EPI_BNSUMS (y with pixel
 *                            stride ldy goes in as io.mask) and returns the partial count, or 0 when the chosen kernel
 *                            cannot carry it (the caller then keeps the separate ds_bn_bwd_reduce pass)
 *   ds_conv_plan_norm_supported   would the launch apply d.norm_rstd / d.norm_shift (BatchNorm + ReLU on load)?
 *   ds_conv_plan_bnb_supported    dgrad: would it apply the layer's BatchNorm backward on load (d.bnb)?
 *   ds_conv_plan_enable_pool3     forward 1x1 conv behind a 3x3 / 1 SAME max pool (an Inception block's Branch_3): the pool is
 *                            formed on load (d.pool_argmax = the winners' bytes, what ds_maxpool_fwd would record) and
 *                            plan.partials is recomputed; returns 1, or 0 when the chosen kernel cannot (the plan is unchanged
 *                            and the caller keeps the separate pool pass)
 *   ds_conv_prepare_weights  plan.w_bytes > 0: converts the HWIO filter into the form the family reads (G g G^T, bf16 /
 *                            fp8 / three-piece K-loop order); redo whenever the filter changes.  fp8: wscale =
 *                            float[plan.wscale_floats]
 *   ds_conv_run              x, w (the HWIO filter when plan.w_bytes == 0, else the prepared one), z, the optional
 *                            epilogue operands in `io` (NULL = none).  Between runs the caller may change
 *                            plan.d.ldx / ldz / flags (within the planned epilogue set) / x_dtype / norm_* / mask_*.   */
#define DS_CONV_FWD 0
#define DS_CONV_DGRAD 1
#define DS_ARITH_F32 0       /* exact fp32 MFMA, incl. fp32 Winograd: the 1e-3 parity path                              */
#define DS_ARITH_BF16 1      /* bf16 multiplies, fp32 accumulation / storage of z                                       */
#define DS_ARITH_FP8 2       /* e4m3 / e5m2 multiplies with per-tensor power-of-two scales where they beat bf16, bf16 elsewhere */
#define DS_ARITH_F32X3 3     /* fp32 arithmetic, the forward 1x1 convs with fp32 products from three bf16 pieces         */
#define DS_FAM_IGEMM 0
#define DS_FAM_WINO2 1
#define DS_FAM_WINO4 2
#define DS_FAM_STEM 3
#define DS_FAM_BF16D 4
#define DS_FAM_FP8D 5
#define DS_FAM_F32X3 6
#define DS_FAM_WINO4H 7        /* ds_conv_wino4_bf16x2: F(4x4, 3x3) of the bf16-rounded operands on the bf16 matrix cores */
#define DS_FAM_STEM_POOL 8     /* ds_conv_stem_pool: the stem with MaxPool_2a inside (z of ds_conv_run = the pooled maxima)  */
#define DS_PLAN_NO_WINO 1u          /* A/B: implicit GEMM for every 3x3 layer                                           */
#define DS_PLAN_NO_WINO4 2u         /* A/B: F(2x2) wherever Winograd applies                                            */
#define DS_PLAN_NO_STEM_DIRECT 4u   /* A/B: the stem through the generic kernel on a 4-channel copy of the batch        */
#define DS_PLAN_NO_BF16_DIRECT 8u   /* A/B: the LDS-staged bf16 kernel everywhere                                       */
#define DS_PLAN_ACT16 16u           /* the net keeps activations in 16-bit storage (only the register-direct kernels read it) */
#define DS_PLAN_FP8_EVERYWHERE 64u  /* A/B: ds_conv_fp8 wherever it applies (default: only where it beats the bf16 kernels) */
#define DS_PLAN_FP8_WIDE_RULE 128u  /* A/B: fp8 for every 1x1 / 3x3 layer with >= 64 reduction channels into >= 96 columns  */
#define DS_PLAN_NO_WINO4H 256u      /* A/B: the 16-bit configurations' 3x3 input gradients on the direct bf16 kernels only  */
#define DS_PLAN_STEM_POOL 512u      /* with DS_PLAN_PACKED_RGB: MaxPool_2a_3x3 inside the stem kernel where the map allows  */
#define DS_PLAN_NO_SPLITK 1024u     /* A/B: never split the reduction of a fused-Winograd launch (ds_conv_wino4_splitk)      */
#define DS_PLAN_PACKED_RGB 32u      /* Conv2d_1a_7x7: x is the packed [N, H, W, 3] batch, filter stored [7][7][4][Cout]   */
typedef struct ds_conv_layer_plan {
    ds_conv_desc d;          /* descriptor of the chosen launch (dgrad: channel roles swapped, flipped taps)            */
    int32_t family;          /* DS_FAM_*                                                                                 */
    int32_t role, arith;
    int32_t partials;        /* P of the DS_EPI_STATS / DS_EPI_BNSUMS partials float[2][d.Cout][P] (0: none planned)     */
    int32_t w_cin, w_cout, k; /* the filter [k][k][w_cin][w_cout]                                                         */
    int32_t a_format;        /* fp8: format of the activation operand (DS_FP8_E4M3 forward, DS_FP8_E5M2 dgrad)           */
    int32_t x16_ok;          /* the family reads 16-bit activation storage (d.x_dtype)                                   */
    int64_t w_bytes;         /* bytes of the prepared filter; 0: the family reads the HWIO tensor in place               */
    int64_t wscale_floats;   /* fp8: floats of the filter's scale record                                                 */
    double alg_flops;        /* algorithmic FLOPs of one launch (2 M N K of the convolution, stem with Cin = 3)          */
    int32_t splitk;          /* > 1: DS_FAM_WINO4 runs as ds_conv_wino4_splitk with that many reduction slices            */
    int32_t reserved0;
    int64_t ws_bytes;        /* ... and ds_conv_run needs io.ws of that many bytes (0: none)                              */
} ds_conv_layer_plan;
typedef struct ds_conv_io {
    const float *bias;       /* DS_EPI_BIAS                                                                              */
    const float *mask;       /* DS_EPI_MASK source / DS_EPI_BNSUMS: the consumer's activation y (or z with d.mask_*)     */
    float *stats;            /* DS_EPI_STATS / DS_EPI_BNSUMS partials                                                    */
    const float *pivot;      /* DS_EPI_STATS pivot                                                                       */
    const float *x_amax;     /* fp8: max|x| record of the activation operand                                             */
    const float *wscale;     /* fp8: the filter's scale record (ds_conv_prepare_weights)                                 */
    const ds_bn_finalize_in_launch *fin;   /* nullable: ds_bn_finalize inside the launch (ds_conv_plan_finalize_tickets > 0)   */
    void *ws;                /* plan.ws_bytes > 0: scratch of at least that many bytes (16-byte aligned), private to the stream  */
    size_t ws_bytes;
} ds_conv_io;
int ds_conv_plan(ds_conv_layer_plan *plan, int32_t role, int32_t arith, uint32_t options, int32_t N, int32_t H, int32_t W,
                 int32_t w_cin, int32_t w_cout, int32_t k, int32_t stride, int32_t ldx, int32_t ldz, int32_t flags);
int ds_conv_plan_set_flags(ds_conv_layer_plan *plan, int32_t flags);
int ds_conv_plan_enable_bnsums(ds_conv_layer_plan *plan, int32_t ldy);
int ds_conv_plan_norm_supported(const ds_conv_layer_plan *plan);
int ds_conv_plan_bnb_supported(const ds_conv_layer_plan *plan);      /* would ds_conv_run take plan.d.bnb?                 */
int ds_conv_plan_enable_pool3(ds_conv_layer_plan *plan, uint8_t *argmax);
int ds_conv_plan_finalize_tickets(const ds_conv_layer_plan *plan);      /* > 0: ds_conv_run honours io.fin (that many ticket words) */
int ds_conv_prepare_weights(const ds_conv_layer_plan *plan, const float *w_hwio, void *w_prepared, float *wscale,
                            void *stream);
int ds_conv_run(const ds_conv_layer_plan *plan, const void *x, const void *w, float *z, const ds_conv_io *io, void *stream);

/* Conv2DBackpropFilter / MatMul-transposed (wgrad), split over pixels.
 *   dw[tap, ci, co] = sum_m x[pixel(m)+tap, ci] * dz[m, co]
 * only reached for the trainable scope image_model/inception_v1.py:229-250,302-303 and the
 * tf.get_variable weights im_text_rnn_model.py:89,98-104.  `ws` holds split partials.      */
size_t ds_conv_wgrad_workspace(const ds_conv_desc *d);
int ds_conv_wgrad(const ds_conv_desc *d, const float *x, const float *dz, int32_t lddz, float *dw,
                  void *ws, size_t ws_bytes, void *stream);

/* slim.batch_norm train mode (center, no scale), slim/nets/inception_utils.py:48-70.
 * finalize: partials (about `pivot`, the vector given to ds_conv_igemm; nullable; may alias `mean`) ->
 * mean, rstd, shift = beta - mean*rstd, moving statistics update.                           */
int ds_bn_finalize(const float *stats, int32_t P, int64_t count, int32_t C, const float *beta, const float *pivot,
                   float eps, float decay, float *mean, float *rstd, float *shift, float *moving_mean,
                   float *moving_var, void *stream);

/* ... for a layer whose z is stored CENTRED about the pivot (ds_conv_desc.z_dtype): additionally mean_c = mean - pivot and
 * shift_c = beta - mean_c * rstd, the vectors ds_bn_apply_relu_z16 / ds_bn_bwd_reduce / ds_bn_bwd_apply_z16 take with that z.  */
int ds_bn_finalize_centered(const float *stats, int32_t P, int64_t count, int32_t C, const float *beta, const float *pivot,
                            float eps, float decay, float *mean, float *rstd, float *shift, float *moving_mean,
                            float *moving_var, float *mean_c, float *shift_c, void *stream);
/* The same for up to four layers in ONE launch (the three convs that close an Inception block -- Branch_1 / Branch_2 3x3 and
 * Branch_3 1x1, inception_v1.py:86-95 -- finish on three streams; their finalizes ran as three single-purpose launches of one
 * workgroup per channel each).  Per channel the arithmetic is ds_bn_finalize's: bit-identical.                            */
typedef struct ds_bn_finalize_job {
    const float *stats;       /* float[2][C][P] about `pivot`                                          */
    int32_t P, C;
    int64_t count;
    const float *beta;
    const float *pivot;       /* nullable; may alias mean                                               */
    float *mean, *rstd, *shift;
    float *moving_mean, *moving_var;      /* nullable                                                   */
} ds_bn_finalize_job;
int ds_bn_finalize_multi(const ds_bn_finalize_job *jobs, int32_t njobs, float eps, float decay, void *stream);

/* ds_bn_finalize and the ds_bn_apply_relu pass that reads its result as ONE launch: the first C workgroups finalize a channel
 * each and publish, the others wait for all C (device-side ticket) and stream.  Saves the dependent-launch boundary between
 * the two (~4 us of idle GPU per layer).  ticket: two zero-initialised uint32 owned by the caller, one pair per layer that
 * may be in flight at a time (the launch leaves them zero).  Results bit-identical to the two launches.                   */
int ds_bn_finalize_apply_relu(const float *stats, int32_t P, int64_t count, int32_t C, const float *beta, const float *pivot,
                              float eps, float decay, float *mean, float *rstd, float *shift, float *moving_mean,
                              float *moving_var, const float *z, int64_t M, const struct ds_segments *dst, uint32_t *ticket,
                              void *stream);

/* y = relu(z*rstd + shift) scattered to up to 4 channel segments (branch outputs written
 * straight into the concat buffer: replaces tf.concat, inception_v1.py:96 ... :248).       */
typedef struct ds_segments {
    int32_t nseg;
    int32_t c_begin[4];       /* first channel of the segment in the [M,C] source/gradient           */
    int32_t c_end[4];
    int32_t ld[4];            /* pixel stride of the destination, in elements                        */
    void *ptr[4];             /* destination (already offset to its first channel)                   */
    int32_t dtype[4];         /* ds_bn_apply_relu only: DS_DTYPE_F32 (0) or DS_DTYPE_BF16 -- the activation is  */
                              /* stored rounded to bf16 (16-bit activation storage of the bf16 / fp8          */
                              /* configurations); gradient segments are always fp32                           */
    float *amax[4];           /* ds_bn_apply_relu only, nullable: device word that receives max(y) of the segment by  */
                              /* atomic max (the caller zeroes it once per step): the per-tensor scale of the fp8     */
                              /* conv that reads the segment, without a separate ds_absmax pass                     */
    const float *ptr2[4];     /* ds_bn_bwd_apply / _bf16 only, nullable: a SECOND addend of the segment's gradient with */
                              /* the same layout (dy = *ptr + *ptr2).  An Inception block's input gradient is the sum of */
                              /* the fused 1x1 dgrad's output and Branch_3's pool gradient: where the dgrad cannot        */
                              /* accumulate (16-bit configurations) the two stay separate tensors, written concurrently   */
                              /* on two streams, and the one consumer adds them as it reads                              */
} ds_segments;
int ds_bn_apply_relu(const float *z, int64_t M, int32_t C, const float *rstd, const float *shift,
                     const ds_segments *dst, void *stream);
/* The same reading z from bf16 storage (ds_conv_desc.z_dtype = DS_DTYPE_BF16: the 16-bit configurations' frozen layers). */
int ds_bn_apply_relu_z16(const void *z16, int64_t M, int32_t C, const float *rstd, const float *shift,
                         const ds_segments *dst, void *stream);
/* slim.batch_norm with is_training=False (evaluate_*, mode != 'train': im_text_rnn_model.py:65,171-207):
 * rstd = rsqrt(moving_variance + eps), shift = beta - moving_mean*rstd, then ds_bn_apply_relu.  */
int ds_bn_infer_prepare(const float *beta, const float *moving_mean, const float *moving_var, float eps,
                        int32_t C, float *rstd, float *shift, void *stream);

/* BatchNorm(train)+ReLU backward: g = dy*(y>0); dbeta = sum g; dz = rstd*(g - mean(g) - xhat*mean(g*xhat)).
 * dy is gathered from the same segments the forward scattered to.                          */
int ds_bn_bwd_partials(int64_t M, int32_t C);
/* z: [M, ldz] (ldz >= C: a column sub-range of a layer is reduced by passing z, mean, rstd, shift offset to its
 * first channel and dy segments numbered from 0); partials float[2][C][P].  z_dtype DS_DTYPE_BF16: the kernel is run
 * on a POOLED activation in 16-bit storage (z := ypool, mean := beta, rstd := 1, shift := 0: the sums of a layer that
 * feeds nothing but a max pool, from a quarter of the elements).                                                */
int ds_bn_bwd_reduce(const void *z, int32_t ldz, int32_t z_dtype, const ds_segments *dy, int64_t M, int32_t C,
                     const float *mean, const float *rstd, const float *shift, float *partials, void *stream);
int ds_bn_bwd_finalize(const float *partials, int32_t P, int64_t M, int32_t C, float *dbeta, float *coef,
                       void *stream);
/* The same finalize when the sums of a layer's column segments come from different producers: kind 0 = partials of
 * ds_bn_bwd_reduce over that column range (sum g, sum g*xhat); kind 1 = partials a dgrad launch emitted with
 * DS_EPI_BNSUMS (sum g, sum g*y: for y > 0, y = xhat + beta, so sum g*xhat = sum g*y - beta*sum g).  s[i] / q[i] point
 * at the P[i] partials of the segment's first channel; channel c of the segment is P[i]*(c - c_begin[i]) further. */
typedef struct ds_bn_sum_segments {
    int32_t nseg;
    int32_t c_begin[4], c_end[4];
    int32_t P[4];
    int32_t kind[4];
    const float *s[4];
    const float *q[4];
    int32_t P2[4];            /* > 0: a SECOND source of the same kind for the segment (the sums are linear in the gradient: */
    const float *s2[4];       /* with dy = a + b in two tensors, each producer emits the sums of its own addend), P2[i]    */
    const float *q2[4];       /* partials per channel at s2[i] / q2[i]; added behind the first source's                     */
} ds_bn_sum_segments;
int ds_bn_bwd_finalize_segs(const ds_bn_sum_segments *sg, int64_t M, int32_t C, const float *beta, float *dbeta,
                            float *coef, void *stream);
/* ... and when the segments are different LAYERS (the block-closing convs above, whose z and dy are column slices of the
 * block's concat buffers): beta[i] / dbeta[i] (dbeta or dbeta[i] nullable) are segment i's own vectors, indexed from its
 * first channel; coef is float[2][C] over the concatenated columns -- what ONE ds_bn_bwd_apply over those columns reads.  */
int ds_bn_bwd_finalize_multi(const ds_bn_sum_segments *sg, int64_t M, int32_t C, const float *const *beta,
                             float *const *dbeta, float *coef, void *stream);
/* ds_bn_bwd_finalize_segs / _multi and the ds_bn_bwd_apply (dz_dtype DS_DTYPE_F32: dz [M, ldz]) / ds_bn_bwd_apply_bf16
 * (DS_DTYPE_BF16: dz [M, lddz]) pass behind it as ONE launch (see ds_bn_finalize_apply_relu; ticket as there).  beta / dbeta:
 * one vector for all columns, or -- beta_v non-null -- the segments' own (ds_bn_bwd_finalize_multi).  Bit-identical.     */
int ds_bn_bwd_finalize_apply(const ds_bn_sum_segments *sg, const float *beta, float *dbeta, const float *const *beta_v,
                             float *const *dbeta_v, float *coef, const float *z, int32_t ldz, const ds_segments *dy, int64_t M,
                             int32_t C, const float *mean, const float *rstd, const float *shift, void *dz, int32_t dz_dtype,
                             int32_t lddz, float *amax, uint32_t *ticket, void *stream);
/* amax (nullable): device word that receives max|dz| by atomic max (zeroed by the caller): the scale of an fp8 dgrad */
/* ldz: row stride of z AND dz in floats (>= C, % 4 == 0): a layer whose conv writes straight into its slice of the   */
/* Inception concat buffer is differentiated in place there                                                          */
int ds_bn_bwd_apply(const float *z, int32_t ldz, const ds_segments *dy, int64_t M, int32_t C, const float *mean,
                    const float *rstd, const float *shift, const float *coef, float *dz, float *amax, void *stream);
/* The same with dz written to a SEPARATE bf16 tensor (pixel stride lddz) and z left as it is: the 16-bit configurations' 1x1  */
/* input gradients (ds_conv_bf16 with x_dtype = DS_DTYPE_BF16) read 2 instead of 4 bytes per element and get exactly the values */
/* they would have rounded on load (RNE), so Conv2DBackpropInput has the same bits                                            */
int ds_bn_bwd_apply_bf16(const float *z, int32_t ldz, const ds_segments *dy, int64_t M, int32_t C, const float *mean,
                         const float *rstd, const float *shift, const float *coef, void *dz16, int32_t lddz, float *amax,
                         void *stream);

/* ... and with z itself in bf16 storage (ds_conv_desc.z_dtype): dz always goes to the separate bf16 tensor               */
int ds_bn_bwd_apply_z16(const void *z16, int32_t ldz, const ds_segments *dy, int64_t M, int32_t C, const float *mean,
                        const float *rstd, const float *shift, const float *coef, void *dz16, int32_t lddz, float *amax,
                        void *stream);

/* slim.max_pool2d SAME/VALID (inception_v1.py:67,79,94,118,208) with arg-max record, and MaxPoolGrad. */
/* act_dtype: storage type of x AND y (DS_DTYPE_F32 / DS_DTYPE_BF16; max and arg-max are exact in either).   */
int ds_maxpool_fwd(const void *x, void *y, uint8_t *argmax, int32_t N, int32_t H, int32_t W, int32_t C,
                   int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW,
                   int32_t act_dtype, void *stream);
/* BatchNorm + ReLU + 3x3 max pool of a conv that feeds nothing else (Conv2d_1a_7x7 -> MaxPool_2a_3x3,
 * Conv2d_2c_3x3 -> MaxPool_3a_3x3, inception_v1.py:63-79): y = relu(rstd*maxpool(z) + shift), identical to
 * maxpool(relu(bn(z))) because rstd > 0; the full-resolution activation is never materialised.           */
int ds_maxpool_bn_relu_fwd(const float *z, const float *rstd, const float *shift, void *y, uint8_t *argmax,
                           int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad_t,
                           int32_t pad_l, int32_t OH, int32_t OW, int32_t y_dtype, float *amax, void *stream);
/* ... and its backward: BatchNorm(+ReLU) backward of that conv straight from the POOLED gradient (3x3 stride-2
 * SAME pools).  MaxPoolGrad's full-resolution result is rebuilt per 2x2 input patch on the fly instead of being
 * written and re-read twice.  reduce -> partials float[2][C][P] (P = ds_bn_pool_bwd_partials), then
 * ds_bn_bwd_finalize as usual, then apply (dz may alias z).                                                   */
int ds_bn_pool_bwd_partials(int32_t N, int32_t OH, int32_t OW, int32_t C);
int ds_bn_pool_bwd_reduce(const float *z, const float *dpool, const uint8_t *argmax, int32_t N, int32_t H, int32_t W,
                          int32_t C, int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW, const float *mean,
                          const float *rstd, const float *shift, float *partials, void *stream);
int ds_bn_pool_bwd_apply(const float *z, const float *dpool, const uint8_t *argmax, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t pad_t, int32_t pad_l, int32_t OH, int32_t OW, const float *mean,
                         const float *rstd, const float *shift, const float *coef, float *dz, void *stream);
int ds_maxpool_bwd(const float *dy, const uint8_t *argmax, float *dx, int32_t accumulate, int32_t N,
                   int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                   int32_t OH, int32_t OW, void *stream);

/* MaxPoolGrad of a 3x3 / 1 SAME pool (Branch_3, inception_v1.py:94 ... :246) that ALSO emits the BatchNorm-backward sums of
 * the layer(s) whose activation y the pool read: with accumulate = 1 and dx holding the other paths' gradient the launch sees
 * the complete dL/dy, so per channel   partials[0][c][p] = sum g,  partials[1][c][p] = sum g*y,  g = dx (y > 0)
 * over workgroup p's pixels -- the DS_EPI_BNSUMS form (ds_bn_bwd_finalize_segs kind 1), float[2][C][P] with
 * P = ds_maxpool3_bwd_sums_partials(N, W, C).  y: [N, H, W, C] fp32 or bf16 (y_dtype); dx is bit-identical to ds_maxpool_bwd's. */
int ds_maxpool3_bwd_sums_partials(int32_t N, int32_t W, int32_t C);
int ds_maxpool3_bwd_sums(const float *dy, const uint8_t *argmax, float *dx, int32_t accumulate, const void *y,
                         int32_t y_dtype, int32_t N, int32_t H, int32_t W, int32_t C, float *partials, void *stream);

/* The same MaxPoolGrad with its output gradient dy in bf16 storage (the 16-bit labels: Branch_3's 1x1 Conv2DBackpropInput writes
 * it rounded, ds_conv_desc.z_dtype on that launch -- 2 B written and 2 B read per element instead of 4).  partials == NULL: the
 * plain pass (what ds_maxpool_bwd does for k = 3, stride 1, SAME; y ignored); else with the sums, as ds_maxpool3_bwd_sums.     */
int ds_maxpool3_bwd_dy16(const void *dy16, const uint8_t *argmax, float *dx, int32_t accumulate, const void *y, int32_t y_dtype,
                         int32_t N, int32_t H, int32_t W, int32_t C, float *partials, void *stream);

/* slim.avg_pool2d 7x7 VALID + slim.dropout (inception_v1.py:299-301).  mask_in==NULL: draw
 * Bernoulli(keep) from a counter-based generator keyed by (seed, element); the mask used is
 * written to mask_out (needed by the backward).  keep>=1 disables dropout.  seed_dev (nullable)
 * is a device counter added to `seed`, so a captured hipGraph draws a fresh mask per replay.  */
int ds_avgpool_dropout_fwd(const float *x, int32_t N, int32_t HW, int32_t C, float keep, uint64_t seed,
                           const uint64_t *seed_dev, const float *mask_in, float *mask_out, float *out,
                           void *stream);
int ds_avgpool_dropout_bwd(const float *dout, const float *mask, int32_t N, int32_t HW, int32_t C,
                           float keep, float *dx, void *stream);

/* tf.nn.embedding_lookup (im_text_rnn_model.py:85): out row (b,t) <- table[ids[b,t]].
 * time_major!=0 writes row t*B+b (the layout the LSTM consumes), else b*T+t.               */
int ds_gather_rows(const float *table, const int64_t *ids, float *out, int32_t B, int32_t T, int32_t D,
                   int64_t table_rows, int32_t time_major, void *stream);

/* Gradient of tf.nn.embedding_lookup w.r.t. the table (a dense [table_rows, D] result of what TF returns as
 * IndexedSlices): dtable[v] = sum_{(b,t): ids[b,t]==v} dx[row(b,t)], rows summed in ascending b*T+t order
 * (deterministic).  The reference's table is trainable=False (im_text_rnn_model.py:82), so this only backs
 * the optional fine-tuning switch.  D <= 512.                                                        */
int ds_embedding_grad(const float *dx, const int64_t *ids, float *dtable, int32_t B, int32_t T, int32_t D,
                      int64_t table_rows, int32_t time_major, void *stream);

/* BasicLSTMCell gate math + dynamic_rnn length masking (im_text_rnn_model.py:89-90).
 * gates [B,4H] holds x_t*Wx+bias (i,j,f,o) on entry -- the recurrent term h_{t-1}*Wh is added
 * from `nslabs` split-K slabs rec_slabs[s*slab_stride + ...] (nslabs may be 0 when the GEMM
 * accumulated in place) -- and the activations (sigmoid i, tanh j, sigmoid(f+forget_bias),
 * sigmoid o) on exit.                                                                        */
int ds_lstm_cell_fwd(float *gates, const float *rec_slabs, int32_t nslabs, int64_t slab_stride,
                     const float *c_prev, const float *h_prev, const int64_t *seq_len, int32_t t, int32_t B,
                     int32_t H, float forget_bias, float *c_out, float *h_out, void *stream);
/* one BPTT step: d(loss)/d(h_t) = dh + sum of `nslabs` split-K slabs of dgates_{t+1}*Wh^T;
 * dc = grad wrt c_t; writes dgates [B,4H], dc_prev, and dh_carry (= the h-gradient for rows with
 * t >= seq_len, which dynamic_rnn copies through, else 0).                                   */
int ds_lstm_cell_bwd(const float *acts, const float *c_t, const float *c_prev, const float *dh,
                     const float *dh_slabs, int32_t nslabs, int64_t slab_stride, const float *dc,
                     const int64_t *seq_len, int32_t t, int32_t B, int32_t H, float *dgates, float *dc_prev,
                     float *dh_carry, void *stream);

/* Length-sorted batches for the text tower (round 6).  ds_seq_sort_desc: perm[j] = the sample that takes sorted position j
 * (descending length, ties by ascending index: deterministic), len_sorted[j] = min(seq_len[perm[j]], T); one small launch,
 * B <= 4096.  ds_permute_rows: dst row j = src row perm[j] (gather = 1) or dst row perm[j] = src row j (gather = 0); rows of
 * `cols` elements of elem_bytes 4 or 8, row strides lds / ldd in elements.                                              */
int ds_seq_sort_desc(const int64_t *seq_len, int32_t B, int32_t T, int32_t *perm, int64_t *len_sorted, void *stream);
int ds_permute_rows(const void *src, int64_t lds, void *dst, int64_t ldd, const int32_t *perm, int32_t rows, int32_t cols,
                    int32_t elem_bytes, int32_t gather, void *stream);
/* The same recurrence for the WHOLE sequence in one launch per direction (lstm_seq: BasicLSTMCell +
 * tf.nn.dynamic_rnn(sequence_length) + the gather_nd of the last valid output, im_text_rnn_model.py:89-92):
 * the recurrent weights stay in registers as MFMA fragments partitioned over the workgroups, cell state and
 * carried gradients stay in registers, h_t / dgates_t are exchanged between the workgroups of a 32-row group
 * through write-through stores and an arrival counter (no grid-wide barrier: batch rows are independent).
 *   wh     rows [D, D+H) of the TF kernel: [H, 4H], row stride ldw, gate order i, j, f, o
 *   gates  [T, B, 4H]   in: x_t Wx + bias for every step (the hoisted input projection); out: activations
 *   h, c   [T+1, B, H]  slot 0 = initial state (zeros in the reference), slot t+1 = state after step t with
 *                       dynamic_rnn's copy-through past seq_len, so h[T] is the last valid output
 *   rows   row groups (32 batch rows each) a workgroup walks per time step: 1, 2, 4 or 8.  1 = the shortest
 *          sequence time (one workgroup per (16 units, 32 rows) pair); R > 1 = 1/R of the CUs for a longer time,
 *          one row group's hand-off wait covered by work on the others (beside a concurrent kernel that needs
 *          whole CUs).  Scheduling only: every cell's arithmetic and summation order are the same.  With rows = 1
 *          and H <= 512 the launch is XCD-local: the H / 16 workgroups of a row group share one XCD, so the per-step
 *          exchange stays inside its L2 (text-only step at B = 256: 1.67 -> 1.27 ms).  With rows = 1, H = 64 ... 512
 *          and a batch whose ceil(B / 16) * H / 16 workgroups all fit the device, the row groups are 16 rows
 *          (v_mfma_f32_16x16x4_f32 instead of 32x32x2: another summation order of the recurrent product, same
 *          accuracy): a 32-row step is bound by one CU's matrix rate, so a small batch spreads over twice the CUs
 *          (text-only step at B = 64: 0.89 -> 0.75 ms).
 *   ws     ds_lstm_seq_workspace(B, H) bytes of device scratch, ZEROED ONCE BY THE CALLER: forward and backward
 *          arrival counters (each launch re-zeroes its own) and one sticky error word per direction.
 * Supported: H in {32, 64, 128, 256, 512, 1024} with H / 16 <= the device's compute units (ds_lstm_seq_supported);
 * other sizes use the step-wise pair above.  All H / 16 workgroups of a row group must become resident for the
 * launch to finish; every wait is bounded, a timeout sets the direction's error word (results then invalid) and
 * ds_lstm_seq_status -- called wherever the caller synchronises anyway -- reports it instead of a hang:
 * 0 = ok, bit 0 = a forward launch timed out, bit 1 = a backward launch; a reported failure is cleared by the read
 * (the words are sticky only until then).  The XCD-local grid is used only when min(row groups, 8) * H / 16 workgroups
 * fit the device at once; smaller or partitioned devices get the 2-D grid.  Re-entrant: no process-global state;
 * two sequences on two streams need two workspaces.                                                        */
/* rows | DS_LSTM_SKIP_MASKED (round 6): a row group stops exchanging after its LONGEST row's last step -- past it dynamic_rnn
 * only copies the state through (forward: carried h / c are written; backward: dgates = 0), so the steps cost a few stores
 * instead of a recurrent GEMM and a hand-off.  Per-row results are unchanged to the bit (the activations stored in `gates` are
 * not written for those steps: nothing reads them).  Pays when the rows of a group have similar lengths: sort the batch with
 * ds_seq_sort_desc first.                                                                                              */
#define DS_LSTM_SKIP_MASKED 256
int ds_lstm_seq_supported(int32_t B, int32_t H);
size_t ds_lstm_seq_workspace(int32_t B, int32_t H);
int ds_lstm_seq_fwd(float *gates, const float *wh, int32_t ldw, float *h, float *c, const int64_t *seq_len, int32_t T,
                    int32_t B, int32_t H, float forget_bias, int32_t rows, void *ws, size_t ws_bytes, void *stream);
/* BPTT: dgates [T, B, 4H] (zero rows past seq_len) from d(loss)/d(h[T]) = dh_last [B, ld_dh]; acts = the
 * activations ds_lstm_seq_fwd left in `gates`.  The two weight gradients are GEMMs over dgates afterwards. */
int ds_lstm_seq_bwd(const float *acts, const float *wh, int32_t ldw, const float *c, const float *dh_last,
                    int32_t ld_dh, const int64_t *seq_len, int32_t T, int32_t B, int32_t H, float *dgates, int32_t rows,
                    void *ws, size_t ws_bytes, void *stream);
int ds_lstm_seq_status(void *ws, int32_t B);
/* Debug aid (process-global, never called by the product path): device buffer of T*8 uint64; workgroup (0,0) of the
 * following ds_lstm_seq_fwd launches stamps s_memtime at its phase boundaries (scripts/lstm_phase_prof.py).  NULL = off. */
#ifdef DS_TUNING
int ds_debug_lstm_seq_set_profile(void *buf);
int ds_debug_lstm_seq_set_profile_bwd(void *buf);      /* the same for the ds_lstm_seq_bwd launches (row t = time step t) */
#endif

/* slim.losses.softmax_cross_entropy + its gradient (im_text_rnn_model.py:124-125):
 * loss[0] = mean_b(logsumexp(z_b) - z_b[y_b]); dlogits = (softmax - onehot) * grad_scale / B;
 * grad_scale_dev (nullable) is a device scalar multiplied in (autograd's upstream gradient);
 * loss / dlogits may each be NULL. */
int ds_softmax_ce(const float *logits, const int64_t *labels, int32_t B, int32_t C, float grad_scale,
                  const float *grad_scale_dev, float *loss, float *dlogits, void *stream);

/* tf.train.AdamOptimizer.apply_gradients over one flat parameter buffer
 * (im_text_rnn_model.py:134-135).  g_eff = g*grad_scale + (i < n_wd ? wd*theta : 0): the first
 * n_wd entries are the slim conv `weights` that carry the L2 regulariser
 * (slim/nets/inception_utils.py:63-64).  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the host;
 * lr_t_dev (nullable) overrides it from device memory (hipGraph replay). */
int ds_adam_tf(float *theta, const float *g, float *m, float *v, int64_t n, int64_t n_wd, float wd,
               float grad_scale, float lr_t, const float *lr_t_dev, float beta1, float beta2, float eps,
               void *stream);

/* small helpers (deterministic two-stage reductions; scratch is caller-provided) */
/* out[0] = sum x^2 (= 2*tf.nn.l2_loss); scratch >= 256 floats                                */
int ds_sumsq(const float *x, int64_t n, float *scratch, float *out, void *stream);
/* BiasAddGrad: out[c] = sum_m x[m*ld + c]; scratch >= 64*C floats                            */
int ds_colsum(const float *x, int64_t M, int32_t C, int32_t ld, float *scratch, float *out, void *stream);
/* Second half of a split-K GEMM (ds_conv_igemm with ds_conv_desc.splits > 1 leaves slab s of partial sums at
 * z + s * z_split_stride, row stride lds): out[m][n] = epilogue(sum_s slabs[s][m][n]), slabs added in index order
 * (deterministic); flags: DS_EPI_BIAS / DS_EPI_ACCUM / DS_EPI_MASK / DS_EPI_RELU, applied in ds_conv_igemm's order.
 * For the batch x features GEMMs of the heads (MatMul + BiasAdd + Relu, im_text_rnn_model.py:95-105; Logits,
 * image_model/inception_v1.py:302), whose single row tile would otherwise walk K serially. */
int ds_slab_epilogue(const float *slabs, int32_t splits, int64_t slab_stride, int32_t lds, int64_t M, int32_t N, float *out,
                     int32_t ldo, const float *bias, const float *mask, int32_t ldmask, int32_t flags, void *stream);
int ds_copy2d(const float *src, int32_t lds, float *dst, int32_t ldd, int64_t rows, int32_t cols,
              void *stream);
int ds_pad_channels(const float *src, int32_t cs, float *dst, int32_t cd, int64_t pixels, void *stream);
int ds_fill(float *dst, int64_t n, float value, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DS_KERNELS_H */
