/*
 * epos_hip.h -- C ABI of libepos_hip.so, the MI355X (gfx950) implementation of
 * the EPOS inference hot path (thodan/epos): DeepLabv3+/Xception-65 forward,
 * many-to-many 2D-3D correspondence extraction, per-object PnP-RANSAC.
 *
 * Conventions
 *   - plain pointers and sizes only; every `d_*` / `const float* x` marked
 *     [device] is a DEVICE pointer (HBM), everything else is host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     device entry points only enqueue work on it and never synchronise, so
 *     they can be captured into a hipGraph;
 *   - return value: 0 = OK, < 0 = error (EPOS_E_*); -1000 - hipError_t for HIP
 *     runtime failures. epos_last_error() gives a message for the last failure
 *     on the calling thread;
 *   - activations are NHWC fp32 (the reference's layout and dtype,
 *     SURVEY.md section 8); `ld*` are row strides in ELEMENTS so that a kernel
 *     can read or write a channel slice of a wider (concat) buffer.
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef EPOS_HIP_H_
#define EPOS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPOS_OK 0
#define EPOS_E_INVALID (-1)   /* bad argument (NULL, non-multiple-of-4 channels, ...) */
#define EPOS_E_CAPACITY (-2)  /* caller-provided output capacity too small */
#define EPOS_E_NODEVICE (-3)  /* no HIP device available */
#define EPOS_E_INTERNAL (-4)  /* a device-side consistency check failed (see epos_last_error) */
#define EPOS_E_HIP_BASE (-1000)

#define EPOS_ABI_VERSION 7   /* 2: EposPointwiseArgs.Ws, EposConv3x3Args.Ws, split weight packing; 3: epos_separable_conv_f32; 4: epos_solve_pnp_ransac; 5: fp16-pair GEMM (Wh, a_amax, c_amax, epos_pack_pointwise_weights_h2, epos_absmax_f32); 6: epos_separable_conv_f32 with fp16-pair intermediates on the fp16-pair kernel, epos_separable_conv_fused_state ; 7 (round 5, the diet): the measured-slower opt-ins left the library -- EposSepConvArgs lost sync / stats and epos_separable_conv_f32 issues the two launches, EposPointwiseArgs.softmax64 is reserved, removed: epos_separable_conv_sync_words, epos_separable_conv_fused_state, epos_set_h2_latency_tile_limit, epos_pointwise_workspace_bytes, epos_pointwise_conv_grouped_ws_f32, epos_pointwise_conv_grouped_sk_f32 */

int epos_abi_version(void);
const char* epos_last_error(void);
/* Number of visible HIP devices (>= 0) or a negative error. */
int epos_device_count(void);
/* Measurement aid (bench.py): a one-wave kernel on `stream` that spins for
 * `microseconds` and writes {shader-clock cycles, 100 MHz ticks} to the DEVICE
 * buffer out2[2] -- cycles / ticks * 100 = the core clock in MHz while the
 * surrounding work runs (the fp32 MFMA roof scales with it). */
int epos_clock_probe(int64_t* out2, int microseconds, void* stream);

/* ------------------------------------------------------------------------- *
 * Network layers (replace the TF1.12 ops that model.py / net_xception.py lower
 * to; call sites: SURVEY.md section 2.2).
 * ------------------------------------------------------------------------- */

/* Packs a 1x1-conv weight matrix W[K][N] (TF HWIO with H=W=1, i.e. [Cin][Cout])
 * into the tile order the MFMA kernel streams: [ceil(K/32)*8][Npad][4] floats,
 * Npad = round_up(N, 128), zero padded. Host-side helper (host pointers).
 * Returns the number of floats written (or required, if dst == NULL). */
int64_t epos_pack_pointwise_weights(const float* w_kn, int K, int N, float* dst);

/* The same matrix for the split-operand GEMM (pointwise_gemm_split_f32): every fp32
 * weight is cut into three bf16 pieces (8 + 8 + 8 significand bits; the cut itself is
 * exact, hi + mid + lo == w) and stored in the fragment order of
 * v_mfma_f32_32x32x16_bf16. The kernel then forms each product a*w from SIX of the nine
 * piece products (the three lowest-order cross terms, below 2^-23 |a*w|, are dropped):
 * fp32-equivalent, not exact -- its measured error against fp64 is below the fp32-MFMA
 * kernel's (tests/test_gpu_layers.py). Layout:
 * [ceil(N/128)][ceil(K/16)][4 column blocks][3 pieces][64 lanes][8 bf16], zero padded.
 * Host-side helper (host pointers). Returns the number of BYTES written (or required,
 * if dst == NULL). */
int64_t epos_pack_pointwise_weights_split(const float* w_kn, int K, int N, void* dst);

/* The same matrix for the fp16-pair GEMM (pointwise_gemm_h2_f32, the default fp32 GEMM
 * since round 3): every column n is scaled by a power of two 2^e_n that puts its largest
 * weight into [2^14, 2^15), and every scaled weight t is stored as TWO fp16 values
 *     hi = rn_fp16(t),  mid = rn_fp16((t - hi) * 2^11)        (t - hi is exact in fp32)
 * so that hi + mid * 2^-11 carries 22-23 significant bits of t (error <= 1 ulp of the 24-bit
 * significand). The kernel forms a*w from THREE fp16 MFMA products,
 *     ah*wh                      (main accumulator)
 *     ah*wm + am*wh              (correction accumulator, scaled by 2^-11 in the epilogue)
 * (am*wm, below 2^-22 |a*w| and of random sign, is dropped): fp32-equivalent, not exact; its
 * measured error against fp64 is below the fp32-MFMA kernel's (tests/test_gpu_layers.py).
 * Half the matrix-pipe work of the bf16 x 6 split above, 4 instead of 6 bytes per weight.
 * Layout: [ceil(N/128)][ceil(K/16)][4 column blocks][2 pieces][64 lanes][8 fp16] (fragment
 * order of v_mfma_f32_32x32x16_f16), then round_up(N,128) floats 2^-e_n.
 * Every finite matrix qualifies (round 6): a scaled weight t is reproduced to
 *     |hi + mid * 2^-11 - t| <= max(2^-22 |t|, 2^-36)
 * i.e. to full precision down to 2^-28 of its column's maximum and, below that (fp16
 * subnormal pieces, which the matrix pipe does not flush), with an ABSOLUTE error of at most
 * 2^-50 x the column maximum -- the same graceful degradation as the activation side, and
 * below the fp32 rounding of any sum the column's larger weights take part in. (Until ABI 7's
 * first release a single such weight made the packer refuse the whole matrix.) The function
 * returns 0 and writes nothing only for Inf / NaN weights or a column whose scale leaves the
 * fp32 exponent range (column maximum outside ~2^-85 .. 2^115); the caller then keeps the
 * bf16 x 6 split kernel for this layer (Wh = NULL).
 * Host-side helper (host pointers). Returns the number of BYTES written (or required, if
 * dst == NULL; 0 = not representable). */
int64_t epos_pack_pointwise_weights_h2(const float* w_kn, int K, int N, void* dst);

/* Absolute-maximum slots. The fp16-pair GEMM scales its fp32 A operand by a power of two
 * chosen from an UPPER BOUND of max|A| that it reads from device memory when it starts:
 * a slot is 64 uint32 words [device] holding the bit patterns of non-negative floats; the
 * bound is the maximum over the 64 words. Producers combine into a slot with atomic max
 * (the GEMM epilogues do when c_amax is given; epos_absmax_f32 does for any other tensor);
 * the caller zeroes a slot (epos_amax_clear) before the first producer of the step runs.
 * Any upper bound is valid (max-pool / subsample / bilinear resize outputs may reuse their
 * input's slot); a bound that is too large by 2^j costs j of the ~27 octaves below the
 * bound within which elements keep full precision. Overflow is impossible by construction.
 *
 * epos_absmax_f32: slot = max(slot, max |X[r, 0..cols)|) over `rows` rows of ldx floats. */
#define EPOS_AMAX_WORDS 64
int epos_absmax_f32(const float* X, int64_t ldx, int64_t rows, int64_t cols,
                    uint32_t* slot, void* stream);
/* Zeroes n_slots consecutive slots (a kernel on `stream`: a memset NODE of a captured graph
 * is not reliably ordered before the kernel nodes behind it when graphs replay concurrently). */
int epos_amax_clear(uint32_t* slots, int64_t n_slots, void* stream);

/* out[m, n] = act( sum_k A[row(m), k] * W[k, n] + bias[n] (+ R[m, n]) )
 * = slim.conv2d(kernel 1x1, stride `sub`) + folded BatchNorm (+ residual add)
 * (+ ReLU): net_xception.py:167-182 (pointwise half of separable_conv2d_same),
 * :296-302 (shortcut), model.py:223-224,237,257-258,349-352 (ASPP/decoder 1x1),
 * model.py:449-456 (logits, bias, no BN).
 * A [device]: rows of `lda` floats; with sub > 1 the row of output pixel
 * (b, y, x) is input pixel (b, y*sub, x*sub) of a [B, Hi, Wi] map (TF 'SAME'
 * 1x1 stride-2 conv samples even indices); M = B*Ho*Wo output pixels.
 * Wp [device]: packed by epos_pack_pointwise_weights; bias [device]: Npad floats
 * (or NULL); R [device]: optional residual rows of ldr floats; C [device]: output
 * rows of ldc floats. K % 4 == 0, lda % 4 == 0 required. */
typedef struct EposPointwiseArgs {
  const float* A; int64_t lda;
  const float* Wp; const float* bias;
  const float* R; int64_t ldr;
  float* C; int64_t ldc;
  int32_t M, N, K;
  int32_t relu;       /* apply ReLU last */
  int32_t relu_in;    /* apply ReLU to A on load (pre-activation) */
  int32_t sub;        /* spatial subsampling of A rows (1 or 2) */
  int32_t Ho, Wo, Hi, Wi;   /* only read when sub > 1 */
  const void* Ws;     /* optional [device]: the same weights packed by
                       * epos_pack_pointwise_weights_split (NULL = not provided) */
  /* ---- ABI 5: fp16-pair GEMM (taken when Wh is given, relu_in == 0 and M > 8;
   *      EPOS_GEMM_H2=0 disables it; otherwise Ws -> bf16 x 6 kernel, else fp32 MFMA) */
  const void* Wh;     /* optional [device]: epos_pack_pointwise_weights_h2 of the weights */
  const uint32_t* a_amax;   /* optional [device]: slot bounding max|A| (see above). NULL
                       * with Wh given: the library measures A itself first (one memset +
                       * one reduction launch into an internal slot ring of 256 slots --
                       * convenient for single calls; plans pass their own slots) */
  const uint32_t* a_amax2;  /* optional second slot: the bound is the larger of the two
                       * (A = a concat written by two producers) */
  float a_gain, a_bias; /* bound = a_gain * slot + a_bias (a_gain == 0 reads as 1, 0):
                       * A = a depthwise conv of the tensor the slot describes,
                       * a_gain = max_c sum_taps |w|, a_bias = max_c |bias| */
  int32_t a_presplit; /* A is already fp16 pairs (EposDepthwiseArgs.y_h2 wrote it with the
                       * same a_amax / a_amax2 / a_gain / a_bias): needs Wh and a_amax */
  uint32_t* c_amax;   /* optional [device]: slot that receives max|C| over the elements
                       * this call writes (atomic max). Needs the float4 epilogue:
                       * N % 4 == 0, ldc % 4 == 0, C (and R) 16-byte aligned, relu_in == 0,
                       * M > 8; EPOS_E_INVALID otherwise */
  /* ---- ABI 6 */
  float* col_sums;    /* optional [device]: [ceil(M / 32)][col_ld] -- the epilogue also writes,
                       * for every block of 32 consecutive rows, the column sums of what it
                       * stores (after bias / ReLU), in a fixed order. With rows = the pixels of
                       * an image (H*W % 32 == 0) epos_global_avg_pool_partial_f32 turns them
                       * into the per-image channel means (model.py:220) without re-reading
                       * the tensor. fp16-pair kernel only (EPOS_E_INVALID otherwise), no
                       * residual, N % 4 == 0 */
  int64_t col_ld;
  int32_t c_stream;   /* != 0: write C with streaming (non-temporal) stores -- for outputs that
                       * are not re-read soon (the 413 MB of dense heads would otherwise sweep
                       * the 256 MB Infinity Cache clean of the other images' working sets);
                       * fp16-pair kernel's float4 epilogue only, ignored elsewhere */
  int32_t reserved0;  /* 0 (ABI 6 carried a fused fragment softmax here; measured slower than
                       * its own launch, removed in ABI 7: profiles/r04/ab_head_softmax.txt) */
} EposPointwiseArgs;
int epos_pointwise_conv_f32(const EposPointwiseArgs* args, void* stream);

/* Grouped form: `count` (1..8) independent problems in ONE launch, so that small
 * problems (the four ASPP branches, the three logit heads, a shortcut next to a
 * separable conv) fill the chip together. All problems of a group must agree on
 * relu_in, on a_presplit and on whether a residual is present. On the fp16-pair kernel a
 * group WITH residuals is issued as `count` consecutive launches (same bits: an element's value
 * does not depend on the launch that computes it; the one-launch form with residuals measured
 * no better and left the library in ABI 7); if one of them fails the earlier ones are already
 * enqueued and epos_last_error() names the failing problem's index. */
int epos_pointwise_conv_grouped_f32(const EposPointwiseArgs* args, int count,
                                    void* stream);

/* (ABI <= 6 also had a persistent stream-K form of the grouped GEMM on the fp32-MFMA ring,
 * epos_pointwise_conv_grouped_sk_f32 / _ws_f32 + a workspace: correct, tested, and slower
 * than the data-parallel kernels end to end (177 vs 213 images/s in round 1); removed in
 * ABI 7 -- DESIGN.md "Stream-K", history up to commit 1b05025.) */

/* Dense 3x3 conv + folded BatchNorm (+ ReLU) as an IMPLICIT GEMM: the im2col matrix
 * only exists as LDS tiles filled by LDS-DMA from the shifted input pixels. Semantics =
 * resnet_utils.conv2d_same (external/slim/nets/resnet_utils.py:77-122): stride 1 ->
 * 'SAME' with dilation `rate` (zero pad = rate); stride 2 -> zero pad `rate` on every
 * side + VALID. Output (y, x) reads input (y*stride + (ky-1)*rate, x*stride + (kx-1)*rate).
 * Replaces conv1_2 (net_xception.py:462-463) and the stride-1 root convs and the
 * bottleneck 3x3 convs of net_resnet_v1_beta.py:38-112. X [device]: [B, H, W] pixels,
 * rows of ldx floats, Cin channels used (Cin % 32 == 0); Wp [device]:
 * epos_pack_pointwise_weights of the [9*Cin][Cout] matrix whose row (ky*3+kx)*Cin + c is
 * TF's HWIO w[ky][kx][c][:]; bias [device]: round_up(Cout,128) floats or NULL; Y
 * [device]: [B, Ho, Wo] rows of ldy floats, Ho = (H-1)/stride + 1. */
typedef struct EposConv3x3Args {
  const float* X; int64_t ldx;
  const float* Wp; const float* bias;
  float* Y; int64_t ldy;
  int32_t B, H, W, Cin, Cout;
  int32_t stride, rate;
  int32_t relu;
  const void* Ws;     /* optional [device]: epos_pack_pointwise_weights_split of the same
                       * [9*Cin][Cout] matrix (NULL = fp32-MFMA kernel) */
  /* ---- ABI 5, as in EposPointwiseArgs */
  const void* Wh;     /* optional [device]: epos_pack_pointwise_weights_h2 of that matrix */
  const uint32_t* x_amax;   /* optional [device]: slot bounding max|X| */
  uint32_t* y_amax;   /* optional [device]: receives max|Y| */
} EposConv3x3Args;
int epos_conv3x3_f32(const EposConv3x3Args* args, void* stream);

/* Depthwise 3x3 conv + folded BatchNorm (+ optional ReLU before and after):
 * the depthwise half of net_xception.py:167-182 / model.py:80-88. stride 1 ->
 * TF 'SAME' (zero pad `rate`); stride 2 -> fixed_padding (net_xception.py:74-93)
 * + VALID, i.e. zero pad 1 before / 1 after with rate 1. w9c [device]: [9][C]
 * tap-major, BN scale folded in; bias [device]: [C]. C % 4 == 0. */
typedef struct EposDepthwiseArgs {
  const float* X; int64_t ldx;     /* [B, Hi, Wi] pixels, rows of ldx floats */
  const float* w9c; const float* bias;
  float* Y; int64_t ldy;           /* [B, Ho, Wo] pixels */
  int32_t B, Hi, Wi, Ho, Wo, C;
  int32_t stride, rate;
  int32_t relu_in, relu_out;
  /* ---- ABI 5: fp16-pair OUTPUT for the fp16-pair GEMM that consumes Y (its A operand).
   * y_h2 != 0: instead of 4 fp32 values per 16 bytes the kernel writes, for the same 4
   * channels c..c+3, [hi(c) hi(c+1) hi(c+2) hi(c+3) | mid(c) .. mid(c+3)] (8 fp16 = the same
   * 16 bytes, same addresses, same ldy): hi = rn_fp16(y * s), mid = rn_fp16((y * s - hi) *
   * 2^11), s = the power of two that puts the BOUND  gain * max(x_amax, x_amax2) + bias0
   * into [2^14, 2^15). The bound must hold for |Y| (gain = max_c sum_taps |w|, bias0 =
   * max_c |bias| with x_amax bounding |X|); the GEMM (EposPointwiseArgs.a_presplit, same
   * slots / gain / bias) derives the same s. Each activation is then split ONCE instead of
   * once per column tile of the GEMM, and the GEMM's loop carries no conversion at all. */
  int32_t y_h2;
  const uint32_t* x_amax;
  const uint32_t* x_amax2;   /* optional second slot */
  float gain, bias0;
} EposDepthwiseArgs;
int epos_depthwise3x3_f32(const EposDepthwiseArgs* args, void* stream);

/* Separable conv as ONE call: depthwise 3x3 (+BN, ReLU before/after) followed by the
 * pointwise 1x1 (+BN, +residual, +ReLU) -- slim.separable_conv2d as net_xception.py:96-194
 * (separable_conv2d_same, stride 1) and model.py:58-97 (split_separable_conv2d) build it.
 * Requirements: pw.A == dw.Y, pw.lda == dw.ldy, pw.K == dw.C, pw.M == dw.B * dw.Ho * dw.Wo.
 * dw.Y [device] keeps the depthwise output (fp32, or fp16 pairs with dw.y_h2 +
 * pw.a_presplit). Since ABI 7 this is exactly epos_depthwise3x3_f32(&dw) followed by
 * epos_pointwise_conv_f32(&pw) on the stream: the single-launch forms of rounds 2 and 4
 * (depthwise as a producer phase of the GEMM's workgroups, on the bf16 x 6 and on the
 * fp16-pair kernel; both bit-identical to the two launches) measured slower on every
 * configuration (C2 351 vs 420, C3 359 vs 425 images/s; profiles/r02, r04, r05) and left the
 * product library in round 5 -- DESIGN.md (e), git history up to commit 1b05025. */
/* The fp16-pair GEMM takes 128 x 64 instead of 128 x 128 output tiles for a launch with at
 * most `max_tiles` 128 x 128 tiles (default 100; environment EPOS_H2_BN64_MAX_TILES; 0 =
 * never): launches that would leave most CUs idle (ASPP 1x1 of one image: 76 tiles) get
 * twice the workgroups. Results do not depend on the tile. Returns the previous limit.
 * Process-wide; meant for tuning and for the tests that run both tiles. */
int epos_set_h2_narrow_tile_limit(int max_tiles);
typedef struct EposSepConvArgs {
  EposDepthwiseArgs dw;
  EposPointwiseArgs pw;
} EposSepConvArgs;
int epos_separable_conv_f32(const EposSepConvArgs* args, void* stream);

/* im2col for a dense 3x3 conv (slim resnet_utils.conv2d_same,
 * external/slim/nets/resnet_utils.py:77-122, used at net_xception.py:460-463):
 * col[m, (ky*3+kx)*C + c] = X[b, y*stride - pad + ky*rate, x*stride - pad + kx*rate, c]
 * (zero outside), columns zero-padded to ldcol. With preprocess != 0 the input
 * is first mapped x -> x*(2/255) - 1 (feature.py:171-174) for in-bounds taps. */
typedef struct EposIm2colArgs {
  const float* X; int64_t ldx;
  float* col; int64_t ldcol;
  int32_t B, Hi, Wi, Ho, Wo, C;
  int32_t stride, rate, pad;
  int32_t preprocess;
  /* ---- ABI 6: optional [device] absmax slot table to zero in the same launch (amax_words
   * uint32 words; must not exceed the launch's thread count = B*Ho*Wo*ldcol): a plan whose
   * first launch is this one needs no epos_amax_clear launch of its own */
  uint32_t* amax_clear;
  int64_t amax_words;
} EposIm2colArgs;
int epos_im2col3x3_f32(const EposIm2colArgs* args, void* stream);

/* Per-image channel means from the 32-row block sums an fp16-pair GEMM wrote
 * (EposPointwiseArgs.col_sums): Y[b, c] = (sum over the `blocks` blocks of image b) / hw. */
int epos_global_avg_pool_partial_f32(const float* P, int64_t ldp, float* Y, int32_t B,
                                     int32_t blocks, int32_t C, int32_t hw, void* stream);

/* Global mean over H*W (model.py:220): X [B, HW, C] (ldx) -> Y [B, C]. */
int epos_global_avg_pool_f32(const float* X, int64_t ldx, float* Y, int B,
                             int HW, int C, void* stream);

/* Bilinear resize, align_corners=True (misc.py:94-107 -> tf.image.resize_bilinear):
 * X [B, Hi, Wi, C] (ldx) -> Y [B, Ho, Wo, C] (ldy). Hi = Wi = 1 broadcasts
 * (image-pooling branch, model.py:225-226). C % 4 == 0. */
int epos_resize_bilinear_f32(const float* X, int64_t ldx, float* Y, int64_t ldy,
                             int B, int Hi, int Wi, int Ho, int Wo, int C,
                             void* stream);

/* ResNet-v1-beta backbone helpers (BASELINE config C5). max pool 3x3 stride 2,
 * TF 'SAME' (slim.max_pool2d at net_resnet_v1_beta.py:190): X [B,Hi,Wi,C] ->
 * Y [B,ceil(Hi/2),ceil(Wi/2),C]. */
int epos_maxpool3x3_s2_f32(const float* X, int64_t ldx, float* Y, int64_t ldy,
                           int B, int Hi, int Wi, int C, void* stream);
/* slim resnet_utils.subsample (external/slim/nets/resnet_utils.py:59-74): keeps
 * every factor-th pixel, Y [B,(Hi-1)/f+1,(Wi-1)/f+1,C]. */
int epos_subsample_f32(const float* X, int64_t ldx, float* Y, int64_t ldy, int B,
                       int Hi, int Wi, int C, int factor, void* stream);
/* Y = relu(A + B) over n contiguous floats (net_resnet_v1_beta.py:86 when the
 * pre-sum conv3 output is itself an end point, feature.py:50-54). */
int epos_add_relu_f32(const float* A, const float* B, float* Y, int64_t n,
                      void* stream);

/* Sparse update: for b < n_blocks, dst[offsets[b] .. offsets[b] + width) = src[b * width ..]
 * (all pointers [device]; offsets in elements; blocks must not overlap). Synthetic-workload
 * support: bench.py --planted-poses overwrites the head values of the target objects with
 * values rendered from known poses, between the network and the correspondence stage. */
int epos_scatter_blocks_f32(float* dst, const int64_t* offsets, const float* src,
                            int64_t n_blocks, int width, void* stream);

/* uint8 -> float32, n values (both pointers [device], 16-byte aligned): the
 * tf.cast(decode_image(...), tf.float32) of the reference's input pipeline
 * (datagen.py:435-436) done on the device, so that decoded frames are uploaded as bytes.
 * Exact. */
int epos_u8_to_f32(const uint8_t* X, float* Y, int64_t n, void* stream);

/* In-place softmax over groups of `G` consecutive floats (model.py:677-678):
 * X holds n_groups * G floats, group g at X + g*G (G <= 64). */
int epos_softmax_groups_f32(float* X, int64_t n_groups, int G, void* stream);

/* Per-pixel argmax over C channels -> int64 label (model.py:683); first maximum
 * wins (tf.argmax / np.argmax tie rule). */
int epos_argmax_i64(const float* X, int64_t ldx, int64_t* labels, int64_t P,
                    int C, void* stream);

/* ------------------------------------------------------------------------- *
 * Correspondence extraction (replaces epos_lib/corresp.py:9-101,
 * establish_many_to_many, and misc.py:14-26).
 * One "slot" = one (image, object) pair to extract.
 * ------------------------------------------------------------------------- */
typedef struct EposCorrSlot {
  int32_t image;     /* image index in the batch */
  int32_t obj_id;    /* 1-based object id: channel obj_id of obj_confs,
                        channel obj_id-1 of the fragment heads (corresp.py:46,60) */
} EposCorrSlot;

/* Fragment-confidence softmax (model.py:678) for the given slots only: X is the
 * dense frag_conf buffer f32 [B, P, O, F]; slots [device]. Used by the sparse-head
 * mode, where the fragment heads exist only for the target objects. */
int epos_softmax_slots_f32(float* X, const EposCorrSlot* slots, int S, int P,
                           int O, int F, void* stream);

/* Pass 1+2: per slot, count masked pixels and correspondences and compute the
 * raster-order exclusive offsets. All buffers [device].
 *   obj_confs  f32 [B, P, O+1]      (P = h*w pixels of the head map)
 *   frag_confs f32 [B, P, O, F]     (F == 64)
 *   px_off, corr_off  i32 [S, P]    scratch/outputs (exclusive scans)
 *   frag_mask  u64 [S, P]           kept-fragment bitmask per pixel (0 = not masked)
 *   totals     i32 [S, 2]           {masked pixels, correspondences} per slot
 */
int epos_corr_count(const float* obj_confs, const float* frag_confs,
                    const EposCorrSlot* slots /*[device]*/, int S, int B, int P,
                    int O, int F, float min_obj_conf, float min_frag_rel_conf,
                    int32_t* px_off, int32_t* corr_off, uint64_t* frag_mask,
                    int32_t* totals, void* stream);

/* Pass 3: fills the correspondence arrays. slot_base i64[S] [device] gives each
 * slot's first row in the pooled output arrays (exclusive scan of totals[:,1],
 * computed by epos_corr_slot_bases or by the host); `capacity` rows are
 * available; rows beyond it are not written and *overflow is set to 1.
 *   frag_coords f32 [B, P, O, F, 3]
 *   frag_centers f64 [O, F, 3], frag_sizes f64 [O, F]  (model store, obj_id-1 major)
 * Outputs [device]: px_id i64[N], frag_id i64[N], coord_2d f64[N,2],
 * coord_3d f64[N,3], conf/conf_obj/conf_frag f32[N] with the exact arithmetic of
 * corresp.py:55-57,71-78,82-84 (see oracle/corresp_ref.py). W = head map width,
 * inv_scale = 1 / output_scale. */
typedef struct EposCorrOut {
  int64_t* px_id; int64_t* frag_id;
  double* coord_2d; double* coord_3d;
  float* conf; float* conf_obj; float* conf_frag;
} EposCorrOut;
int epos_corr_fill(const float* obj_confs, const float* frag_confs,
                   const float* frag_coords, const double* frag_centers,
                   const double* frag_sizes, const EposCorrSlot* slots, int S,
                   int B, int P, int W, int O, int F, double inv_scale,
                   const int32_t* px_off, const int32_t* corr_off,
                   const uint64_t* frag_mask, const int64_t* slot_base,
                   int64_t capacity, const EposCorrOut* out, int32_t* overflow,
                   void* stream);

/* slot_base[s] = sum_{t<s} totals[t][1]; slot_base[S] = grand total. [device] */
int epos_corr_slot_bases(const int32_t* totals, int S, int64_t* slot_base,
                         void* stream);

/* project_to_surface (corresp.py:87-88, datagen.py:128-154: igl::AABB::squared_distance):
 * out[i] = the point of the triangle mesh (verts [nv,3] f64, faces [nf,3] int32, all
 * device) closest to pts[i] ([n,3] f64); face_idx [n] int32 or NULL receives the face.
 * Exact sweep over all faces, ties -> lowest face index. */
int epos_project_to_mesh_f64(const double* pts, int64_t n, const double* verts,
                             int64_t nv, const int32_t* faces, int64_t nf, double* out,
                             int32_t* face_idx, void* stream);

/* ------------------------------------------------------------------------- *
 * Model preprocessing (replaces epos_lib/fragment.py:8-54, fragmentation_fps,
 * called once per object by ObjectModelStore.fragment_models, datagen.py:86-126).
 * vertices f64[V,3] [device]; outputs [device]: centers f64[F,3] (the FPS picks, in
 * order), center_idx i32[F] (their vertex indices), vertex_frag_ids i32[V] (nearest
 * centre of every vertex); nn_dist f64[V] is scratch. Bit-identical to the
 * reference (fp64, same operation order, lowest-index tie rule).
 * ------------------------------------------------------------------------- */
int epos_fragmentation_fps(const double* vertices, int64_t V, int num_frags,
                           double* nn_dist, double* centers, int32_t* center_idx,
                           int32_t* vertex_frag_ids, void* stream);

/* ------------------------------------------------------------------------- *
 * Pose fitting (replaces pyprogressivex.find6DPoses, scripts/infer.py:470-488;
 * the un-vendored danini/progressive-x pybind11 module).
 * ------------------------------------------------------------------------- */
typedef struct EposFitParams {
  double threshold;                 /* inlier_thresh, tau_r [px]   (infer.py:76-79)  */
  double neighborhood_ball_radius;  /* neighbour_max_dist, tau_d: two correspondences
                                     * are neighbours iff their distance in
                                     * (x, y, s X, s Y, s Z) is <= tau_d (infer.py:80-82) */
  double spatial_coherence_weight;  /* lambda of GC-RANSAC's labelling energy (0 = off) */
  double scaling_from_millimeters;  /* s above (0.1: mm -> cm, infer.py:106-110)     */
  double max_tanimoto_similarity;   /* infer.py:112-114                              */
  double conf;                      /* required_progx_confidence: a failed proposal of a
                                     * multi-instance search is retried until the samples
                                     * drawn reach this confidence (max two extra rounds) */
  double proposal_engine_conf;      /* required_ransac_confidence: RANSAC stops once
                                     * (1 - w^3)^it <= 1 - this (1.0 = always max_iters) */
  double min_coverage;              /* min_hypothesis_quality, tau_q                 */
  double min_triangle_area;         /* tau_t                                         */
  int32_t max_iters;                /* max_fitting_iterations (400)                  */
  int32_t min_point_number;         /* 6 (infer.py:483)                              */
  int32_t max_model_number;         /* num_instances; -1 = as many as found          */
  int32_t max_model_number_for_optimization;  /* joint refinement only up to this many
                                     * instances (max_model_number_for_pearl)        */
  int32_t use_prosac;               /* sample from a growing confidence-sorted prefix */
  int32_t lo_iters;                 /* Gauss-Newton refits per local-optimisation stage (8) */
  int32_t gc_sweeps;                /* relabelling sweeps of the spatial-coherence step
                                     * (default 2; 0 = thresholded inliers only)     */
  int32_t pearl_iters;              /* joint refinement iterations of multi-instance
                                     * results (default 2; 0 = off)                  */
} EposFitParams;
void epos_fit_params_default(EposFitParams* p);

/* Host-pointer drop-in for pyprogressivex.find6DPoses: xy f64[n,2] (pixels),
 * xyz f64[n,3] (mm), K f64[9] row-major. Outputs (caller-owned):
 *   poses  f64[max_k*12]: instance j occupies poses[12*j .. 12*j+11] as
 *          R row-major in [0..8] (R[r][c] = poses[12*j + 3*r + c]) followed by
 *          t in [9..11] -- NOT a row-major 3x4 [R|t]: reading the 12 doubles as a 3x4
 *          matrix gives wrong poses. (The reference's module returns the instances
 *          stacked as a [3k,4] array which infer.py:490-495 slices into R, t; the binding
 *          in INTEGRATION.md rebuilds exactly that from this layout.)
 *   labels i32[n] (instance index or -1), scores f64[max_k].
 * Returns k >= 0 (0 <=> the reference's `pose_ests is None`) or < 0. */
int epos_find6d_poses(const double* xy, const double* xyz, int64_t n,
                      const double* K, const EposFitParams* p, uint64_t seed,
                      double* poses_out, int32_t* labels_out, double* scores_out,
                      int32_t max_k);

/* Batched device entry: S independent fitting problems (slots) in one call, all
 * buffers [device], nothing synchronises.
 *   xy f64[N,2], xyz f64[N,3] pooled; slot s owns rows [slot_base[s], slot_base[s+1])
 *   Ks f64[S,9]; max_models i32[S] (per-slot num_instances, <= max_k)
 *   seeds u64[S]
 *   work: scratch of epos_fit_workspace_bytes(S, N_capacity, p) bytes
 * Outputs: poses f64[S,max_k,12] (per instance: R row-major [0..8], then t [9..11], as
 *          above), scores f64[S,max_k], num_models i32[S], labels i32[N] (instance index
 *          within the slot or -1). num_models[s] == -1: a device-side consistency check
 *          failed for that slot (a hand-off between the workgroups that fit it together
 *          timed out); its outputs are not valid and the call should be repeated --
 *          epos_find6d_poses returns EPOS_E_INTERNAL in that case.
 * PRECONDITION (not checked): the rows of every slot are in image-row order --
 * xy[.,1] non-decreasing within [slot_base[s], slot_base[s+1]) -- which is the raster order
 * epos_corr_fill writes. The spatial-coherence sweeps and the joint refinement find a
 * point's neighbours in a window of rows around it and stop at the first row farther than
 * neighborhood_ball_radius: with unsorted rows neighbourhoods are silently truncated and
 * the poses change. Callers with arbitrary row order use epos_find6d_poses (it sorts
 * internally and keeps the caller's order for PROSAC) or sort first; with
 * spatial_coherence_weight == 0 (or gc_sweeps == 0) and max_k == 1 no order is needed. */
int64_t epos_fit_workspace_bytes(int S, int64_t n_capacity, const EposFitParams* p,
                                 int32_t max_k);
int epos_find6d_poses_device(const double* xy, const double* xyz,
                             const int64_t* slot_base, int S, int64_t n_capacity,
                             const double* Ks, const int32_t* max_models,
                             const uint64_t* seeds, const EposFitParams* p,
                             int32_t max_k, void* work, double* poses,
                             double* scores, int32_t* num_models, int32_t* labels,
                             void* stream);

/* ------------------------------------------------------------------------- *
 * The OpenCV fitting method (replaces cv2.solvePnPRansac(objectPoints, imagePoints, K,
 * None, iterationsCount, reprojectionError, confidence=0.99, flags=cv2.SOLVEPNP_EPNP),
 * scripts/infer.py:505-528; OpenCV 3.4.2, README.md:29, is not vendored).
 * What is computed (the published behaviour of that call, restated):
 *   - points are rounded to float32 on entry; minimal sets of 5 correspondences drawn by
 *     cv::RNG (state 2^64 - 1, `next() % n`, repeated indices re-drawn);
 *   - EPnP (Lepetit et al., IJCV 2009) on each set -> one pose; its inliers: squared
 *     reprojection error, float32 arithmetic on the float32 projection, <= (float)err^2;
 *   - a set with strictly more inliers than the best so far (and > 4) becomes the best and
 *     the iteration cap becomes round(log(1 - confidence) / log(1 - w^5));
 *   - EPnP once more over ALL inliers of the best set: that pose is returned.
 * Own factorisations (cyclic Jacobi, Householder QR) instead of cv::SVD: poses agree with
 * an LAPACK-based statement of EPnP to ~1e-9, not bit for bit with OpenCV -- unpinned.
 * ------------------------------------------------------------------------- */
typedef struct EposPnpRansacParams {
  int32_t iterations_count;     /* max_fitting_iterations (400)       infer.py:515 */
  int32_t min_point_number;     /* slots / calls with fewer correspondences give no pose:
                                 * 6 in the script (infer.py:420-422 skips them before the
                                 * call); 0 = OpenCV's own rule (n >= 5)                */
  double reprojection_error;    /* inlier_thresh (4.0 px)             infer.py:516 */
  double confidence;            /* 0.99                               infer.py:517 */
} EposPnpRansacParams;
void epos_pnp_ransac_params_default(EposPnpRansacParams* p);

/* Host-pointer drop-in: xy f64[n,2] (imagePoints), xyz f64[n,3] (objectPoints), K f64[9]
 * row-major (fx, fy, cx, cy are used, as by OpenCV without distortion). Outputs
 * (caller-owned): pose_out f64[12] = R row-major in [0..8], then t in [9..11] (the layout of
 * epos_find6d_poses, NOT a row-major 3x4; R = Rodrigues(rvec) of the
 * reference), inlier_mask_out u8[n], info_out i32[4] or NULL = {index of the winning set,
 * its inlier count, the final iteration cap, sets evaluated}. Returns 1 (pose found),
 * 0 (`pose_est_success` false) or < 0. */
int epos_solve_pnp_ransac(const double* xy, const double* xyz, int64_t n, const double* K,
                          const EposPnpRansacParams* p, double* pose_out,
                          uint8_t* inlier_mask_out, int32_t* info_out);

/* Batched device entry (slots as for epos_find6d_poses_device), all buffers [device]:
 * poses f64[S,12] (R row-major [0..8], t [9..11]), success i32[S], inlier_mask u8[N],
 * info i32[S,4] or NULL;
 * work = epos_pnp_ransac_workspace_bytes(S, N_capacity, p) bytes. Nothing synchronises. */
int64_t epos_pnp_ransac_workspace_bytes(int S, int64_t n_capacity,
                                        const EposPnpRansacParams* p);
int epos_solve_pnp_ransac_device(const double* xy, const double* xyz,
                                 const int64_t* slot_base, int S, int64_t n_capacity,
                                 const double* Ks, const EposPnpRansacParams* p, void* work,
                                 double* poses, int32_t* success, uint8_t* inlier_mask,
                                 int32_t* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* EPOS_HIP_H_ */
