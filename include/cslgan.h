/*
 * cslgan.h — C-ABI of libcslgan_hip.so: the MI355X (gfx950) kernels behind the csl-gan
 * DP discriminator step.
 *
 * The reference (twosixlabs/csl-gan) is pure Python and has no FFI layer: the boundary a
 * maintainer binds is the set of PyTorch / Opacus-fork calls its D-step makes.  Every entry
 * point below names the reference call (file:line under /root/reference) whose device work
 * it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C types only: device pointers are void* / float*, sizes are int / int64_t,
 *     the stream is the raw hipStream_t handle passed as void* (0 = default stream);
 *   - every function returns 0 on success or a negative cslgan_status; the message of the
 *     last failure on the calling thread is returned by cslgan_last_error();
 *   - all work is enqueued asynchronously on the given stream; the library never allocates,
 *     frees or synchronises (graph-capturable).  Workspaces are caller-owned;
 *   - activations are NHWC fp32 ("channels last"): x[n][h][w][c]; conv weights are
 *     KRSC: w[cout][kh][kw][cin] (the memory of a torch channels_last [cout,cin,kh,kw] tensor).
 */
#ifndef CSLGAN_H
#define CSLGAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSLGAN_ABI_VERSION 6

typedef enum {
    CSLGAN_OK = 0,
    CSLGAN_ERR_INVALID_ARG = -1,   /* bad shape / null pointer / unsupported configuration */
    CSLGAN_ERR_LAUNCH = -2,        /* hipLaunchKernel / hipMemsetAsync reported an error */
    CSLGAN_ERR_NO_DEVICE = -3      /* no gfx950 device visible */
} cslgan_status;

/* activation / epilogue codes for cslgan_conv2d_* */
enum { CSLGAN_ACT_NONE = 0, CSLGAN_ACT_LRELU02 = 1, CSLGAN_ACT_RELU = 2, CSLGAN_ACT_TANH = 3 };

#define CSLGAN_MAX_SEGS 16
#define CSLGAN_MAX_ROLES 8      /* row blocks of one fused critic pass (adaptive / generated / private / penalty rows) */

/* A batch of row-major [n_rows, len[i]] matrices (one per parameter tensor): the materialised
 * per-sample gradients p.grad_sample viewed as [passes*B, numel(p)]  (train.py:233,311-315). */
typedef struct {
    int32_t n_seg;
    int32_t _pad;
    const float* in[CSLGAN_MAX_SEGS];      /* device, [n_rows, len] with row stride row_stride */
    float* out[CSLGAN_MAX_SEGS];           /* device, [len]            (clip_accum only)       */
    const float* noise[CSLGAN_MAX_SEGS];   /* device, [len] pre-drawn N(0,1) or NULL -> Philox  */
    int64_t len[CSLGAN_MAX_SEGS];
    int64_t row_stride[CSLGAN_MAX_SEGS];   /* in elements */
    int64_t rows[CSLGAN_MAX_SEGS];         /* clip_accum only: > 0 = this segment has its OWN row count (column sums of slab sets of
                                            * different heights in one launch; needs factors == NULL), 0 = the call's n_rows */
    const uint64_t* call_counter;          /* device, nullable (clip_accum only): the Philox offset used is
                                            * offset + 64 * *call_counter — a step captured in a HIP graph passes its
                                            * noise-call counter here instead of by value, so every replay draws new noise */
} cslgan_segs_t;

int cslgan_version(void);
const char* cslgan_last_error(void);
/* Name of the device kernel most recently launched by the calling thread through a conv / clip / norm entry, as rocprofv3
 * --kernel-trace lists it (e.g. "igemm_halo_kernel<128,false>"); "" before the first launch.  Measurement aid: bench.py
 * tags its per-launch HIP-event timings with it. */
const char* cslgan_last_kernel(void);
/* number of visible HIP devices (>=0) or a negative status */
int cslgan_device_count(void);

/* ---- per-sample norm / clip / noise  (the fused DP kernel family) ------------------------ */

/* out_sq[s * n_rows + r] = sum_j in[s][r][j]^2.  Replaces
 * opacus.utils.tensor_utils.calc_sample_norms (train.py:311-314) and the norm half of
 * privacy_engine.clip() (train.py:399).  out_sq is overwritten (zeroed on-stream first). */
int cslgan_sample_sqnorm_f32(const cslgan_segs_t* segs, int64_t n_rows, float* out_sq, void* stream);

/* bfloat16 storage of the materialised per-sample gradients (fp32 accumulate): segs->in[] point to bf16
 * elements, everything else as the _f32 form.  Halves the HBM bytes of the norm / clip passes
 * (17.3 MB/img/clipped pass for D64); BASELINE.json config 5. */
int cslgan_sample_sqnorm_bf16(const cslgan_segs_t* segs, int64_t n_rows, float* out_sq, void* stream);

/* Clip factors from squared norms:  f = min(1, C / (sqrt(sq) + eps)).
 *   flat != 0 : one factor per row from the all-segment norm; max_norm[0] is C; out_f is [n_rows]
 *   flat == 0 : per segment; max_norm[s] is C_s; out_f is [n_seg, n_rows]
 * rows < first_private_row get factor 1 (split mode: the generated-data pass is not clipped,
 * train.py:112-113).  out_norm (nullable) receives the norms ([1 or n_seg, n_rows]).
 * Replaces norm_clipper.calc_clipping_factors (train.py:324). max_norm is a DEVICE pointer. */
int cslgan_clip_factors_f32(const float* sq, int n_seg, int64_t n_rows, const float* max_norm, int flat,
                            float eps, int64_t first_private_row, float* out_f, float* out_norm, void* stream);

/* out[s][j] = beta*out[s][j] + scale * ( sum_r f(s,r) * in[s][r][j] + noise_std[s] * z[s][j] )
 *   factors: device [n_rows] (factors_per_seg==0) or [n_seg,n_rows] (factors_per_seg!=0), NULL -> 1
 *   noise_std: DEVICE [n_seg] or NULL (no noise); z from segs->noise[s] when given, else
 *   Philox4x32-10 + Box-Muller keyed by (seed, offset, s, j).
 * Replaces privacy_engine.clip() + accum_grads_across_passes() (train.py:399-402) and, with
 * noise/scale, the engine-wrapped d_optimizer.step() noise + 1/B (train.py:484). */
int cslgan_clip_accum_noise_f32(const cslgan_segs_t* segs, int64_t n_rows, const float* factors,
                                int factors_per_seg, const float* noise_std, uint64_t seed, uint64_t offset,
                                float scale, float beta, void* stream);

int cslgan_clip_accum_noise_bf16(const cslgan_segs_t* segs, int64_t n_rows, const float* factors,
                                 int factors_per_seg, const float* noise_std, uint64_t seed, uint64_t offset,
                                 float scale, float beta, void* stream);

/* backprop_clip.py:18-22 l2_clip: rows with ||t_r|| > C are scaled to norm C (no epsilon).
 * in/out may alias.  norms_ws: caller workspace [n_rows] floats. */
int cslgan_l2_clip_rows_f32(const float* in, float* out, int64_t n_rows, int64_t len, float C,
                            float* norms_ws, void* stream);

/* MeanSampler.sample (mean_sampler.py:75-84; called from train.py:200-202, 214-216) on device-resident mean samples:
 *   out[i][:] = mean_samples[labels[i]][perms[i]][:] + noise_mean_std * z_i + noise_std * z_{i,:}
 * mean_samples [n_classes][num_samples][len], labels (nullable when n_classes == 1) and perms [n] int64, out [n][len];
 * z ~ N(0,1) from Philox4x32-10 keyed (seed, offset): the caller advances offset per call.
 * perms == NULL (num_samples <= 1024): the kernel draws them — image i takes entry i mod num_samples of the (i / num_samples)-th
 * uniform random permutation (torch.cat of randperms, mean_sampler.py:76).  labels == NULL with n_classes > 1: uniform labels
 * are drawn (mean_sampler.py:77).  labels_out (nullable, [n] int64) receives the labels used. */
int cslgan_mean_sample_f32(const float* mean_samples, int n_classes, int num_samples, int64_t len, const int64_t* labels,
                           const int64_t* perms, int64_t n, float noise_mean_std, float noise_std, uint64_t seed, uint64_t offset,
                           float* out, int64_t* labels_out, void* stream);

/* Row L2 norms of a [n_rows, len] matrix (gradient_penalty.py:52-53) and the backward of
 * norm: gin[r][j] = gnorm[r] * in[r][j] / norm[r]. */
int cslgan_row_l2norm_f32(const float* in, int64_t n_rows, int64_t len, float* out_norm, void* stream);
int cslgan_row_l2norm_bwd_f32(const float* in, const float* norm, const float* gnorm, int64_t n_rows,
                              int64_t len, float* gin, void* stream);

/* ---- convolution family: fp32 MFMA implicit GEMM ------------------------------------------ */

/* cslgan_conv_t.compute.  All tensors are fp32 in HBM either way.
 *   CSLGAN_COMPUTE_F32 : v_mfma_f32_32x32x2_f32 — bit-for-bit an fp32 fmaf chain; the default and the headline path.
 *   CSLGAN_COMPUTE_BF16: operands rounded to bfloat16 (round-to-nearest-even) on their way into LDS, v_mfma_f32_32x32x16_bf16
 *                        with fp32 accumulate (BASELINE.json configs[4]; `--compute_dtype bf16`).  Honoured by conv2d_fwd,
 *                        conv2d_s2_fwd, conv2d_dgrad, conv2d_wgrad_grouped(_bf16out), conv2d_wgrad_scaled; the vector-ALU
 *                        (1..4 channel) kernels, the RGB first layer (3 -> 64, 5x5, stride 2), single-output linear layers and the
 *                        Gram-norm entries compute in fp32 regardless (a fraction of a per cent of the FLOP).
 *   CSLGAN_COMPUTE_BF16X3: fp32 emulated on the bf16 matrix cores — every operand is split into three bfloat16 pieces
 *                        (x = hi + mid + lo, 3 x 8 mantissa bits = fp32's 24), a product is the sum of the six largest of the nine
 *                        piece products (each exact in fp32), fp32 accumulate: per-product error about one fp32 ulp at 2.67x the
 *                        fp32 MFMA rate.  Same entries as BF16. */
enum { CSLGAN_COMPUTE_F32 = 0, CSLGAN_COMPUTE_BF16 = 1, CSLGAN_COMPUTE_BF16X3 = 2 };

typedef struct {
    int32_t N, H, W, C;          /* input  x[N][H][W][C]                                     */
    int32_t K, R, S;             /* filter w[K][R][S][C]                                     */
    int32_t stride, pad;
    int32_t compute;             /* CSLGAN_COMPUTE_F32 (exact fp32 MFMA) or CSLGAN_COMPUTE_BF16 (see below) */
    int32_t P, Q;                /* output y[N][P][Q][K]                                     */
    /* ABI v6: optional device scratch for launches that would leave most of the chip idle (the critic's last convs: 64-384
     * workgroups on 256 CUs).  When given, the LDS-halo kernel of csrc/igemm_x3.hip may divide the REDUCTION channels over up to
     * split_ws_floats / (output floats) workgroups per output tile, each writing its partial sums here, and a second launch adds
     * the partials in a fixed order and applies bias / activation / mask (deterministic: no atomics).  NULL / 0: never split. */
    void* split_ws;
    int64_t split_ws_floats;
    /* ABI v6: GroupNorm statistics from the conv's epilogue (cslgan_conv2d_fwd_x3_f32 only).  When gn_part is given, every
     * workgroup of the halo kernel also leaves, for each of its 64-row patches and each of the gn_groups channel groups it covers,
     * the pair (sum, centred sum of squares about the patch's own mean) of the values it stores:
     *   gn_part[((n * (P*Q/64) + patch) * gn_groups + g) * 2 + {0, 1}]
     * cslgan_groupnorm_apply_parts_f32 combines the P*Q/64 pairs of an image exactly (Chan et al.) — the statistics pass over the
     * activation (one launch and one read of the tensor per normalisation) is gone.  Needs stride 1, P % 8 == Q % 8 == 0, act ==
     * NONE, K % gn_groups == 0 with K / gn_groups in {1, 2, 4, 8, 16, 32} and a shape the halo kernel takes; anything else is an
     * error (the caller then normalises with cslgan_groupnorm_act_f32). */
    float* gn_part;
    int32_t gn_groups;
    int32_t in_relu;
    /* ABI v6: a per-(sample, channel) affine map (+ ReLU when in_relu != 0) applied to x while it is staged —
     *   x'[n][h][w][c] = max(in_scale[n*C + c] * x[n][h][w][c] + in_shift[n*C + c], in_relu ? 0 : -inf), zero padding AFTER the map —
     * i.e. GroupNorm + ReLU folded into the consuming conv (cslgan_groupnorm_affine_parts_f32 makes the two tables from the
     * producing conv's statistics): the normalised activation is never written to HBM.  Taken by cslgan_conv2d_fwd_x3_f32 on the
     * shapes the LDS-halo kernel runs and by the 1..4-output-channel kernel behind cslgan_conv2d_fwd_f32 (64 input channels);
     * anything else is an error.  NULL: x is used as it is. */
    const float* in_scale;
    const float* in_shift;
} cslgan_conv_t;

/* y = act(conv(x, w) + bias [+ residual]).  Replaces torch.nn.Conv2d / nn.Linear forward at
 * DCResNet_models.py:131-132,145 (D), :16 (the conv inside UpsampleConv, run on the depth-to-space tensor with
 * channel-folded filters, see cslgan_depth_to_space_f32), :26,:36,:85,:95-104 (G), MNIST_models.py:17-23,41-46.
 *   bias      : [K] or NULL
 *   residual  : NULL or r[N][P][Q][K], added before act (ResBlockUp's "o + s", DCResNet_models.py:38)
 * A Linear layer is the 1x1 case H=W=P=Q=R=S=1. */
int cslgan_conv2d_fwd_f32(const cslgan_conv_t* p, const float* x, const float* w, const float* bias,
                          const float* residual, int act, float* y, void* stream);

/* cslgan_conv2d_fwd_f32 with compute == CSLGAN_COMPUTE_BF16X3 and the filter pre-split into its three bfloat16 pieces
 * (or compute == CSLGAN_COMPUTE_BF16 and the filter pre-rounded: one piece, the same layout):
 * w3_ws is a caller workspace of 3 * K*R*S*C bfloat16 (= 1.5 * K*R*S*C floats), rebuilt from w when repack != 0 and reused
 * otherwise (the caller knows when w changed).  Stride-1 convs on 8x8-patchable grids with C % 16 == 0 and K >= 64 then run on
 * the LDS-halo kernel whose filter operand goes straight from that workspace to registers; other shapes ignore it.
 * compute == CSLGAN_COMPUTE_F32 is accepted too (round 4): the workspace then holds an fp32 copy of the filter in the same
 * step-major order (K*R*S*C floats of the same allocation) and the kernel runs v_mfma_f32_32x32x2_f32 on the same staging —
 * exact fp32 products, the result of cslgan_conv2d_fwd_f32 up to summation order. */
int cslgan_conv2d_fwd_x3_f32(const cslgan_conv_t* p, const float* x, const float* w, void* w3_ws, int repack,
                             const float* bias, const float* residual, int act, float* y, void* stream);
/* The workspace of cslgan_conv2d_fwd_x3_f32 on its own (ABI v5): w[rows][taps][red] fp32 -> w3_ws[pieces][(red/16)*taps + tap][rows][16]
 * bfloat16 (pieces = 3: hi / mid / lo; 1: the rounded filter; 0: an fp32 copy, [step][rows][16] floats).  Lets a caller that replays a recorded graph refresh the pieces of a
 * FROZEN filter in place only when the filter has changed, instead of inside every replay; red % 16 != 0 writes nothing. */
int cslgan_split_filter_x3_f32(const float* w, int rows, int taps, int red, void* w3_ws, int pieces, void* stream);

/* gx = conv_transpose(gy, w) [* lrelu'(mask)]: the data gradient (autograd of the conv above;
 * "conv_transpose2d" in the north star).  wt_ws: caller workspace of K*R*S*C floats receiving
 * the repacked filters (rebuilt when repack != 0, reused otherwise).  mask (nullable) has gx's shape:
 * gx *= (mask > 0 ? 1 : 0.2). */
int cslgan_conv2d_dgrad_f32(const cslgan_conv_t* p, const float* gy, const float* w, float* wt_ws, int repack,
                            const float* mask, float* gx, void* stream);

/* cslgan_conv2d_dgrad_f32 with compute == CSLGAN_COMPUTE_BF16X3 (fp32 from three bfloat16 pieces) or CSLGAN_COMPUTE_BF16 on the
 * LDS-halo kernel of csrc/igemm_x3.hip (ABI v5): w3_ws is a second caller workspace of 3 * K*R*S*C bfloat16 receiving the
 * repacked parity-class matrices split into their pieces in step-major order (rebuilt with wt_ws when repack != 0).  Takes
 * stride 1-2, K % 16 == 0, C >= 64, class grids 8x8-patchable or 4x4; other shapes run the gather kernels in the same
 * arithmetic and ignore w3_ws.  compute == CSLGAN_COMPUTE_F32: w3_ws receives fp32 step-major class matrices and the launch is the
 * exact-fp32 form of the same kernel (8x8-patchable class grids only).  Same call it replaces: the autograd data gradient of nn.Conv2d (DCResNet_models.py:131-132)
 * and its use inside the penalty's double backward (gradient_penalty.py:48-54). */
int cslgan_conv2d_dgrad_x3_f32(const cslgan_conv_t* p, const float* gy, const float* w, float* wt_ws, void* w3_ws, int repack,
                               const float* mask, float* gx, void* stream);

/* Grouped weight gradient:  gw[g][k][r][s][c] = alpha * sum_{n in group g} sum_{p,q} gy[n,p,q,k] x[n,..,c]
 * with groups of `group` consecutive samples (N % group == 0).
 *   group == 1 -> per-sample gradients p.grad_sample (Opacus hook, train.py:387)
 *   gw == NULL -> norms only ("ghost" mode)
 *   sq (nullable): [N/group] += sum of squares of alpha*gw[g]  (caller zeroes)
 * The dense gradient is group == N, or any group followed by cslgan_clip_accum_noise_f32. */
int cslgan_conv2d_wgrad_grouped_f32(const cslgan_conv_t* p, const float* gy, const float* x, int group,
                                    float alpha, float* gw, float* sq, void* stream);

/* Per-sample weight gradients of a batch made of consecutive ROW BLOCKS with their own outputs, in one launch — the fused
 * discriminator pass of train.py:204-245 + 382-389 carries the adaptive-clipping rows (norms only), the generated rows (only
 * their sum is used) and the private rows (materialised) in one batch, and three launches of 640 workgroups each fill
 * 512 slots 62 % where one launch of 1920 fills them 94 %.  Block b covers samples [block_first[b], block_first[b+1]) (the last
 * one ends at N); sample n of block b writes gw[b] + (n - block_first[b]) * K*R*S*C (nothing when gw[b] is NULL) and adds
 * ||alpha * g_n||^2 into sq[b][n - block_first[b]] (skipped when sq[b] is NULL; the caller zeroes).  gw and sq are HOST arrays of
 * n_blocks device pointers.  fp32 only, on the shapes igemm_wgh takes (stride 1-2, 2..5 filter columns, K % 64 == 0,
 * C % 64 == 0, P % 8 == 0, Q % 8 == 0); other shapes return CSLGAN_ERR_INVALID_ARG and the caller uses one
 * cslgan_conv2d_wgrad_grouped_f32 call per block. */
#define CSLGAN_MAX_WGRAD_BLOCKS 4
int cslgan_conv2d_wgrad_blocks_f32(const cslgan_conv_t* p, const float* gy, const float* x, float alpha, int n_blocks,
                                   const int32_t* block_first, float* const* gw, float* const* sq, void* stream);

/* Stride-2 forward conv (the critic's convs, DCResNet_models.py:131) through the LDS-halo kernel: the four parity
 * sub-images of x are convolved at stride 1 and accumulated in one workgroup.  wcls_ws: K*R*R*C floats receiving the
 * filters regrouped by parity class (rebuilt when repack != 0).  Same result as cslgan_conv2d_fwd_f32, to which it falls
 * back for shapes the halo kernel does not take (C % 32, K >= 64, 8x8-patchable or 4x4 output grid). */
int cslgan_conv2d_s2_fwd_f32(const cslgan_conv_t* p, const float* x, const float* w, float* wcls_ws, int repack,
                             const float* bias, int act, float* y, void* stream);

/* The same stride-2 forward conv with compute == CSLGAN_COMPUTE_BF16X3 / CSLGAN_COMPUTE_BF16 on the LDS-halo kernel of
 * csrc/igemm_x3.hip (ABI v5): w3_ws = 3 * K*R*R*C bfloat16 receiving the parity-class matrices split into their pieces
 * (rebuilt with wcls_ws when repack != 0).  C % 16 == 0, K >= 64, 8x8-patchable or 4x4 output grid; other shapes fall back
 * to cslgan_conv2d_fwd_f32 in the same arithmetic.  compute == CSLGAN_COMPUTE_F32: as for cslgan_conv2d_dgrad_x3_f32. */
int cslgan_conv2d_s2_fwd_x3_f32(const cslgan_conv_t* p, const float* x, const float* w, float* wcls_ws, void* w3_ws, int repack,
                                const float* bias, int act, float* y, void* stream);

/* Clip-weighted grouped weight gradient: as cslgan_conv2d_wgrad_grouped_f32 with gy of sample n multiplied by
 * row_scale[n] on load.  With row_scale = the per-sample clip factors f_b and group = N this is the clipped sum
 * sum_b f_b g_b (privacy_engine.clip() + accumulate, train.py:399-417) without materialising p.grad_sample. */
int cslgan_conv2d_wgrad_scaled_f32(const cslgan_conv_t* p, const float* gy, const float* x, const float* row_scale,
                                   int group, float alpha, float* gw, void* stream);

/* Dense weight gradient of a stride-1 conv with 1..4 output channels and 64 input channels (G's output conv in
 * train_G, train.py:502-511), on the vector ALU.  Each of the n_blocks workgroups writes one partial [K][R*S][64]
 * row into partial[n_blocks][K*R*S*64]; the caller sums the rows (cslgan_clip_accum_noise_f32).  Needs P, Q
 * multiples of 8 and at most 9 taps. */
int cslgan_conv2d_wgrad_skinny_f32(const cslgan_conv_t* p, const float* gy, const float* x, float alpha, float* partial,
                                   int n_blocks, void* stream);

/* Per-sample squared norms of the weight gradient WITHOUT forming it:  sq[n] += alpha^2 * sum_{p,p'}
 * (GY_n GY_n^T)[p,p'] (XU_n XU_n^T)[p,p']  — the same value cslgan_conv2d_wgrad_grouped_f32(group=1, gw=NULL)
 * accumulates (opacus calc_sample_norms, train.py:311-314), 30x fewer FLOP for the critic's last conv.
 * Needs P*Q <= 64, K % 32 == 0, C % 32 == 0.  sq: [N], caller zeroes. */
int cslgan_conv2d_wgrad_sqnorm_gram_f32(const cslgan_conv_t* p, const float* gy, const float* x, float alpha, float* sq,
                                        void* stream);

/* Same, storing gw as bfloat16 (round-to-nearest-even); sq is the norm of the ROUNDED values, i.e. of what
 * the clip kernels will read back. */
int cslgan_conv2d_wgrad_grouped_bf16out_f32(const cslgan_conv_t* p, const float* gy, const float* x, int group,
                                            float alpha, void* gw_bf16, float* sq, void* stream);

/* Per-group bias gradient gb[g][k] = alpha * sum_{n in g, p, q} gy[n,p,q,k]; sq as above. */
int cslgan_bias_grad_grouped_f32(const float* gy, int N, int PQ, int K, int group, float alpha,
                                 float* gb, float* sq, void* stream);

/* ---- bf16 STORAGE (BASELINE.json configs[4]; csrc/igemm_bf16s.hip) -------------------------------------------------
 * The entries above read fp32 tensors whatever cslgan_conv_t.compute says.  These read and write ACTIVATIONS and ACTIVATION
 * GRADIENTS stored as bfloat16 in HBM (NHWC, the same index maps), with a bfloat16 copy of the fp32 master filter; sums are
 * fp32 (v_mfma_f32_32x32x16_bf16), weight gradients / norms are fp32.  `*_bf16` flags give the element type of the tensor
 * named before them (0 = float, 1 = bfloat16).  Replaces the same reference calls as the fp32 entries: nn.Conv2d / nn.Linear
 * forward and data gradient (DCResNet_models.py:131-132,145), per-sample weight gradients (train.py:373,387). */

/* y = act(conv(x, w) + bias [+ residual]); x bf16 [N,H,W,C] with C % 8 == 0; w the fp32 KRSC filter, wb_ws a caller-owned
 * bf16 workspace of K*R*S*C elements holding bf16(w) (written here when repack != 0: once per parameter version);
 * residual (nullable) and y are bf16 or fp32 as flagged. */
int cslgan_conv2d_fwd_bf16s(const cslgan_conv_t* p, const void* x, const float* w, void* wb_ws, int repack, const float* bias,
                            const void* residual, int res_bf16, int act, void* y, int y_bf16, void* stream);

/* gx = conv_transpose(gy, w) (* lrelu'(mask)); gy bf16 [N,P,Q,K] with K % 8 == 0; wt_ws: bf16 workspace of K*R*S*C elements for
 * the per-parity-class filter matrices (written when repack != 0); mask (nullable) has gx's shape AND element type. */
int cslgan_conv2d_dgrad_bf16s(const cslgan_conv_t* p, const void* gy, const float* w, void* wt_ws, int repack, const void* mask,
                              void* gx, int gx_bf16, void* stream);

/* cslgan_conv2d_wgrad_grouped_f32 on bf16 gy [N,P,Q,K] and bf16 x [N,H,W,C] (K % 8 == 0, C % 8 == 0): gw fp32 (bf16 when
 * gw_bf16; nullable) and / or sq[N/group] += ||alpha gw_g||^2. */
int cslgan_conv2d_wgrad_grouped_bf16s(const cslgan_conv_t* p, const void* gy, const void* x, int group, float alpha, void* gw,
                                      int gw_bf16, float* sq, void* stream);

/* The critic's head nn.Linear(C, 1) (DCResNet_models.py:145) on bf16 features x [N, C], C % 8 == 0 (its forward is
 * cslgan_conv2d_fwd_bf16s with K == 1): data gradient gx[n,:] = bf16(gy[n] * bf16(w) (* lrelu'(mask[n,:]))) from fp32 gy [N], with
 * bf16 mask / gx [N, C]; grouped weight gradient gw[N/group, C] = alpha * sum_{n in g} gy[n] x[n,:] (fp32, nullable) and / or
 * sq[N/group] += ||gw_g||^2. */
int cslgan_linear_k1_dgrad_bf16s(const float* gy, const float* w, const void* mask_bf16, int N, int64_t C, void* gx_bf16, void* stream);
int cslgan_linear_k1_wgrad_bf16s(const float* gy, const void* x_bf16, const float* row_scale, int N, int64_t C, int group, float alpha,
                                 float* gw, float* sq, void* stream);   /* row_scale (nullable, [N]): gy[n] is weighted by row_scale[n] */

/* The critic's RGB first layer (3 -> 64 channels, 5x5, stride 2; csrc/conv_c3.hip) at the edge of the bf16-stored chain: fp32 image
 * and fp32 arithmetic as in cslgan_conv2d_fwd_f32 / cslgan_conv2d_wgrad_grouped_f32(group = 1), with the OUTPUT stored as bfloat16 /
 * the output gradient read as bfloat16.  Only the shapes the first-layer kernels take (even image, P % 8 == 0, Q % 16 == 0 forward;
 * Q in {16, 32, 64} with whole-row strips for the weight gradient); anything else is an error. */
int cslgan_conv2d_c3_fwd_bf16out(const cslgan_conv_t* p, const float* x, const float* w, const float* bias, int act, void* y_bf16,
                                 void* stream);
int cslgan_conv2d_c3_wgrad_bf16gy(const cslgan_conv_t* p, const void* gy_bf16, const float* x, float alpha, float* gw, float* sq,
                                  void* stream);

/* nn.GroupNorm(groups, C) + ReLU (DCResNet_models.py:55-57) writing bf16-stored activations: x fp32 or bfloat16 (x_bf16), y and
 * x_shuffled bfloat16; everything else as cslgan_groupnorm_act_f32 (fp32 statistics, d2s_W > 0 = depth-to-space output layout). */
int cslgan_groupnorm_act_bf16s(const void* x, int x_bf16, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                               float eps, int relu, float* stats_ws, void* y_bf16, int d2s_W, void* x_shuffled_bf16, void* stream);

/* The two 1..4-channel ends of the bf16-stored chain on the vector-ALU kernel (csrc/igemm_skinny.hip), reading bfloat16, computing and
 * writing fp32: the generator's output conv (64 -> 3 channels, stride 1; DCResNet_models.py:85) on a bf16 input, and the data gradient
 * of the critic's RGB first layer (the image gradient of gradient_penalty.py:48-50) from a bf16 output gradient.  Shapes the
 * vector-ALU kernel does not take are an error. */
int cslgan_conv2d_fwd_skinny_bf16in(const cslgan_conv_t* p, const void* x_bf16, const float* w, const float* bias, int act, float* y,
                                    void* stream);
int cslgan_conv2d_dgrad_skinny_bf16in(const cslgan_conv_t* p, const void* gy_bf16, const float* w, float* wt_ws, int repack, float* gx,
                                      void* stream);

/* cslgan_conv2d_wgrad_scaled_f32 on bf16 gy / x: gw[N/group] = alpha * sum_{n in g} row_scale[n] * (weight gradient of sample n) —
 * clip() + accumulate of ghost-clipped layers (train.py:399-402).  The fp32 weight is applied to each sample's ACCUMULATED product,
 * never to a bfloat16 operand, so every summed contribution is row_scale[n] times exactly the gradient
 * cslgan_conv2d_wgrad_grouped_bf16s(group = 1) would store.  Needs P*Q % 64 == 0. */
int cslgan_conv2d_wgrad_scaled_bf16s(const cslgan_conv_t* p, const void* gy, const void* x, const float* row_scale, int group, float alpha,
                                     float* gw, void* stream);

/* Element-type conversions at the edges of the bf16-stored chain (round-to-nearest-even / exact widening). */
int cslgan_cast_f32_bf16(const float* in, void* out_bf16, int64_t n, void* stream);
int cslgan_cast_bf16_f32(const void* in_bf16, float* out, int64_t n, void* stream);

/* cslgan_act_bwd_f32 / cslgan_bias_grad_grouped_f32 on bf16 tensors (fp32 sums; n % 8 == 0; K % 8 == 0, 256 % (K/8) == 0). */
int cslgan_act_bwd_bf16(const void* g, const void* y, int64_t n, float slope, void* out, void* stream);
int cslgan_bias_grad_grouped_bf16(const void* gy, int N, int PQ, int K, int group, float alpha, float* gb, float* sq, void* stream);

/* ---- pointwise / normalisation ------------------------------------------------------------- */

/* out = g * (y > 0 ? 1 : slope)   (LeakyReLU / ReLU backward from the OUTPUT y) */
int cslgan_act_bwd_f32(const float* g, const float* y, int64_t n, float slope, float* out, void* stream);

/* UpsampleConv's data movement (DCResNet_models.py:13-15): torch.cat([x]*4, 1) followed by F.pixel_shuffle(., 2).
 * pixel_shuffle is channel-major, so up[c][2h+i][2w+j] = x[(4c + 2i + j) mod C][h][w]: for C % 4 == 0 the C channels of
 * `up` are the C/4 channels of the plain depth-to-space tensor
 *       ps[n][2h+i][2w+j][c'] = x[n][h][w][4c' + 2i + j]                         (NHWC, [N][2H][2W][C/4])
 * repeated four times, and the conv that follows (DCResNet_models.py:16) equals a conv of ps with the filter summed over
 * the four channel groups — a quarter of the multiply-adds, the sums only re-associate the reference arithmetic.
 *   cslgan_depth_to_space_f32 : inverse == 0: ps from x[N][H][W][C];  inverse != 0: x[N][H][W][C] from ps (its gradient)
 *   cslgan_fold_channels4_f32 : unfold == 0: wf[row][c'] = sum_q w[row][c' + q*C/4], rows = K*R*S of a KRSC filter;
 *                               unfold != 0: gw[row][c' + q*C/4] = gwf[row][c']  (the gradient of w from that of wf)
 * The normalisation entries below can write their output (and the raw input) directly in the ps layout. */
int cslgan_depth_to_space_f32(const float* in, int N, int H, int W, int C, int inverse, float* out, void* stream);
int cslgan_fold_channels4_f32(const float* in, int64_t rows, int C, int unfold, float* out, void* stream);

/* GroupNorm(groups) + optional ReLU on NHWC x[N][HW][C] (DCResNet_models.py:55-57,63-67,101-102).
 * stats_ws: caller workspace [2*N*groups] floats (sum, centred sum of squares per statistic).
 * d2s_W > 0: the image is HW/d2s_W rows of d2s_W pixels and y is written depth-to-space shuffled,
 * y[n][2h+i][2w+j][c'] = act(norm(x))[n][h][w][4c'+2i+j] (C % 4 == 0); x_shuffled (nullable) receives the raw x in the
 * same layout — the inputs of ResBlockUp's convUp and shortcut (DCResNet_models.py:29-34) from one read of x.
 * d2s_W == 0: y has x's layout and x_shuffled must be NULL.
 * scratch (nullable; ABI v5 meaning): caller workspace of 2*N*groups*CSLGAN_NORM_PARTIAL_BLOCKS floats (no initialisation needed)
 * for per-workgroup partial statistics: the normalisation then runs as TWO launches (statistics, apply — the apply kernel adds the
 * partials in its prologue and publishes the final pairs to stats_ws) instead of four (zero, statistics with atomics, finalize,
 * apply).  One scratch must not be shared by launches on different streams. */
#define CSLGAN_NORM_PARTIAL_BLOCKS 64
/* The apply half of cslgan_groupnorm_act_f32 on statistics a conv epilogue left behind (cslgan_conv_t.gn_part): part holds
 * n_part = HW/64 (sum, centred sum of squares) pairs per image and group; everything else as cslgan_groupnorm_act_f32
 * (stats_ws receives the final pairs).  n_part <= CSLGAN_NORM_PARTIAL_BLOCKS. */
/* The same statistics as a per-(sample, channel) affine map for cslgan_conv_t.in_scale / in_shift:
 * scale[n*C + c] = gamma[c] * rstd(n, group of c), shift[n*C + c] = beta[c] - mean(n, group of c) * scale[n*C + c]. */
int cslgan_groupnorm_affine_parts_f32(const float* part, int n_part, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                                      float eps, float* scale, float* shift, void* stream);
int cslgan_groupnorm_apply_parts_f32(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                                     float eps, int relu, const float* part, int n_part, float* stats_ws, float* y, int d2s_W,
                                     float* x_shuffled, void* stream);
int cslgan_groupnorm_act_f32(const float* x, const float* gamma, const float* beta, int N, int HW, int C,
                             int groups, float eps, int relu, float* stats_ws, float* y, int d2s_W, float* x_shuffled,
                             float* scratch, void* stream);

/* Training-mode BatchNorm2d (+ optional ReLU) on NHWC x[rows][C], rows = N*H*W: batch statistics per channel,
 * running_mean / running_var (nullable pair) updated with `momentum` and the unbiased variance — the bn=True
 * generator of the non-per-sample modes (init_util.py:46, DCResNet_models.py:23,25,84).
 * stats_ws: caller workspace [2*C] floats.  rows_per_image / d2s_W / x_shuffled: as for GroupNorm (rows_per_image
 * = H*W is only read when d2s_W > 0). */
int cslgan_batchnorm_act_f32(const float* x, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                             int relu, float momentum, float* running_mean, float* running_var, float* stats_ws,
                             float* y, int64_t rows_per_image, int d2s_W, float* x_shuffled, float* scratch /* as above, 2*C floats + 1 ticket */,
                             void* stream);

/* Eval-mode BatchNorm2d (+ optional ReLU): y = act((x - running_mean) / sqrt(running_var + eps) * gamma + beta) — the
 * generator in eval() when images are sampled (train.py:298-308, gensamples.py:26-41). */
int cslgan_batchnorm_eval_act_f32(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, int64_t rows, int C, float eps, int relu, float* stats_ws,
                                  float* y, int64_t rows_per_image, int d2s_W, float* x_shuffled, void* stream);

/* Backward of GroupNorm / BatchNorm (+ReLU): dx, dgamma, dbeta from the forward's x, y (ReLU mask) and the
 * stats workspace the forward filled.  rows_per_stat = H*W (GroupNorm) or rows (BatchNorm, groups = C).
 * ws: caller workspace of cslgan_norm_bwd_ws_floats(...) floats. */
int cslgan_norm_act_bwd_f32(const float* x, const float* dy, const float* y, const float* gamma, const float* stats,
                            int64_t rows, int64_t rows_per_stat, int C, int groups, float eps, int relu, float* ws,
                            float* dx, float* dgamma, float* dbeta, void* stream);
int64_t cslgan_norm_bwd_ws_floats(int64_t rows, int64_t rows_per_stat, int C, int groups);

/* Adam (torch.optim.Adam semantics, train.py:76): in-place on p, m, v.  step is 1-based. */
int cslgan_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                         float b2, float eps, float weight_decay, int step, void* stream);
/* Same with the (1-based) step count read from device memory: a step captured in a HIP graph must not bake it in. */
int cslgan_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                             float b2, float eps, float weight_decay, const int32_t* step_dev, void* stream);

/* Adam over n_seg <= CSLGAN_MAX_SEGS tensors in ONE launch (the optimizer's per-parameter loop, train.py:76,484).  step_dev
 * (nullable): device int32 holding the 1-based step — when given, `step` is ignored (HIP-graph replay). */
int cslgan_adam_multi_f32(int n_seg, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                          float lr, float b1, float b2, float eps, float weight_decay, int step, const int32_t* step_dev,
                          void* stream);

/* ---- fused glue of the D-step (csrc/step_kernels.hip): each replaces a run of 4-us elementwise / reduction launches ---- */
/* The critic's losses over the row blocks of one pass (DCResNet_models.py:149-153: real_loss = -mean, fake_loss = +mean):
 * vec[s] = scale[s] * sum(x[block s]), total = sum_s vec[s].  sizes / scale: HOST arrays of n_seg <= CSLGAN_MAX_ROLES entries. */
int cslgan_segment_means_f32(const float* x, int n_seg, const int32_t* sizes, const float* scale, float* vec, float* total,
                             void* stream);
/* gx[i] = (g_total + g_vec[s(i)]) * scale[s(i)]; either gradient (device scalar / [n_seg]) may be NULL, not both. */
int cslgan_segment_means_bwd_f32(const float* g_total, const float* g_vec, int n_seg, const int32_t* sizes, const float* scale,
                                 float* gx, void* stream);
/* train.py:488-496, the logger.stats[...] += lines of train_D: acc7 += {adv loss (G gate), D Adv Loss, D Real Loss, D Fake Loss,
 * D Real Acc = 100*mean(d_real > 0), D Fake Acc = 100*mean(d_fake < 0), D Penalty (penalty may be NULL)}. */
int cslgan_dstep_stats_f32(const float* d_real, int n_real, const float* d_fake, int n_fake, const float* real_loss,
                           const float* fake_loss, const float* penalty, float* acc7, void* stream);
/* update_grad_logging (train.py:310-329) from the squared norms the clip already holds: sq[n_layers, ld], logged columns
 * [col0, col0 + B).  per_layer: one row of statistics per layer against max_norm[layer]; else ONE row from the flat norm
 * sqrt(sum_l sq) against max_norm[0].  Adds mean / population std / max of the norms, the clip norm and the fraction of
 * samples with min(1, C/(norm+eps)) < 0.999 to the five accumulators (each [n_layers] or [1]). */
int cslgan_grad_log_stats_f32(const float* sq, int n_layers, int64_t ld, int64_t col0, int B, const float* max_norm, int per_layer,
                              float eps, float* acc_mean, float* acc_std, float* acc_max, float* acc_c, float* acc_clipped,
                              void* stream);
/* Adaptive clipping of one fused critic pass in ONE launch (train.py:233-243 update_adaptive_clipping_params + :324
 * calc_clipping_factors; the device form ran stack, sqrt, mean, mul, stack, clip-factor, gather and three row-scale launches):
 *   r[l]   = mean_i (stat_max: max_i) sqrt(sq_adapt[l][i])                     over the n_adapt adaptive rows of layer l
 *   C      = scalar * r[l] per layer (per_layer) | scalar * ||r||_2 (one flat clip norm)
 *   sq_out = the layers' sq_rows stacked [n_layers][n_rows];  f = min(1, C / (norm + eps)), 1 for rows < first_private_row:
 *            per_layer [n_layers][n_rows] from each layer's own norm, else [n_rows] from the flat norm sqrt(sum_l sq)
 *   f_mat  (nullable, per_layer) [n_mat][n_rows] = f[mat_layer[m]];  jobs: job_dst[j][i] = job_scale[j] * f[job_layer[j]][job_first[j] + i]
 *            (flat: f[job_first[j] + i]) for i < job_count[j] — the row weights of the clip-weighted weight-gradient launches.
 * Pointers inside the struct are DEVICE pointers; the struct itself is read on the host. */
#define CSLGAN_MAX_CLIP_LAYERS 32
#define CSLGAN_MAX_CLIP_JOBS 16
typedef struct {
    int32_t n_layers, n_mat, n_jobs, _pad;
    const float* sq_adapt[CSLGAN_MAX_CLIP_LAYERS];
    const float* sq_rows[CSLGAN_MAX_CLIP_LAYERS];
    int32_t mat_layer[CSLGAN_MAX_CLIP_LAYERS];
    float* job_dst[CSLGAN_MAX_CLIP_JOBS];
    int64_t job_first[CSLGAN_MAX_CLIP_JOBS];
    int32_t job_layer[CSLGAN_MAX_CLIP_JOBS];
    int32_t job_count[CSLGAN_MAX_CLIP_JOBS];
    float job_scale[CSLGAN_MAX_CLIP_JOBS];
} cslgan_adaptive_clip_t;
int cslgan_adaptive_clip_f32(const cslgan_adaptive_clip_t* a, int64_t n_adapt, int64_t n_rows, int stat_max, float scalar, int per_layer,
                             float eps, int64_t first_private_row, float* r_out, float* c_out, float* sq_out, float* f_out, float* f_mat,
                             void* stream);
/* gradient_penalty.py:36: out[b,:] = alpha[b] * real[b,:] + (1 - alpha[b]) * fake[b,:]. */
int cslgan_lerp_rows_f32(const float* real, const float* fake, const float* alpha, int64_t n_rows, int64_t len, float* out,
                         void* stream);
/* gradient_penalty.py:52-54 (+ the weight and batch mean of :41,:65) in one launch: norm[b] = ||t[b,:]||, per[b] = coef *
 * (norm-1)^2 (one_sided: max(norm-1,0)^2), total (nullable) = sum_b per[b].  ticket: one zero-initialised device uint32 the
 * kernel uses to find its last workgroup and leaves at zero. */
int cslgan_lipschitz_term_f32(const float* t, int64_t n_rows, int64_t len, int one_sided, float coef, float* norm, float* per,
                              float* total, uint32_t* ticket, void* stream);
/* gt[b,:] = (g_total + g_per[b]) * coef * 2 (norm_b - 1)[clamped] * t[b,:] / norm_b; either gradient may be NULL, not both. */
int cslgan_lipschitz_term_bwd_f32(const float* t, const float* norm, const float* g_total, const float* g_per, int64_t n_rows,
                                  int64_t len, int one_sided, float coef, float* gt, void* stream);


/* ---- input pipeline (ABI v5) ----------------------------------------------------------------------------------------------------
 * out[n][h][w][c] = src[n][h][flip[n] ? W-1-w : w][c] * scale + bias: a batch of uint8 NHWC images (the preprocessed-tensor cache of
 * csl_gan_amd/pipeline.py, uploaded as 1 byte per element) becomes the normalised fp32 channels-last batch the critic reads.
 * Replaces the per-image ToTensor / RandomHorizontalFlip / Normalize of datasets.py:41-47 (flip: nullable [N] bytes drawn by the host
 * with p = 0.5; scale = 1/127.5, bias = -1 for Normalize(0.5, 0.5)). */
int cslgan_u8_to_f32_nhwc(const void* src_u8, const void* flip_u8, int N, int H, int W, int C, float scale, float bias, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CSLGAN_H */
