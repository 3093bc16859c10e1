/*
 * passion_hip.h -- C ABI of libpassion_hip.so: the MI355X (gfx950) kernels behind the
 * PASSION training hot path (mmformer / RFNet encoder-decoder forward+backward and the
 * self-distillation / Dice / weighted-CE losses).
 *
 * The reference (Jun-Jie-Shi/PASSION) has no native layer: every entry point below replaces a
 * stock torch/ATen op that the reference calls from code/models/{mmformer,rfnet,blocks}.py and
 * code/utils/criterions.py (file:line cited per function).  The Python host side
 * (passion_amd/ops.py) binds these with ctypes and wraps them in torch.autograd.Function.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (torch allocates); the library never
 *    allocates, frees or synchronises.  `stream` is a hipStream_t passed as void*.
 *  - Activations are channels-last: [B][D][H][W][C], C fastest.  dtype is PASSION_F32 or
 *    PASSION_BF16 (storage only -- accumulation is always f32).
 *  - Return value: 0 on success, -(hipError_t) on a HIP failure, -1000-k for argument errors.
 *  - Thread-safe given distinct streams.
 */
#ifndef PASSION_HIP_H
#define PASSION_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PASSION_F32 0
#define PASSION_BF16 1
#define PASSION_PAD_ZEROS 0
#define PASSION_PAD_REFLECT 1
#define PASSION_MAX_SEG 4

/* 2 since passion_seg_t carries dx_add (round 6): a caller built against another layout must not pass descriptors */
#define PASSION_ABI_VERSION 2
int passion_abi_version(void);
/* developer A/B switches, also read from the environment at load time (PASSION_CONV_LEAN, PASSION_CONV_PADOUT,
 * PASSION_FORCE_F32_MFMA, ...): "conv_lean", "conv_padout", "force_f32_mfma", "conv_target", "conv_merge",
 * "wgrad_wide", "conv_dbg", "conv_march", "conv_c1" (0: general kernel for the first encoder layer), "wgrad_march" (0: wgrad_lean_kernel also where the marching weight-gradient kernel
 * applies; environment PASSION_WGRAD_MARCH=0), "conv_s2" (0: zero-stuffed stride-2 input gradient), "conv_pw" / "conv_pw8" (0: the
 * 1x1x1 layers on the lean brick kernels instead of the streaming ones / the 8-channel 32x32x8 form), "conv_pwfinal" (0: pointwise
 * backward as dgrad + finish pass instead of the finish-free order with the means from the weight gradient), "flash_split" (n > 0:
 * that many key splits in the fused attention, 0 = automatic); round 6: "wgrad_group_s2", "c1_mfma", "conv_s2f", "wgrad_s2m" (0: the named
 * round-6 kernel back on the one it replaced) and "pack_f32" (0: passion_conv_pack_weight(s) write the two bf16 operand
 * images only -- for a caller that runs every conv on bf16 tensors with 8-channel-aligned concat segments; the kernels that read the f32
 * images then return -1024 instead of computing on stale ones).  PASSION_UPSAMPLE_SHARED=0 (environment only) keeps the flat x2 upsampling
 * kernels for bf16 tensors.  Returns the previous value, -1 for an unknown name.  The parity tests use it to run the
 * same conv through the general and the lean kernels. */
int passion_set_option(const char* name, int value);

/* ---- InstanceNorm statistics ------------------------------------------------------------
 * Replaces the reduction half of nn.InstanceNorm3d(affine=False) (code/models/blocks.py:18,
 * applied at blocks.py:313 / :368): per (b,c) mean and 1/sqrt(biased_var + eps) over D*H*W.
 * x: [B][V][C] (batch stride `bstride` elements); scratch: passion_instnorm_scratch_doubles(B,V,C) doubles
 * (per-workgroup partial sums, folded in a fixed order: no atomics, bitwise reproducible);
 * stats out: [B][C][2] f32 = (mean, rstd).  C must be a multiple of 4. */
int64_t passion_instnorm_scratch_doubles(int B, int64_t V, int C);
int passion_instnorm_stats(const void* x, int dtype, int B, int64_t V, int C, int64_t bstride,
                           float eps, double* scratch, float* stats, void* stream);

/* ---- fused (InstanceNorm -> LeakyReLU ->) Conv3d ----------------------------------------
 * Replaces general_conv3d_prenorm (code/models/blocks.py:300-316), bare nn.Conv3d
 * (code/models/mmformer.py:28,90,136-140,337-340,355) and, with several segments, the
 * MaskModal + torch.stack/cat feeding it (mmformer.py:316-325,405-409; Decoder cat at :96,:162).
 * An input "segment" is one tensor of the virtual channel concatenation. */
typedef struct {
    const void* x;        /* [B][Vin][C] */
    const float* stats;   /* [B][C][2] (mean, rstd): apply (x-mean)*rstd then LeakyReLU; NULL: raw */
    const uint8_t* mask;  /* [B][mask_ld] modality-present flags; NULL: present. Absent => contributes 0 */
    void* dx;             /* backward only: gradient wrt x, same shape/dtype as x; NULL: not needed */
    int64_t bstride;      /* batch stride of x (and dx) in elements */
    int32_t C;
    int32_t mask_col;
    int32_t mask_ld;
    int32_t bmod;         /* > 0: x (and stats) hold only bmod samples; sample b reads x[b % bmod] -- several masked passes over
                             the same features batched into one launch.  mask stays indexed by b; dx has bmod samples too and
                             receives the SUM of the passes' gradients.  0: plain batch */
    const void* dx_add;   /* backward only (round 6): a contiguous tensor of dx's shape and dtype that is ADDED into dx -- the gradient that
                             reached x by another path (the identity branch of x + c3(c2(x)), code/models/mmformer.py:34-38: its sum with
                             this conv's input gradient then costs no pass of its own).  NULL: none.  Not with bmod > 0 */
    const void* dx_add2;  /* a second such tensor (x read by two further consumers: a level's feature goes to decoder_sep and to the fusion
                             blocks); only with dx_add */
} passion_seg_t;

typedef struct {
    int32_t x_dtype;      /* dtype of segments (and of dx) */
    int32_t y_dtype;      /* dtype of y / residual / dy */
    int32_t B, D, H, W;   /* input spatial size */
    int32_t ksize;        /* 1 or 3 (padding = ksize/2) */
    int32_t stride;       /* 1 or 2 */
    int32_t pad_mode;     /* PASSION_PAD_* */
    int32_t nseg;
    int32_t Cout;         /* multiple of 4 */
    float lrelu_slope;
    passion_seg_t seg[PASSION_MAX_SEG];
} passion_conv_desc_t;

/* Size in floats of the packed-weight workspace for a conv (forward pack + dgrad pack). */
int64_t passion_conv_pack_floats(int Cin, int Cout, int ksize);
/* weight: f32 [Cout][Cin][k][k][k] (the reference nn.Conv3d layout) -> packed workspace. */
int passion_conv_pack_weight(const float* weight, int Cin, int Cout, int ksize, float* packed,
                             void* stream);
/* the same for n weights in one launch (all conv weights are re-packed once per optimizer step) */
int passion_conv_pack_weights(int n, const float* const* weights, const int* Cin, const int* Cout,
                              const int* ksize, float* const* packed, void* stream);
/* y[b][vo][co] = bias[co] + residual + sum_{tap,ci} W * act(x_padded).  residual may be NULL. */
int passion_conv3d_fwd(const passion_conv_desc_t* d, const float* packed, const float* bias,
                       const void* residual, void* y, void* stream);
/* The same, and the InstanceNorm statistics of y as well (the nn.InstanceNorm3d of the NEXT general_conv3d_prenorm,
 * code/models/blocks.py:313, 18): stats f32 [B][Cout][2] = (mean, 1/sqrt(biased var + eps)) over the output volume, as
 * passion_instnorm_stats(y) would give.  The bf16 stride-1 kernels accumulate the sums in their epilogue (no extra pass over
 * y, no extra launch); every other case runs the statistics kernels after the conv.  Scratch comes from the library's
 * per-stream region. */
int passion_conv3d_fwd_stats(const passion_conv_desc_t* d, const float* packed, const float* bias,
                             const void* residual, void* y, float* stats, float eps, void* stream);
/* Grouped launches along the batch: G <= 4 weight sets, sample b uses set b / (B / G) -- the four modality encoders of
 * mmformer (code/models/mmformer.py:385-388: four Encoder() instances of identical shape, one per modality) as ONE launch
 * per layer over the stacked batch [flair | t1ce | t1 | t2].  packed[g] / bias[g]: the sets' packed workspaces / biases
 * (bias may be NULL); one launch needs the workspaces equally spaced (packed[g + 1] - packed[g] constant, a multiple of 4
 * floats) and bf16 tensors -- anything else runs set by set.  Segments must not be batched passes (bmod = 0).
 * Backward: dW[g] receives [dweight | dbias] of set g (Cout*Cin*k^3 + Cout floats, overwritten); NULL = no weight gradients. */
int passion_conv3d_fwd_grouped(const passion_conv_desc_t* d, int G, const float* const* packed,
                               const float* const* bias, const void* residual, void* y, float* stats, float eps,
                               void* stream);
int passion_conv3d_bwd_grouped(const passion_conv_desc_t* d, int G, const float* const* packed, const void* dy,
                               float* const* dW, float* sums, void* stream);
/* Backward of the whole block.  dy: [B][Vout][Cout].
 *  dweight: f32 [Cout][Cin][k^3] and dbias f32 [Cout] are OVERWRITTEN (either may be NULL).  They are produced
 *  without atomics: every workgroup stores its partial sums into its own slot of a slab inside `scratch`,
 *  a second kernel folds the slots in a fixed order (bitwise reproducible).
 *  For every segment with dx != NULL the gradient wrt its raw x is written (through LeakyReLU and
 *  InstanceNorm when stats != NULL).
 *  scratch: f32 workspace of passion_conv_bwd_scratch_floats(d) floats (InstanceNorm-backward sums + slabs). */
int64_t passion_conv_bwd_scratch_floats(const passion_conv_desc_t* d);
int passion_conv3d_bwd(const passion_conv_desc_t* d, const float* packed, const void* dy,
                       float* dweight, float* dbias, float* scratch, void* stream);
/* The same backward in two phases, for a volume that is split along D over the ranks (BASELINE.json configs[4]): InstanceNorm
 * backward needs sum(g) and sum(g * x_hat) over the WHOLE volume.  phase 1: weight / bias gradient and the dgrad; on return
 * scratch[0 .. 2 * B * Cin) holds this rank's sums (b-major, (sum g, sum g x_hat) per input channel) -- the caller all-reduces
 * them in place.  phase 2: the finish with 1 / V multiplied by vscale (= V_local / V_global) and the mean-subtraction terms
 * applied to the depth planes [dwin_lo, dwin_hi) only (the halo planes of an extended slab belong to the neighbours).
 * Same arguments and scratch in both calls.  Batched-pass segments (bmod) are not supported here (-1019). */
int passion_conv3d_bwd_phase(const passion_conv_desc_t* d, const float* packed, const void* dy, float* dweight, float* dbias,
                             float* scratch, int phase, float vscale, int dwin_lo, int dwin_hi, void* stream);
/* ... and of a grouped launch (passion_conv3d_bwd_grouped): the four modality encoders of a sharded volume, one launch per layer and
 * phase (reference code/models/mmformer.py:385-388,397-415).  -1021: no grouped kernels for this shape / dtype (go set by set). */
int passion_conv3d_bwd_grouped_phase(const passion_conv_desc_t* d, int G, const float* const* packed, const void* dy,
                                     float* const* dW, float* scratch, int phase, float vscale, int dwin_lo, int dwin_hi,
                                     void* stream);

/* ---- trilinear resampling, align_corners=True ---------------------------------------------
 * Replaces nn.Upsample(scale_factor=k, mode='trilinear', align_corners=True)
 * (code/models/mmformer.py:70,143,366-370).  x: [B][D][H][W][C] -> y: [B][kD][kH][kW][C]. */
int passion_upsample_fwd(const void* x, void* y, int dtype, int B, int D, int H, int W, int C,
                         int k, void* stream);
/* ... and the InstanceNorm statistics of y ([B][C][2] f32 = mean, 1/sqrt(var + eps)) from the same pass (the decoders'
 * d*_c1 convs normalise the upsampled tensor, code/models/mmformer.py:98-101). */
int passion_upsample_fwd_stats(const void* x, void* y, int dtype, int B, int D, int H, int W, int C,
                               int k, float* stats, float eps, void* stream);
/* backward: k <= 2 one gather pass (scratch unused, may be NULL); k > 2 three separable passes through
 * `scratch` (passion_upsample_bwd_scratch_floats floats; NULL falls back to the one-pass gather). */
int64_t passion_upsample_bwd_scratch_floats(int B, int D, int H, int W, int C, int k);
int passion_upsample_bwd(const void* dy, void* dx, int dtype, int B, int D, int H, int W, int C,
                         int k, float* scratch, void* stream);

/* ---- depth sharding (BASELINE.json configs[4]: one volume split along D over the ranks; SURVEY 8b "halo_pack/unpack", 8e row 2).
 * halo_pack: plane d of every sample of a (B, D, P) tensor -> contiguous (B, P) send buffer (P elements per plane, P*sizeof % 16 == 0);
 * halo_unpack: a received (B, P) buffer -> plane d of y, overwriting (forward: the halo planes of the extended slab) or
 * accumulating (backward: the neighbour's gradient of our boundary plane).  The exchange itself is torch.distributed (RCCL). */
int passion_halo_pack(const void* x, void* buf, int dtype, int B, int D, int64_t P, int d, void* stream);
int passion_halo_unpack(const void* buf, void* y, int dtype, int B, int D, int64_t P, int d, int accumulate, void* stream);
/* round 5: the extended slab (B, n_lo + Dl + n_hi, P) of a 3x3x3 conv / upsampling input in ONE launch -- planes below n_lo - 1 and above
 * n_lo + Dl zero, plane n_lo - 1 / n_lo + Dl from its source (mode 0 zeros, 1 the (B, P) buffer lo / hi = the neighbour's plane,
 * 2 plane lo_pl / hi_pl of x itself = reflect at a global end) -- and its adjoint: gx = the owned planes of g, plus the halo plane's
 * gradient onto plane fold_lo / fold_hi (-1: none) where the halo was a reflection; send (2, B, P) = the two halo planes' gradients
 * (zeros where n_lo / n_hi = 0), which go back to their owners.  (passion_amd/depth.py HaloExtendFn; a zero-padded crop's adjoint is
 * the same launch with modes 0.) */
int passion_halo_extend(const void* x, const void* lo, const void* hi, void* ext, int dtype, int B, int Dl, int64_t P, int n_lo, int n_hi,
                        int lo_mode, int hi_mode, int lo_pl, int hi_pl, void* stream);
int passion_halo_extend_bwd(const void* g, void* gx, void* send, int dtype, int B, int Dl, int64_t P, int n_lo, int n_hi, int fold_lo, int fold_hi,
                            void* stream);
/* InstanceNorm statistics of a volume whose D axis is split over the ranks ("per IN: all-reduce of (sum x, sum x^2)"): the
 * (mean, rstd) pairs of a rank's slab of n voxels -> (n, n mean, n (var + mean^2)) in double, which the ranks SUM; and back. */
int passion_stats_to_moments(const float* stats, double* mom, int64_t BC, double n, double eps, void* stream);
int passion_moments_to_stats(const double* mom, float* stats, int64_t BC, double eps, void* stream);
/* trilinear x k (align_corners=True) where the D axis is one window of a larger volume: the rank holds input planes
 * [id0, id0 + Dl_in) of Dg and produces output planes [od0, od0 + Dl_out) of k*Dg with GLOBAL source coordinates;
 * bwd = the adjoint restricted to those outputs, gradient of every local input plane (halo planes included). */
int passion_upsample_window_fwd(const void* x, void* y, int dtype, int B, int Dl_in, int id0, int Dg, int Dl_out, int od0,
                                int H, int W, int C, int k, void* stream);
int passion_upsample_window_bwd(const void* dy, void* dx, int dtype, int B, int Dl_in, int id0, int Dg, int Dl_out, int od0,
                                int H, int W, int C, int k, void* stream);
/* round 5: the same adjoint as three separable passes (W and H over the whole axis, D over the window) through a scratch of
 * passion_upsample_window_bwd_scratch_floats floats -- what the k = 4 / 8 / 16 deep-supervision maps of a sharded volume need */
int64_t passion_upsample_window_bwd_scratch_floats(int B, int Dl_out, int H, int W, int C, int k);
int passion_upsample_window_bwd_sep(const void* dy, void* dx, int dtype, int B, int Dl_in, int id0, int Dg, int Dl_out, int od0,
                                    int H, int W, int C, int k, float* scratch, void* stream);

/* ---- layout / softmax -----------------------------------------------------------------------
 * softmax over the class axis (nn.Softmax(dim=1), mmformer.py:91,543,635; criterions.py:93-94
 * with 1/T).  Element (b,v,c) of a tensor lives at b*sb + v*sv + c*sc. */
int passion_softmax_fwd(const float* logits, int64_t lsb, int64_t lsv, int64_t lsc, float* probs,
                        int64_t psb, int64_t psv, int64_t psc, int B, int64_t V, int C,
                        float inv_temp, void* stream);
int passion_softmax_bwd(const float* probs, int64_t psb, int64_t psv, int64_t psc,
                        const float* dprobs, int64_t gsb, int64_t gsv, int64_t gsc, float* dlogits,
                        int64_t lsb, int64_t lsv, int64_t lsc, int B, int64_t V, int C,
                        float inv_temp, void* stream);
/* target one-hot (f64 or f32, [B][C][V]) -> f32 channels-last [B][V][C] (datasets_nii.py:151-153 output). */
int passion_target_to_cl(const void* target, int is_f64, float* out, int B, int64_t V, int C,
                         void* stream);
/* generic strided f32/bf16 -> strided f32/bf16 copy of a [B][V][C] tensor (layout changes). */
int passion_copy_strided(const void* src, int sdtype, int64_t ssb, int64_t ssv, int64_t ssc,
                         void* dst, int ddtype, int64_t dsb, int64_t dsv, int64_t dsc, int B,
                         int64_t V, int C, void* stream);

/* ---- segmentation losses --------------------------------------------------------------------
 * softmax_weighted_loss_bs + dice_loss_bs (code/utils/criterions.py:59-76, 25-38) on
 * probabilities.  probs strided as above; target f32 channels-last [B][V][C], C = 4.
 * sums: f32 scratch of passion_wce_dice_sums_floats(B) floats; its first [B][4*C] hold (num, l, r, celog per class) after the
 * call -- the backward (and the depth-sharded all-reduce) read those -- the rest is the reduction's per-block slots: the sums
 * are added in a fixed order, so the result repeats bit for bit run to run.
 * out: [B][2] = (weighted CE, Dice loss). */
int64_t passion_wce_dice_sums_floats(int B);
/* Bt: samples of `target` ([Bt][V][C]); prediction b is scored against target b % Bt (the four per-modality predictions of
 * decoder_sep against the one label volume, code/models/mmformer.py:546-549).  0 = B. */
int passion_wce_dice_fwd(const float* probs, int64_t psb, int64_t psv, int64_t psc,
                         const float* target, int B, int Bt, int64_t V, int C, float* sums, float* out,
                         void* stream);
/* dprobs (same strides as probs) = g[b][0]*dCE/dp + g[b][1]*dDice/dp ; g: [B][2] */
int passion_wce_dice_bwd(const float* probs, int64_t psb, int64_t psv, int64_t psc,
                         const float* target, const float* sums, const float* g, float* dprobs,
                         int B, int Bt, int64_t V, int C, void* stream);
/* The same two losses straight from LOGITS (f32 channels-last [B][V][C], contiguous): softmax(logits * inv_temp) is formed inside the
 * pass and never stored -- nn.Softmax(dim=1) at the end of Decoder_sep (code/models/mmformer.py:112-113) followed by
 * softmax_weighted_loss_bs + dice_loss_bs on its output (:546-549), whose 4 B probability volumes have no other reader in training.
 * sums / out / Bt as passion_wce_dice_fwd; the backward writes d(loss)/d(logits) (through the softmax Jacobian), g: [B][2].  Results
 * are bit for bit those of passion_softmax_fwd -> passion_wce_dice_fwd and passion_wce_dice_bwd -> passion_softmax_bwd. */
int passion_softmax_wce_dice_fwd(const float* logits, const float* target, int B, int Bt, int64_t V, int C, float inv_temp,
                                 float* sums, float* out, void* stream);
int passion_softmax_wce_dice_bwd(const float* logits, const float* target, const float* sums, const float* g, float* dlogits,
                                 int B, int Bt, int64_t V, int C, float inv_temp, void* stream);
/* temp_kl_loss_bs (criterions.py:92-103) on already softmaxed(/T) [and upsampled] probabilities.
 * out[b] = T^2 * mean_{c,v} pt*(log pt - log ps) with both clamped to [0.005,1].
 * work: f32 scratch of passion_kl_work_floats(B) floats (fixed-order reduction slots).
 * Bt: samples of the teacher tensor pt ([Bt][V][C]); student sample b is held against teacher sample b % Bt -- the four
 * uni-modal PASSION students (code/models/mmformer.py:468-531) batched against the one teacher.  0 = B. */
int64_t passion_kl_work_floats(int B);
int passion_kl_fwd(const float* ps, const float* pt, int B, int Bt, int64_t V, int C, float temp,
                   float* out, float* work, void* stream);
int passion_kl_bwd(const float* ps, const float* pt, const float* g, float* dps, int B, int Bt, int64_t V,
                   int C, float temp, void* stream);
/* prototype_passion_loss_bs (criterions.py:144-180).  fs, ft: [B][V][Cf] features (dtype),
 * target [B][V][4] f32.  work: f32 scratch of passion_proto_work_floats(B,Cf) floats.
 * out: [B][2] = (proto_loss, dist).  Backward writes dfs (dtype) given g[b] = dL/dproto_loss[b].
 * Bt as above: ft [Bt][V][Cf] and target [Bt][V][4] belong to sample b % Bt (0 = B). */
int64_t passion_proto_work_floats(int B, int Cf);
int passion_proto_fwd(const void* fs, const void* ft, int dtype, const float* target, int B, int Bt,
                      int64_t V, int Cf, float* work, float* out, void* stream);
int passion_proto_bwd(const void* fs, const void* ft, int dtype, const float* target,
                      const float* g, int B, int Bt, int64_t V, int Cf, float* work, void* dfs,
                      void* stream);

/* ---- attention ------------------------------------------------------------------------------
 * SelfAttention core (code/models/mmformer.py:220-224): softmax(q k^T * scale) v per (batch, head).
 * qkv: [B][N][3][H][Dh] (the nn.Linear(dim,3*dim) output), out: [B][N][H][Dh].
 * probs: [B][H][N][N] f32, kept for the backward.  Split in two so that the host can apply
 * attention dropout (mmformer.py:222) between the softmax and the PV product.
 * scores_fwd: probs = softmax_rows(scale * q k^T) */
int passion_attn_scores_fwd(const void* qkv, int dtype, int B, int N, int H, int Dh, float scale,
                            float* probs, void* stream);
/* out[b][n][h][:] = sum_key probs[b][h][n][key] * v[b][key][h][:]  (probs may have had dropout applied) */
int passion_attn_pv_fwd(const float* probs, const void* qkv, int dtype, int B, int N, int H, int Dh,
                        void* out, void* stream);
/* writes the V slot of dqkv and dprobs = dout v^T */
int passion_attn_pv_bwd(const float* probs, const void* qkv, const void* dout, int dtype, int B,
                        int N, int H, int Dh, float* dprobs, void* dqkv, void* stream);
/* softmax backward in place on dprobs (x scale), then writes the Q and K slots of dqkv */
int passion_attn_scores_bwd(const float* probs, float* dprobs, const void* qkv, int dtype, int B,
                            int N, int H, int Dh, float scale, void* dqkv, void* stream);

/* Fused attention (flash-style), bf16 tensors with Dh = 64 -- replaces SelfAttention.forward's
 * `attn = (q @ k.transpose(-2, -1)) * self.scale; attn = attn.softmax(dim=-1); attn = self.attn_drop(attn); x = attn @ v`
 * (code/models/mmformer.py:220-224) and its autograd backward without any [N][N] tensor in memory.
 *   fwd: out [B][N][H][64] bf16, lse [B][H][N] f32 (log2 of the row sum of exp2(scale*log2(e)*s); kept for the backward).
 *   bwd: dqkv [B][N][3][H][64] bf16 (all three slots written); delta: scratch of B*H*N floats.  Bitwise reproducible (no atomics).
 * drop_p > 0: attention dropout with keep bits = a counter-based hash of (seed, b, h, query, key); the backward must be given the
 * forward's (drop_p, seed).  Returns -1003 for any other head width (passion_flash_attn_supported tells). */
int passion_flash_attn_supported(int dtype, int Dh);
int passion_flash_attn_fwd(const void* qkv, int B, int N, int H, int Dh, float scale, float drop_p, uint64_t seed,
                           void* out, float* lse, void* stream);
int passion_flash_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, int H, int Dh,
                           float scale, float drop_p, uint64_t seed, float* delta, void* dqkv, void* stream);

/* ---- m2ftrans backbone (reference code/models/m2ftrans.py, code/models/blocks.py:25-297, code/models/mask.py) ----
 * gconv3d: grouped 3x3x3 Conv3d with C input = C output channels in groups of `cpg` (1: DepthWiseConvBlock.conv2, blocks.py:45-47,
 * nn.Conv3d(C, C, 3, 1, 1, groups=C); 4: GroupConvBlock.dwconv, blocks.py:77-79, groups = C / 4), zero padding, stride 1.
 * x, y, dy, dx: channels-last [B][D][H][W][C] in `dtype`; weight: `wsets` sets in the reference layout [wsets][C][cpg][3][3][3] f32,
 * bias [wsets][C] f32 or NULL; sample b uses set b / (B / wsets) (the four key_map_* and four value_map_* blocks of a
 * MultiMaskAttentionLayer, blocks.py:246-297, as one batch; wsets > 1 needs C % 64 == 0 or C in {8, 16, 32}).
 * bwd: dx / (dweight [wsets][C][cpg][27], dbias [wsets][C]) may be NULL; `part`: passion_gconv3d_part_floats floats of scratch
 * for the weight gradient (may be NULL when that returns 0); fixed-order sums (bitwise reproducible). */
int64_t passion_gconv3d_part_floats(int B, int D, int H, int W, int C, int cpg, int wsets);
int passion_gconv3d_fwd(const void* x, const float* weight, const float* bias, void* y, int dtype, int B, int D, int H, int W, int C,
                        int cpg, int wsets, void* stream);
int passion_gconv3d_bwd(const void* x, const float* weight, const void* dy, void* dx, float* dweight, float* dbias, float* part, int dtype,
                        int B, int D, int H, int W, int C, int cpg, int wsets, void* stream);
/* Weight_Attention (m2ftrans.py:420-460): y[b][v][:] = x[b % Bx][v][:] * w[b][v >> k] -- a [Bx][D][H][W][C] feature map scaled by the
 * nearest-upsampled (x 2^k per axis) weight maps w [Bv][D >> k][H >> k][W >> k] f32 of Bv / Bx masked passes (teacher + students)
 * in one launch.  bwd: dx[bx] = sum over the passes of dy * w (w carries no gradient: the attention maps are detached, :380). */
int passion_scale_nearest_fwd(const void* x, const float* w, void* y, int dtype, int Bv, int Bx, int D, int H, int W, int C, int k,
                              void* stream);
int passion_scale_nearest_bwd(const void* dy, const float* w, void* dx, int dtype, int Bv, int Bx, int D, int H, int W, int C, int k,
                              void* stream);
/* MaskedAttention (m2ftrans.py:318-351): passion_attn_scores_fwd with the block mask of mask_gen_fusion (mask.py:5-22) applied
 * before the row softmax (masked_fill(-inf)): the N tokens form G groups of N / G; query group i may look at key group j iff
 * allow[b][i][j] != 0 (allow: [B][G][G] bytes).  Masked probabilities are exactly 0, so passion_attn_pv_* / passion_attn_scores_bwd
 * apply unchanged. */
int passion_attn_scores_masked_fwd(const void* qkv, int dtype, int B, int N, int H, int Dh, float scale, const uint8_t* allow, int G,
                                   float* probs, void* stream);

/* LayerNorm over the channels of a channels-last token volume fused with what follows it in the cross-token blocks
 * (DepthWiseConvBlock / GroupConvBlock, blocks.py:32-109):  y = act(LN(x) * gamma + beta [+ res]),  act 0 = identity, 1 = GELU
 * (erf form).  x, res (or NULL), y, dy, dx, dres (or NULL): [rows][C] in `dtype`; the rows form rows / rows_per_group groups
 * with their own gamma / beta [G][C] f32 (G blocks of equal shape in one launch: the key_map_* / value_map_* blocks of
 * MultiMaskAttentionLayer, blocks.py:246-297).  stat [rows][2] f32 = (mean, rstd), written by fwd and read by bwd.
 * C: a multiple of 8 with C / 8 a power of two, <= 512 (passion_ln_act_supported).  bwd: dres = dy * act'(z) (only needed with
 * act = 1: with act = 0 the residual's gradient is dy itself); dgamma / dbeta [G][C] f32 through `part`
 * (passion_ln_act_part_floats floats), folded in a fixed order. */
int passion_ln_act_supported(int64_t rows, int C, int64_t rows_per_group);
int64_t passion_ln_act_part_floats(int64_t rows, int C, int64_t rows_per_group);
/* out_dtype: type of y (fwd) and dy (bwd): the input's dtype, or PASSION_BF16 for an f32 input without residual -- the f32 residual
 * stream of the transformers (mmformer.py:282-313) normalised straight into the bf16 operand of the GEMM that follows. */
int passion_ln_act_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* stat, int dtype,
                       int out_dtype, int64_t rows, int C, int64_t rows_per_group, float eps, int act, void* stream);
int passion_ln_act_bwd(const void* x, const float* gamma, const float* beta, const void* res, const float* stat, const void* dy,
                       void* dx, void* dres, float* part, float* dgamma, float* dbeta, int dtype, int out_dtype, int64_t rows, int C,
                       int64_t rows_per_group, int act, void* stream);

/* Attention dropout (nn.Dropout on the attention probabilities, code/models/mmformer.py:222) in one pass and without a
 * stored mask: y[i] = keep(i) ? x[i] / (1 - p) : 0 with keep(i) ~ Bernoulli(1 - p) drawn from a counter-based hash of
 * (seed, i).  The same (seed, p) reproduces the mask, which is how the backward applies it to dprobs.  y may alias x. */
int passion_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* round 5: the elementwise glue of a Transformer block (reference code/models/mmformer.py:226,247,256,271-275) in one pass each way:
 * out = [x +] drop(drop(act(y))): act 0 identity / 1 GELU (erf), ndrop 1 or 2 nn.Dropout(p) modules in a row (independent keep bits),
 * x the f32 residual stream or NULL; y / out dtypes (bf16, bf16), (bf16, f32), (f32, f32); n % 8 == 0.  bwd: dy = g * keep * scale
 * [* gelu'(y)] (the residual's gradient is g).  Keep bits from (seed + the step words of passion_set_step_words, element index). */
int passion_token_drop_fwd(const void* y, const float* x, void* out, int64_t n, int y_dtype, int out_dtype, float p, int ndrop, int act,
                           uint64_t seed, void* stream);
int passion_token_drop_bwd(const void* g, const void* y, void* dy, int64_t n, int y_dtype, int out_dtype, float p, int ndrop, int act,
                           uint64_t seed, void* stream);

/* ---- RFNet (post-norm backbone) streaming ops -------------------------------------------------------
 * norm_act: out[b][v][out_coff + c] = sum_k LeakyReLU((x_k - mean_k) * rstd_k), k = 1 or 2 inputs (x1 may be
 * NULL).  Materialises general_conv3d's Conv->IN->LReLU output (code/models/blocks.py:366-370) where a
 * non-conv consumer needs it: the encoder residual (code/models/rfnet.py:37,40,43,46), upsample inputs,
 * prototype features.  out has `out_ld` channels per voxel (writes the slice [out_coff, out_coff + C)). */
int passion_norm_act_fwd(const void* x0, const float* st0, const void* x1, const float* st1, void* out,
                         int dtype, int B, int64_t V, int C, int out_ld, int out_coff, float slope,
                         void* stream);
int64_t passion_norm_act_scratch_floats(int B, int64_t V, int C);
/* gradient wrt ONE input x (with its stats) given dout (same layout as out); call once per input */
int passion_norm_act_bwd(const void* x, const float* stats, const void* dout, void* dx, int dtype, int B,
                         int64_t V, int C, int out_ld, int out_coff, float slope, float* scratch,
                         void* stream);
/* region-aware modal fusion (code/models/blocks.py:495-517, 582-626).  y: 4 device pointers to the
 * modality feature maps [B][V][C]; mask: [B][mask_ld] present flags (NULL: all); prob: [B][V][4] f32
 * region probabilities (detached in the reference, rfnet.py:128).
 * region_gap: sums[b][(i*4+m)*C + c] = sum_v y_m[c] prob_i, sums[b][16*C + i] = sum_v prob_i. */
int64_t passion_region_scratch_floats(int B, int64_t V, int C);
int passion_region_gap_fwd(const void* const* y, const uint8_t* mask, int mask_ld, const float* prob,
                           int dtype, int B, int64_t V, int C, int bmod, float* scratch, float* sums, void* stream);
/* bmod > 0 (B = P * bmod virtual samples, pass p of physical sample bp is row p * bmod + bp): dy[m] holds bmod samples, the SUM of
   the passes' gradients (f32 sum, one rounding) */
int passion_region_gap_bwd(void* const* dy, const uint8_t* mask, int mask_ld, const float* prob,
                           const float* gS, int dtype, int B, int64_t V, int C, int bmod, void* stream);
/* region_mix: out[b][v][i*C + c] = prob_i(v) * sum_m w[b][i][m] * y_m[c](v);  w: [B][4][4] f32 */
int passion_region_mix_fwd(const void* const* y, const uint8_t* mask, int mask_ld, const float* prob,
                           const float* w, void* out, int dtype, int B, int64_t V, int C, int bmod, void* stream);
int passion_region_mix_bwd(const void* const* y, void* const* dy, const uint8_t* mask, int mask_ld,
                           const float* prob, const float* w, const void* dR, int dtype, int B, int64_t V,
                           int C, int bmod, float* scratch, float* dw, void* stream);
/* bmod > 0: y holds only bmod samples and (virtual) sample b of the B given reads y[b % bmod] -- the masked passes of the
 * PASSION students over the same features in one launch; mask, prob, w, out, dy stay indexed by b. */

/* ---- the deep-supervision terms with the x k upsampling inside (code/models/mmformer.py:366-370, 540-544: up2 / up4 / up8 / up16 of
 * a coarse softmax, then softmax_weighted_loss_bs + dice_loss_bs, criterions.py:59-76, 25-38; the students' temp_kl_loss_bs against the
 * upsampled teacher, criterions.py:92-103).  probs_coarse / ps_coarse / pt_coarse: f32 channels-last [B][D][H][W][4] (teacher: Bt
 * samples, student b against teacher b % Bt); target: [Bt][(D k)(H k)(W k)][4].  The x k map is blended in registers from the coarse
 * one (align_corners=True, the arithmetic of passion_upsample_fwd) and never materialised.  sums / work as for the plain entry points;
 * backward: scratch of passion_up_loss_bwd_scratch_floats(B, D, H, W, k) floats (the separable adjoint's two intermediates), result =
 * gradient wrt the COARSE map.  (D k)(H k)(W k) < 2^23; -1021 otherwise (callers then upsample explicitly). */
int passion_wce_dice_up_fwd(const float* probs_coarse, const float* target, int B, int Bt, int D, int H, int W, int k,
                            float* sums, float* out, void* stream);
int passion_wce_dice_up_bwd(const float* probs_coarse, const float* target, const float* sums, const float* g,
                            float* dprobs_coarse, int B, int Bt, int D, int H, int W, int k, float* scratch, void* stream);
int passion_kl_up_fwd(const float* ps_coarse, const float* pt_coarse, int B, int Bt, int D, int H, int W, int k, float temp,
                      float* out, float* work, void* stream);
int passion_kl_up_bwd(const float* ps_coarse, const float* pt_coarse, const float* g, float* dps_coarse, int B, int Bt, int D,
                      int H, int W, int k, float temp, float* scratch, void* stream);
int64_t passion_up_loss_bwd_scratch_floats(int B, int D, int H, int W, int k);

/* ---- bias gradient of the token GEMMs (nn.Linear of code/models/mmformer.py:196-199, 254-262): out[c] = sum_r g[r][c],
 * g [R][C] row-major (dtype), out f32 [C], C even.  Fixed summation order (bit-reproducible, also next to other streams).
 * work: f32 scratch of passion_colsum_work_floats(C) floats. */
int64_t passion_colsum_work_floats(int C);
int passion_colsum(const void* g, int dtype, int64_t R, int C, float* out, float* work, void* stream);
/* the same for G matrices [G][R][C] -> out [G][C] in one launch (work: G x passion_colsum_work_floats(C) floats) */
int passion_colsum_batched(const void* g, int dtype, int G, int64_t R, int C, float* out, float* work, void* stream);

/* ---- optimizer: torch.optim.AdamW(betas, eps, weight_decay, amsgrad) (code/train.py:95-96) for n f32 parameter
 * tensors in ceil(n/40) launches.  Host arrays of device pointers; `step` is the 1-based step count after increment. */
int passion_adamw_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, float* const* max_exp_avg_sq, const int64_t* numel, double lr,
                       double beta1, double beta2, double eps, double weight_decay, int64_t step, int amsgrad,
                       void* stream);
/* ---- a training step recorded once as a HIP graph and replayed (code/train.py:222-289 is the same launch sequence every
 * iteration): whatever changes from replay to replay must come from device memory, not from kernel arguments.
 * passion_adamw_hyper: the three step-dependent scalars of the update on the host, out3 = {lr * weight_decay, lr / (1 - beta1^step),
 * 1 / sqrt(1 - beta2^step)}; passion_adamw_step_dev: the same update as passion_adamw_step reading them from hyper_dev (device,
 * 3 floats) when the kernel runs.  passion_write_words: n <= 16 32-bit words (HOST array, passed by value) stored to dst (device)
 * by a one-block kernel on `stream` -- how the host hands the next replay its scalars.  passion_set_step_words(dev): from now on
 * every attention-dropout launch (passion_flash_attn_*, passion_dropout) adds the 64-bit word at dev[0..1] to its seed when it
 * RUNS (NULL = off, the default): a replay then draws fresh masks, and forward / backward of one replay agree. */
int passion_adamw_hyper(double lr, double beta1, double beta2, double weight_decay, int64_t step, float* out3);
int passion_adamw_step_dev(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, float* const* max_exp_avg_sq, const int64_t* numel,
                           double beta1, double beta2, double eps, int amsgrad, const float* hyper_dev, void* stream);
int passion_write_words(void* dst, int n, const uint32_t* words, void* stream);
/* streaming device copy of `bytes` (multiple of 16, 16-byte aligned pointers): 16 bytes per lane, grid-stride -- the measured
 * HBM copy rate bench.py quotes beside the 8 TB/s nominal peak */
int passion_stream_copy(void* dst, const void* src, int64_t bytes, void* stream);
void passion_set_step_words(const void* dev_words);

/* ---- training input pipeline on the GPU -------------------------------------------------
 * One sample: crop [S0][S1][S2][C] f32 (the RandCrop3D window, code/data/transforms.py:217-230) and its labels
 * seg [S0][S1][S2] int32  ->  x [C][S0][S1][S2] f32 and the one-hot target y_onehot [num_cls][S0][S1][S2] f64
 * (datasets_nii.py:143-153; y_cl = the same as f32 channels-last, either may be NULL) after
 * RandomRotion (transforms.py:86-120: scipy.ndimage.rotate order 0, cval -1, in the plane of sorted axes
 * (rot_a0, rot_a1) of [H,W,D]; rot6 = HOST array {m00, m01, m10, m11, off0, off1} of scipy's affine map, NULL = none),
 * seg_cval = label of a rotated-out voxel: scipy casts cval -1 to the label dtype (0 for the uint8 volumes of
 * preprocess_brats.py:78, -1 = np.eye(4)[-1] = class 3 for signed types);
 * RandomIntensityChange (:233-250: x * scale[h][c] + shift[h][c], device arrays [S0][C] f64) and RandomFlip (:133-155). */
int passion_augment(const float* crop, const int32_t* seg, int S0, int S1, int S2, int C, int num_cls,
                    int rot_a0, int rot_a1, const double* rot6, int seg_cval, int flip0, int flip1, int flip2,
                    const double* scale, const double* shift, float* x, double* y_onehot, float* y_cl,
                    void* stream);

/* ---- in-library kernel timing (bench.py roofline leg) ------------------------------------------
 * passion_prof_enable(mask): bit k of `mask` selects launches of kind k; selected conv3d launches are
 * bracketed by hipEvents on their own stream and their algorithmic bytes / flops recorded.  passion_prof_read drains up to `cap` records:
 * rec[i] = {kind, ms, bytes, flops, name}. kind: 0 fwd k3, 1 fwd k1, 2 dgrad, 3 wgrad, 4 / 5 fused attention forward / backward, 6 the sums fold + finish kernels that
 * complete a dgrad (name "finish of <dgrad kernel>", bytes = the dgrad's); name = the kernel instantiation
 * that ran, spelled as rocprofv3 prints it (e.g. "wgrad_lean_kernel<3, 1>"). */
typedef struct { int32_t kind; float ms; double bytes; double flops; char name[64]; } passion_prof_rec_t;
void passion_prof_enable(int on);
/* ... restricted to launches whose algorithmic bytes lie in [lo, hi] (0, 1e300 = all) */
void passion_prof_filter(double lo, double hi);
int passion_prof_read(passion_prof_rec_t* rec, int cap);
/* like passion_prof_read but leaves the records (and their events) in place: for launches recorded inside a HIP
 * graph capture, whose event pairs are re-recorded by every replay; ms = latest replay, -1 if it cannot be timed
 * (ROCm 7.2: hipEventElapsedTime refuses event nodes of a graph, so this returns -1 there) */
int passion_prof_peek(passion_prof_rec_t* rec, int cap);

#ifdef __cplusplus
}
#endif
#endif
