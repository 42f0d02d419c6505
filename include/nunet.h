/*
 * nunet.h — C ABI of libnunet.so: the MI355X (gfx950) UNet++ hot path.
 *
 * The reference (husheng876/pytorch_nested-unet) has no native interface; its
 * "plugin API" for this path is the Python surface
 *     archs.__dict__[arch](num_classes, input_channels, deep_supervision)   trains.py:219-221
 *     losses.__dict__[loss]()                                               trains.py:213
 *     iou_score(output, target)                                             trains.py:124,128
 *     optim.SGD(...).step()                                                 trains.py:229-231,133
 * Each entry below names the reference code whose arithmetic it replaces.
 *
 * Conventions
 *   - every entry returns 0 on success, a negative NUNET_E* otherwise;
 *     nunet_last_error() returns a thread-local message.
 *   - the caller owns every buffer (PyTorch caching allocator); the library
 *     allocates nothing on the device. Sizes come from *_bytes() queries.
 *   - all work is enqueued on the caller's hipStream_t; nothing synchronises.
 *   - activations are NHWC ("pitch" = elements between consecutive pixels);
 *     packed conv weights are KRSC: [tap][Cout][Cin] with tap = kh*3+kw.
 *   - dtype: storage type of activations and packed weights. Accumulation,
 *     BatchNorm statistics, loss, gradients of parameters are always fp32.
 */
#ifndef NUNET_H
#define NUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nunet_stream_t; /* hipStream_t */

enum { NUNET_F32 = 0, NUNET_BF16 = 1, NUNET_F16 = 2 };
enum {
  NUNET_OK = 0,
  NUNET_EINVAL = -1,  /* bad argument / unsupported shape */
  NUNET_ELAUNCH = -2, /* HIP launch error */
  NUNET_ENODEV = -3   /* no gfx950 device / code object */
};

int nunet_version(void);
const char* nunet_last_error(void);

/* ------------------------------------------------------------------------ */
/* 3x3 convolution, pad 1, stride 1 (nn.Conv2d(ci,co,3,padding=1),           */
/* reference finished/archs1.py:18,20) and its two gradients.                */
/* ------------------------------------------------------------------------ */
/* Per-channel reduction buffers that hundreds of workgroups add to (BatchNorm statistics, BatchNorm-backward sums)
 * are ORDER-INDEPENDENT fixed-point accumulators: one value = NUNET_FX_WORDS int64 words (hi, lo),
 * value = hi * 2^-20 + lo * 2^-60, added with integer atomics (which commute: the total is bit-identical from run to
 * run, unlike fp32 atomics whose result depends on arrival order). A buffer is
 *     int64_t acc[r][2][C][NUNET_FX_WORDS],  r = clamp(256 / C, 1, NUNET_BN_SUM_REPLICAS) replicas actually used
 * (a workgroup adds to replica (its index mod r): same-address atomics serialise at ~12 ns each on MI355X, 576
 * workgroups on one address cost 7 us at the end of a 14 us kernel); consumers sum the r replicas exactly.
 * Size it for NUNET_BN_SUM_REPLICAS replicas and zero it before the producing launch. */
#define NUNET_BN_SUM_REPLICAS 8
#define NUNET_FX_WORDS 2

enum { NUNET_TF_NONE = 0, NUNET_TF_BN_RELU = 1, NUNET_TF_BN_RELU_BWD = 2 };

typedef struct {
  int32_t dtype;
  int32_t N, H, W;
  /* input = channel concat of up to two NHWC sources (zero-copy torch.cat,
   * reference finished/archs1.py:116-131); C0, C1 multiples of the 64-byte channel chunk (32 / 16 for 2- / 4-byte types) */
  const void* src0; int32_t C0, P0;
  const void* src1; int32_t C1, P1; /* C1 = 0: unused */
  const void* wpack;                /* [9][Cout][Cin], dtype; Cin = C0+C1 */
  const float* bias;                /* [Cout] fp32 or NULL */
  /* output = channel split into up to two NHWC destinations */
  void* dst0; int32_t D0, Q0;
  void* dst1; int32_t D1, Q1;       /* D1 = 0: unused */
  int32_t acc_slot_w;               /* width (channels) of one dst0 slot, 0 = D0 */
  uint32_t acc0_mask;               /* bit k: dst0 slot k accumulates (+=) */
  int32_t acc1;                     /* dst1 accumulates */
  int64_t* stats;                   /* fixed-point sums [rep][2][Cout] (see above), pre-zeroed: += sum(y-b), sum((y-b)^2); or NULL */
  float* splitk_ws;                 /* optional fp32 scratch: lets grid-starved layers split the contraction over */
  int64_t splitk_ws_floats;         /* workgroups: up to S slabs of N*H*W*Cout floats, summed in fixed order. NULL: never split */
  /* Optional fused BatchNorm+ReLU backward REDUCE (used for the dgrad of a block's second conv, whose
   * output is the gradient entering the first conv's BN, archs1.py:18-19): with z = relu(bn(bn_y)),
   * dz = (z > 0) ? dst0 : 0, bn_sums[c] += sum dz, bn_sums[Cout + c] += sum dz * xhat. Needs D1 == 0,
   * Q0 == D0, no accumulation. NULL bn_y: off (nunet_bn_relu_bwd_reduce does the same as its own pass). */
  const void* bn_y; int32_t bn_py;  /* raw output of the BN's conv, [N*H*W][bn_py] */
  const float* bn_mean_invstd;      /* [2][Cout] saved by the BN forward */
  const float* bn_gamma; const float* bn_beta;
  int64_t* bn_sums;                 /* fixed-point sums [rep][2][Cout], pre-zeroed */
  /* Optional INPUT TRANSFORM, applied between the global load and the LDS write of the input tile (single source:
   * C1 == 0), so that the tensor between a BatchNorm and the conv that consumes it never makes its own round trip
   * through HBM on the dependency chain:
   *   NUNET_TF_BN_RELU      x = relu(bn(src0)): nn.BatchNorm2d + nn.ReLU between the two convs of a VGGBlock
   *                         (archs1.py:23-30). Training: batch statistics from tf_fx (the sums the producing conv took
   *                         in its epilogue); workgroup 0 also writes tf_mean_invstd and updates the running statistics
   *                         / num_batches_tracked (momentum tf_momentum). Eval: running statistics.
   *   NUNET_TF_BN_RELU_BWD  x = BatchNorm+ReLU backward of src0 (= dL/d relu(bn(tf_y))):
   *                         x = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)), sums from tf_fx, saved
   *                         statistics from tf_mean_invstd; workgroup 0 writes tf_dbeta = sum dz, tf_dgamma =
   *                         sum dz*xhat, tf_dbias = 0 (a conv bias in front of a BatchNorm has zero gradient).
   * tf_store (optional): the transformed tensor [N*H*W][tf_ps] is also written out (each pixel once, by the
   * workgroups of Cout-tile 0) for the weight-gradient kernel. */
  int32_t in_tf;
  int32_t tf_training;
  const void* tf_y; int32_t tf_py;
  const int64_t* tf_fx;
  const float* tf_gamma; const float* tf_beta; const float* tf_conv_bias;
  float* tf_running_mean; float* tf_running_var; int64_t* tf_nbt;
  float* tf_mean_invstd;
  float tf_momentum, tf_eps;
  float* tf_dgamma; float* tf_dbeta; float* tf_dbias;
  void* tf_store; int32_t tf_ps;
  int32_t tile;                     /* workgroup tile, pixels x output channels: 0 = chosen from the shape (the measured policy), 1 = 128 x 32,
                                     * 2 = 128 x 64, 3 = 256 x 32, 4 = 256 x 64 (2 and 4 need Cout % 64 == 0). Results do not depend on it
                                     * except through the K-split (fp32 summation order) of grid-starved layers. */
} nunet_conv_desc;

/* y = conv(cat(src0,src1)) + bias. Also used as dgrad with the flipped,
 * transposed pack (see nunet_pack_weights). */
int nunet_conv3x3_fwd(const nunet_conv_desc* d, nunet_stream_t s);

typedef struct {
  int32_t dtype;
  int32_t N, H, W;
  const void* src0; int32_t C0, P0; /* forward input (concat) */
  const void* src1; int32_t C1, P1;
  const void* dy;   int32_t Cout, PY; /* grad wrt raw conv output */
  float* dw;                        /* slab 0 of the K-split: [9][Cout][Cin] fp32; slab s at dw + s * slab_stride */
  int64_t slab_stride;              /* floats between slabs (0: 9*Cout*Cin) */
  int32_t max_slabs;                /* capacity of the caller's slab buffer (0: unlimited, i.e. nunet_conv3x3_wgrad_slabs of it) */
  int32_t target_wgs;               /* workgroups the launch should reach through the K-split (0: 256) */
  int64_t dw_floats;                /* capacity of `dw` in floats: the launch is refused (NUNET_EINVAL) unless every slab it would write fits */
  int32_t item_shape;               /* output channels x input channels one work item covers: 0 / 11 = 32 x 32 (default), 21 = 64 x 32
                                     * (needs Cout % 64 == 0, else 32 x 32), 12 = 32 x 64 (needs Cin >= 64). Wider items read the two operand
                                     * tiles fewer times but keep fewer workgroups resident; nunet_conv3x3_wgrad_slabs depends on it. */
} nunet_wgrad_desc;

/* Partial weight gradients: the contraction over pixels is split into nunet_conv3x3_wgrad_slabs(d) slices; slice s
 * writes dw_s[tap][co][ci] = sum_{p in slice s} dy[p][co] * x[p+tap][ci] to slab s with plain stores (every slab
 * fully overwritten; no atomics, no pre-zeroing). The caller sums the slabs in a fixed order (nunet_wgrad_reduce,
 * or the plan's batched reduce), which makes the gradient bit-reproducible. */
int32_t nunet_conv3x3_wgrad_slabs(const nunet_wgrad_desc* d);
int nunet_conv3x3_wgrad(const nunet_wgrad_desc* d, nunet_stream_t s);
/* Two independent problems (the two convolutions of one VGGBlock, archs1.py:18,20) in ONE launch. */
int nunet_conv3x3_wgrad_pair(const nunet_wgrad_desc* a, const nunet_wgrad_desc* b, nunet_stream_t stream);
/* out[i] (+)= sum_s slabs[s * slab_stride + i], i < n (n, slab_stride multiples of 4; 16-byte aligned) */
int nunet_wgrad_reduce(const float* slabs, int64_t slab_stride, int32_t nslabs, int64_t n, float* out,
                       int32_t accumulate, nunet_stream_t s);

/* OIHW fp32 -> packed KRSC `dtype`:
 *   wf[tap][co][ci]      (forward;  ci padded with zeros up to cin_pad)
 *   wd[8-tap][ci][co]    (dgrad: flipped taps, transposed; may be NULL) */
int nunet_pack_weights(const float* w_oihw, int32_t cout, int32_t cin, int32_t cin_pad,
                       int32_t dtype, void* wf, void* wd, nunet_stream_t s);
/* native fp32 dw[tap][co][cin_pad] -> OIHW grad (assign or +=) */
int nunet_unpack_wgrad(const float* dw, int32_t cout, int32_t cin, int32_t cin_pad,
                       float* g_oihw, int32_t accumulate, nunet_stream_t s);

/* ------------------------------------------------------------------------ */
/* BatchNorm2d (+ReLU) — nn.BatchNorm2d / nn.ReLU at finished/archs1.py:17-21 */
/* ------------------------------------------------------------------------ */
typedef struct {
  int32_t dtype;
  int32_t N, H, W, C;
  const void* y; int32_t PY;        /* conv output stored WITHOUT its bias */
  const float* conv_bias;           /* [C] or NULL: bias of the producing conv, folded in here */
  const int64_t* stats;             /* fixed-point sums / sums of squares of the stored y (training), as the conv's `stats` */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* save_mean_invstd;          /* [2][C] out (training): mean of the stored y, 1/sqrt(var+eps) */
  int32_t training;
  float momentum, eps;
  void* a; int32_t PA;              /* relu(bn(y)) */
  void* pooled; int32_t PP;         /* optional fused MaxPool2d(2,2) of a, or NULL */
  void* up; int32_t PU;             /* optional fused nn.Upsample(x2, bilinear, align_corners=True) of a -> [N][2H][2W][PU], or NULL:
                                     * extra workgroups of the same launch interpolate relu(bn(y)) from the raw tensor (every tap
                                     * rounded to the storage type first, so the result is bit-identical to nunet_upsample2x_fwd(a)) */
} nunet_bn_fwd_desc;
int nunet_bn_relu_fwd(const nunet_bn_fwd_desc* d, nunet_stream_t s);

typedef struct {
  int32_t dtype;
  int32_t N, H, W, C;
  const void* da; int32_t PDA;      /* grad wrt relu(bn(y)) */
  const void* y;  int32_t PY;
  const float* mean_invstd;         /* [2][C] */
  const float* gamma; const float* beta;
  int64_t* sums;                    /* fixed-point sums [rep][2][C], zeroed by caller: sum dz, sum dz*xhat */
  float* dgamma; float* dbeta;      /* [C] = sum dz*xhat, sum dz (written by _apply) */
  float* dbias;                     /* [C] = 0: the conv bias in front of a BatchNorm has gradient sum(dy) == 0 */
  void* dy; int32_t PDY;            /* grad wrt raw conv output */
} nunet_bn_bwd_desc;
int nunet_bn_relu_bwd_reduce(const nunet_bn_bwd_desc* d, nunet_stream_t s);
int nunet_bn_relu_bwd_apply(const nunet_bn_bwd_desc* d, nunet_stream_t s);

/* BatchNorm+ReLU backward REDUCE fused into the kernel that COMPLETES a gradient tensor: nunet_head_bwd_bnr, the last
 * writer of the output block's gradient, takes the reduce pass of that block's second BatchNorm (sum dz, sum dz * xhat,
 * as nunet_bn_relu_bwd_reduce) on the values it has just stored - one launch less at the head of the backward chain.
 * NULL descriptor: plain kernel. */
typedef struct {
  const void* y; int32_t PY;        /* raw conv output the BatchNorm normalised, [pixels][PY] */
  const float* mean_invstd;         /* [2][C] saved by the BatchNorm forward */
  const float* gamma; const float* beta;
  int64_t* sums;                    /* fixed-point sums [rep][2][C], pre-zeroed */
} nunet_bnr_desc;

/* ------------------------------------------------------------------------ */
/* MaxPool2d(2,2) (archs1.py:82) and Upsample(x2, bilinear, align_corners)   */
/* (archs1.py:83)                                                            */
/* ------------------------------------------------------------------------ */
int nunet_maxpool2x2_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                         const void* x, int32_t PX, void* y, int32_t PY, nunet_stream_t s);
/* dx (+)= route(dy) to the first max in PyTorch scan order */
int nunet_maxpool2x2_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                         const void* x, int32_t PX, const void* dy, int32_t PDY,
                         void* dx, int32_t PDX, int32_t accumulate, nunet_stream_t s);
/* H, W are the INPUT (low-res) extents; output is 2H x 2W */
int nunet_upsample2x_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                         const void* x, int32_t PX, void* y, int32_t PY, nunet_stream_t s);
int nunet_upsample2x_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                         const void* dy, int32_t PDY, void* dx, int32_t PDX,
                         int32_t accumulate, nunet_stream_t s);

/* ------------------------------------------------------------------------ */
/* 1x1 heads: nn.Conv2d(32, num_classes, 1) at archs1.py:105-111,133-143     */
/* ------------------------------------------------------------------------ */
/* logits NCHW fp32 [N][K][H][W] */
int nunet_head_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t K,
                   const void* x, int32_t PX, const float* w, const float* b,
                   float* logits, nunet_stream_t s);
/* dx (+)= dlogits . w ; parameter gradients come back as `nslabs` partial slabs, one per workgroup:
 * dw_slabs[s][K*C + K] = [weights grad | bias grad] of slab s (fully overwritten); the caller sums the
 * slabs (the plan's unpack kernel does). No atomics: same-address float atomics from hundreds of
 * workgroups cost more than the whole kernel. */
int nunet_head_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t K,
                   const void* x, int32_t PX, const float* w, const float* dlogits,
                   void* dx, int32_t PDX, int32_t accumulate,
                   float* dw_slabs, int32_t nslabs, nunet_stream_t s);   /* dw_slabs: nslabs * (K*C + K) floats */
int nunet_head_bwd_bnr(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t K,
                       const void* x, int32_t PX, const float* w, const float* dlogits,
                       void* dx, int32_t PDX, int32_t accumulate,
                       float* dw_slabs, int32_t nslabs, const nunet_bnr_desc* bnr, nunet_stream_t s);

/* ------------------------------------------------------------------------ */
/* BCEDiceLoss (losses.py:103-117), iou_score (metrics.py:6-18)              */
/* ------------------------------------------------------------------------ */
/* Every entry that takes a caller-provided workspace (or the plan arena) also takes its size in bytes and returns
 * NUNET_EINVAL when it is smaller than the matching *_bytes() query says: the library never writes past what the
 * caller said it owns (a layout change on one side of the ABI used to surface as a GPU memory fault).
 * ws: [N][3] (sum p*t, sum p, sum t) + [1] (sum bce) + per-block partial slabs (summed in fixed order). loss: [1]. */
size_t nunet_bce_dice_ws_bytes(int32_t N);
int nunet_bce_dice_fwd(const float* logits, const float* target, int32_t N, int64_t per_sample,
                       float* ws, size_t ws_bytes, float* loss, nunet_stream_t s);
/* dlogits = gscale[0] * dloss/dlogits */
int nunet_bce_dice_bwd(const float* logits, const float* target, int32_t N, int64_t per_sample,
                       const float* ws, size_t ws_bytes, const float* gscale, float* dlogits, nunet_stream_t s);
/* Fused loss step of the training loop (trains.py:118-128,135-136): BCEDice of every head,
 * dlogits of their mean, IoU counts of the last head. logits/dlogits: [heads][N][per].
 * loss_out: [heads+1] (per head, then the mean). meters (may be NULL): double[4]:
 * [0] += mean loss, [1] += IoU of this batch, [2],[3] = intersection / union counts.
 * loss_kind: NUNET_LOSS_BCE_DICE (losses.py:103-117) or NUNET_LOSS_LOVASZ_HINGE (losses.py:120-129, the loss behind the
 * reference's published table README.md:102-108; one class only: the reference squeezes dim 1). */
enum { NUNET_LOSS_BCE_DICE = 0, NUNET_LOSS_LOVASZ_HINGE = 1 };
size_t nunet_loss_step_ws_bytes(int32_t N, int64_t per_sample, int32_t heads, int32_t loss_kind);
int nunet_loss_step(const float* logits, const float* target, int32_t N, int64_t per_sample,
                    int32_t heads, int32_t loss_kind, float* ws, size_t ws_bytes, float* dlogits, float* loss_out,
                    double* meters, float iou_logit_threshold, nunet_stream_t s);
/* LovaszHingeLoss (losses.py:49-96,120-129; per_image=True, mean over images). logits/target: [N][per_image]
 * (num_classes must be 1: the reference squeezes dim 1). Per-image sort: in LDS up to 16384 pixels, chunk sorts +
 * global bitonic merge passes above (up to 2^22); ws of nunet_lovasz_ws_bytes(N, per_image) bytes, 256-byte aligned.
 * dlogits_unit receives d loss / d logits for an upstream gradient of 1; _bwd scales it by gscale[0]. */
size_t nunet_lovasz_ws_bytes(int32_t N, int64_t per_image);
int nunet_lovasz_hinge_fwd(const float* logits, const float* target, int32_t N, int64_t per_image,
                           float* ws, size_t ws_bytes, float* dlogits_unit, float* loss, nunet_stream_t s);
int nunet_lovasz_hinge_bwd(const float* dlogits_unit, const float* gscale, int64_t n, float* dlogits,
                           nunet_stream_t s);
/* counts[0] += |A&B|, counts[1] += |A|B|, A = logits >= logit_threshold, B = target > 0.5.
 * logit_threshold = the smallest fp32 logit x for which the REFERENCE's `torch.sigmoid(x) > 0.5` holds in fp32
 * (metrics.py:10-12; it is a few 1e-8 above zero, sigmoid(x) rounds to exactly 0.5 below it): integer counts stay
 * bit-exact against the reference. The caller finds it by bisection with the reference's sigmoid. */
int nunet_iou_counts(const float* logits, const float* target, int64_t n, float logit_threshold,
                     unsigned long long* counts, nunet_stream_t s);
/* Mask export of the evaluation driver (reference val.py:100-105): out[i] = uint8(sigmoid(logits[i]) * 255), bit-exact
 * against the reference's sigmoid: thresholds[k-1] (k = 1..255, fp32, device) = the smallest logit whose byte is >= k as
 * the REFERENCE computes it (the caller derives them once by bisection with the reference's sigmoid); NaN maps to 0. */
int nunet_sigmoid_u8(const float* logits, const float* thresholds, uint8_t* out, int64_t n, nunet_stream_t stream);

/* ------------------------------------------------------------------------ */
/* optim.SGD.step as configured at trains.py:229-231                         */
/* ------------------------------------------------------------------------ */
/* lr is read from device memory so a captured graph can be re-used across
 * scheduler steps. first != 0: momentum buffer := grad (torch semantics). */
int nunet_sgd_step(float* p, const float* g, float* mom, int64_t n, const float* lr_dev,
                   float momentum, float weight_decay, int32_t nesterov, int32_t first,
                   float grad_scale, nunet_stream_t s);

/* ------------------------------------------------------------------------ */
/* layout helpers                                                            */
/* ------------------------------------------------------------------------ */
/* Device-side input pipeline (reference dataset.py:66-74 + trains.py:258-259,266): uint8 HWC batch ->
 * ((u/255 - mean[c]) / std[c]) * post_scale as NCHW fp32 (post_scale = 1/255 reproduces the reference's
 * second division; masks: mean/std NULL, post_scale 1). aug (NULL = none): per sample
 * rot90 count (bits 0-1, np.rot90 sense; needs H == W when odd) | hflip (bit 2) | vflip (bit 3). */
int nunet_preprocess_u8(const uint8_t* u8_nhwc, int32_t N, int32_t H, int32_t W, int32_t C,
                        const float* mean, const float* stdv, const int32_t* aug, float post_scale,
                        float* out_nchw, nunet_stream_t s);
/* NCHW fp32 -> NHWC dtype with channel padding (zeros) up to cpad */
int nunet_nchw_to_nhwc(const float* x, int32_t N, int32_t C, int32_t H, int32_t W,
                       int32_t dtype, void* y, int32_t cpad, nunet_stream_t s);

/* ------------------------------------------------------------------------ */
/* Whole-network plan: NestedUNet.forward (archs1.py:113-143) + its backward */
/* ------------------------------------------------------------------------ */
typedef struct nunet_plan nunet_plan;
typedef struct {
  int32_t N, H, W;
  int32_t input_channels, num_classes, deep_supervision;
  int32_t dtype;
  int32_t unet; /* 0: NestedUNet, 1: plain UNet (archs1.py:35-71) */
} nunet_plan_cfg;

nunet_plan* nunet_plan_create(const nunet_plan_cfg* cfg);
void nunet_plan_destroy(nunet_plan* p);
size_t nunet_plan_arena_bytes(const nunet_plan* p);
/* number of fp32 parameters / BN running-stat floats / BN layers, in
 * reference state_dict order */
int64_t nunet_plan_param_count(const nunet_plan* p);
int64_t nunet_plan_bnbuf_count(const nunet_plan* p);
int32_t nunet_plan_bn_layers(const nunet_plan* p);
int32_t nunet_plan_num_heads(const nunet_plan* p);

/* params: flat fp32, reference parameters() order (OIHW conv weights).
 * bnbuf:  flat fp32 [running_mean, running_var] per BN in state_dict order.
 * nbt:    int64 per BN layer.
 * input:  NCHW fp32. logits: [heads][N][K][H][W] fp32.
 * arena / arena_bytes: the caller's buffer of at least nunet_plan_arena_bytes(p) bytes (NUNET_EINVAL when smaller), 256-byte aligned.
 * training: bit 0 = training mode (batch statistics, running-stat update); bit 1 = the packed weights in `arena`
 * are current (left so by nunet_plan_update / nunet_plan_repack on these parameters): skip the repack; bit 2 = the image
 * was staged into the arena by nunet_plan_stage_u8 (`input` is ignored and may be NULL). */
int nunet_plan_forward(nunet_plan* p, const float* params, float* bnbuf, int64_t* nbt,
                       const float* input, void* arena, size_t arena_bytes, float* logits, int32_t training,
                       nunet_stream_t s);
/* Device-side input pipeline into the plan (reference dataset.py:66-74, trains.py:258-259,266): a uint8 NHWC batch
 * [N][H][W][input_channels] -> ((u/255 - mean[c]) / std[c]) * post_scale in the storage dtype, written as the padded NHWC
 * image the first conv reads (same arithmetic and rounding as nunet_preprocess_u8 + the layout step of
 * nunet_plan_forward, which bit 2 of `training` then skips). aug: as nunet_preprocess_u8 (NULL = none). */
int nunet_plan_stage_u8(nunet_plan* p, const uint8_t* u8_nhwc, const float* mean, const float* stdv, const int32_t* aug,
                        float post_scale, void* arena, size_t arena_bytes, nunet_stream_t s);
/* grads: flat fp32 in params order. accumulate: += instead of assign. */
int nunet_plan_backward(nunet_plan* p, const float* params, const float* dlogits, void* arena, size_t arena_bytes,
                        float* grads, int32_t accumulate, nunet_stream_t s);
/* Backward in phases, for overlapping the data-parallel gradient exchange with the rest of backward:
 * phases bit 0 = clear scratch + heads + the last anti-diagonal's blocks (75 % of the gradient bytes),
 * bit 1 = the remaining blocks, bit 2 = unpack into `grads`. nunet_plan_backward == phases 7.
 * bit 3 (with bit 0) = leave the pass OPEN: no join - the plan keeps its lanes and dependency state, `s` is not made to wait;
 * nunet_plan_bucket0_wait(p, s2) then orders another stream behind exactly the kernels that complete the first bucket.
 * bit 4 (with bit 1) = CONTINUE that open pass on the same `s` and join at its end. A data-parallel caller issues
 * (1|8), bucket0_wait + the first bucket's exchange on a side stream, (2|16), the second bucket's exchange - eagerly or inside
 * ONE stream capture, where the exchange becomes a branch of the step's graph beside phase 2.
 * The native-layout fp32 gradient scratch lives in the arena at *byte_offset, in gradient-ready order:
 * its first *bucket0_floats floats are final after phase 1, all *total_floats after phase 2; a
 * data-parallel caller all-reduces those two ranges (sum) and then runs phase 4. */
int nunet_plan_backward_phase(nunet_plan* p, const float* params, const float* dlogits, void* arena, size_t arena_bytes,
                              float* grads, int32_t accumulate, int32_t phases, nunet_stream_t s);
int nunet_plan_grad_scratch(const nunet_plan* p, int64_t* byte_offset, int64_t* bucket0_floats,
                            int64_t* total_floats);
/* Exchange of the first bucket beside the rest of the backward pass, without cutting the pass in phases (replaces the
 * bucket hooks of torch DDP over the reference's nn.DataParallel-style replica training).
 * nunet_plan_bucket0_enable(p, 1) returns 1 when armed (0: not available for this plan, <0: error): a later backward call
 * with phases 1|2 together then records an event as soon as the first *bucket0_floats gradients are final - as an external
 * event record node when the call is captured into a graph. nunet_plan_bucket0_wait(p, s) makes stream `s` wait for the
 * most recent such record: call it after launching the pass (or the graph that holds it), then all-reduce bucket 0 on `s`.
 * With an OPEN pass (nunet_plan_backward_phase bit 3) nunet_plan_bucket0_wait needs no arming: `s` waits for the first bucket's
 * producing kernels directly (inside a capture `s` must be a stream the capture has not used yet). */
int nunet_plan_bucket0_enable(nunet_plan* p, int32_t on);
int nunet_plan_bucket0_wait(nunet_plan* p, nunet_stream_t s);
/* Fused optimiser step on the plan's buffers, replacing unpack (nunet_plan_backward_phase bit 2) + nunet_sgd_step +
 * the repack of the next forward: gradient scratch (optionally exchanged between ranks) -> torch.optim.SGD step
 * (reference trains.py:229-231; lr from device memory, momentum buffer `momentum`, weight decay, nesterov,
 * grad_scale = 1/world) on the fp32 master parameters -> both packed weight layouts. `grads` (flat OIHW, may be NULL)
 * receives the scaled gradients. */
int nunet_plan_update(nunet_plan* p, float* params, float* momentum, void* arena, size_t arena_bytes, const float* lr_dev, float mom, float wd,
                      int32_t nesterov, float grad_scale, float* grads, nunet_stream_t s);
/* The optimiser step INSIDE the backward pass: once parameters are set here, every whole backward pass steps each VGGBlock's
 * parameters (scratch -> SGD -> both packed 16-bit layouts, the arithmetic of nunet_plan_update) as an op scheduled behind that
 * block's weight gradients, beside the rest of the pass, and the heads at its end. The caller then calls neither
 * nunet_plan_update nor nunet_plan_sgd and sets bit 1 of the next forward's training flags (weights current).
 * Single-process training only (a data-parallel step exchanges the gradients before the update). params = NULL: off. */
int nunet_plan_set_inpass_update(nunet_plan* p, float* params, float* momentum, const float* lr_dev, float mom, float wd,
                                 int32_t nesterov, float grad_scale, float* grads);
/* The same optimiser step without the repack (the next nunet_plan_forward repacks as usual): gradient scratch -> SGD,
 * one launch instead of unpack + nunet_sgd_step, no OIHW gradient round trip unless `grads` is given. */
int nunet_plan_sgd(nunet_plan* p, float* params, float* momentum, void* arena, size_t arena_bytes, const float* lr_dev, float mom, float wd,
                   int32_t nesterov, float grad_scale, float* grads, nunet_stream_t s);
/* Repack the weight layouts from the fp32 parameters (what nunet_plan_forward does first unless told they are current). */
int nunet_plan_repack(nunet_plan* p, const float* params, void* arena, size_t arena_bytes, nunet_stream_t s);
/* Multi-lane issue (default on; env NUNET_MULTISTREAM=0 disables): the plan forks onto
 * its own streams (one per pyramid level + one per level for weight gradients), with event
 * dependencies per buffer, and re-joins `s` before returning control - all work is ordered
 * before anything the caller enqueues on `s` afterwards, and a capture of `s` records the
 * lanes as parallel branches of the same hipGraph. */
int nunet_plan_set_multistream(nunet_plan* p, int32_t enable);
/* How the plan issues its ops. NUNET_SCHEDULE_LANES (default): forked streams, captured as parallel branches of one hipGraph.
 * NUNET_SCHEDULE_WAVE: ONE stream, dependency order from a critical-path list scheduler, every ready 3x3 convolution of the same
 * kernel variant grouped into one launch - a single-stream graph (ROCm replays those as one batch of pre-built packets) whose
 * concurrency lives inside the launches. Bit-identical results; set before the first forward / capture. */
/* NUNET_SCHEDULE_LIST: the lane of every op is chosen by a list scheduler over the hazard graph of the program order (critical-path
 * priority, earliest start in a simulation with per-op costs) instead of following the op's block; meant for lanes that are real
 * in-order streams (NUNET_SEG_FLAGS). nunet_plan_calibrate(p, 1) ... one captured step replayed on one lane ...
 * nunet_plan_calibrate(p, 0) replaces the built-in cost estimates by the measured isolated cost of every op. */
enum { NUNET_SCHEDULE_LANES = 0, NUNET_SCHEDULE_WAVE = 1, NUNET_SCHEDULE_LIST = 2 };
int nunet_plan_calibrate(nunet_plan* p, int32_t begin);
/* Side lanes of the flag-synchronised program at the lowest stream priority (default 1: the chain lane's workgroups are dispatched
 * first) or at the default priority (0: needed when another library's high-priority stream lives in the process - RCCL's -,
 * beside which lowest-priority queues are served in time slices). Set before the first recording. */
int nunet_plan_set_lane_priority(nunet_plan* p, int32_t lowest);
/* Forget the side lanes: the next NUNET_SEG_* recording measures and picks new streams (the old ones stay alive until the plan is
 * destroyed, so programs recorded on them stay valid and the new candidates land on other hardware-queue slots). For a caller that
 * times its freshly recorded program and finds it far off the single-lane step (TrainStep does: which queue a stream inherits is
 * ROCm's choice, and unchecked 1 process in 5 came up with a program at 4.8 ms per step instead of 1.7). */
int nunet_plan_reset_lanes(nunet_plan* p);
int nunet_plan_set_schedule(nunet_plan* p, int32_t schedule);
/* debug/test access to an intermediate: name like "x0_0", "x2_1" (block
 * outputs, NHWC). Returns byte offset into arena; fills pitch/channels. */
int64_t nunet_plan_feature(const nunet_plan* p, int32_t i, int32_t j, int32_t* pitch,
                           int32_t* channels);

/* ------------------------------------------------------------------------ */
/* Measurement aid (bench.py roofline leg): hipEvent timing of every launch    */
/* issued by this thread between begin and end, aggregated per kernel class.   */
/* profile_end() waits for the events (the only host-synchronising entry).     */
/* flops/bytes are the ALGORITHMIC figures of the launches.        */
/* ------------------------------------------------------------------------ */
typedef struct {
  char name[48];
  int32_t launches;
  double ms, flops, bytes;
} nunet_prof_entry;
int nunet_profile_begin(void);
int nunet_profile_end(nunet_prof_entry* out, int32_t max_entries, int32_t* n_out);
/* ------------------------------------------------------------------------ */
/* hipGraph of a captured step (csrc/graph.hip).                               */
/* Replaces, for this path, what the reference gets from eager PyTorch         */
/* dispatch of trains.py:113-135: the step is captured once and replayed.      */
/* begin/end bracket a stream capture on an explicit stream; end()             */
/* instantiates the graph with a launch stream (hardware queue) of its own.    */
/* ------------------------------------------------------------------------ */
typedef struct nunet_graph nunet_graph;
int nunet_graph_begin(nunet_stream_t stream);
int nunet_graph_end(nunet_stream_t stream, nunet_graph** out);
int nunet_graph_launch(nunet_graph* graph, nunet_stream_t stream);
int nunet_graph_info(const nunet_graph* graph, int32_t* nodes, int32_t* edges_captured, int32_t* edges_final, int32_t* padding, int32_t* lanes);
void nunet_graph_destroy(nunet_graph* graph);

/* ------------------------------------------------------------------------ */
/* Segmented step (csrc/graph.hip): the same captured step as a PROGRAM of    */
/* single-stream graph segments on the plan's real streams, with the          */
/* cross-lane dependencies as event records / waits between graph launches.   */
/* ROCm 7.2 replays a hipGraph that has parallel branches node by node        */
/* (2.6-5 us of launch + synchronisation per node) but a single-stream graph  */
/* as one batch of pre-built packets (0.7 us per node); the step has ~160     */
/* nodes, ~75 of them on its critical chain.                                   */
/* Recording takes two passes over the same body (every launch made through   */
/* this library on `stream` or on the plan's lanes between begin and end):    */
/* dry = 1 launches nothing and finds the events that are waited on across     */
/* streams; dry = 0 records. nunet_seg_end returns the program (NULL after a   */
/* dry pass). The stream must outlive the program: it replays on it.           */
/* ------------------------------------------------------------------------ */
/* mode NUNET_SEG_FLAGS: every lane stream captures ONE single-stream graph for  */
/* the whole step and the cross-lane dependencies become device-side flags      */
/* (a one-thread signal kernel on the producer lane, a one-thread polling kernel */
/* on the consumer lane; csrc/graph.hip). Needs the lanes on distinct hardware   */
/* queues (the plan picks them by measurement); a wait that is not satisfied     */
/* within 4 s sets an error word and the NEXT nunet_seg_launch fails.            */
enum { NUNET_SEG_RECORD = 0, NUNET_SEG_DRY = 1, NUNET_SEG_FLAGS = 2 };
typedef struct nunet_seg nunet_seg;
int nunet_seg_begin(nunet_stream_t stream, int32_t mode);
int nunet_seg_end(nunet_stream_t stream, nunet_seg** out);
/* replay, ordered after what the caller queued on `stream`; `stream` continues after the program's last segment */
int nunet_seg_launch(nunet_seg* prog, nunet_stream_t stream);
int nunet_seg_info(const nunet_seg* prog, int32_t* graph_launches, int32_t* event_records, int32_t* event_waits, int32_t* kernel_nodes);
void nunet_seg_destroy(nunet_seg* prog);

#ifdef __cplusplus
}
#endif
#endif /* NUNET_H */
