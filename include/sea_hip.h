/*
 * sea_hip.h - C ABI of libsea_hip.so: the MI355X (gfx950) attack-side hot path of
 * SEA (Segmentation Ensemble Attack) evaluation and the PIR-AT inner PGD loop.
 *
 * The reference (nmndeep/Robust-Segmentation) is 100 % Python: it has no FFI, its "interface" for
 * this path is the ATen op sequences inside semseg/attacker.py, semseg/val.py, semseg/metrics.py
 * and tools/worse_only.py.  Every entry point below replaces one such sequence and cites it
 * (paths relative to the reference repo root).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all device pointers are HBM addresses on the
 *     current HIP device; the caller allocates and owns every buffer; the library keeps no global
 *     state and never allocates, frees or synchronises -> re-entrant per stream, graph-capturable.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All device
 *     work is stream ordered and asynchronous.
 *   - return value: 0 on success, otherwise a hipError_t value (1 = invalid argument).  Nothing
 *     throws across the ABI.
 *   - tensors are dense, NCHW ("planes") unless a layout argument says otherwise.
 *   - labels: any of int64 / int32 / int16 / uint8 (y_bytes = 8/4/2/1); the ignore label is -1
 *     (255 for uint8).  Labels outside [0,C) are treated as ignored.
 *   - loss modes: 0 mask-ce-avg, 1 mask-ce-bal, 2 js-avg, 3 ce / ce-avg.
 *   - logits dtype: 0 float32, 1 bfloat16, 2 float16 (dlogits has the same dtype and layout).
 */
#ifndef SEA_HIP_H
#define SEA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEA_MODE_MASK_CE 0
#define SEA_MODE_MASK_CE_BAL 1
#define SEA_MODE_JS 2
#define SEA_MODE_CE 3

#define SEA_DTYPE_F32 0
#define SEA_DTYPE_BF16 1
#define SEA_DTYPE_F16 2

#define SEA_LAYOUT_NCHW 0
#define SEA_LAYOUT_NHWC 1

/* library / build identification */
int sea_abi_version(void);
const char* sea_build_info(void);
/* host only: the multiplier / shift the kernels use to divide a work index (< 2^31) by d without an integer division:
 * n / d == (mulhi32(n, *m) + n) >> *s.  Exposed for the unit test of that arithmetic. */
int sea_fastdiv_magic(uint32_t d, uint32_t* m, uint32_t* s);

/* ------------------------------------------------------------------------------------------------
 * K1  APGD L-inf step with momentum.            replaces semseg/attacker.py:389-410, 456
 *   g2 = x_adv - x_old ; z = x_adv + step[b]*sign(grad) ; z = clip(clip(z, x-eps, x+eps), 0, 1)
 *   z = x_adv + (z - x_adv)*a + g2*(1-a)        ; out = clip(clip(z, x-eps, x+eps), 0, 1)
 * Bit-exact w.r.t. the float32 op sequence of the reference.  `out` must not alias an input.
 * n_per_img = 3*H*W elements per image; step_b has B entries.
 */
int sea_apgd_linf_step(const float* x, const float* x_adv, const float* x_old, const float* grad,
                       const float* step_b, float eps, float a, float* out, int B,
                       int64_t n_per_img, void* stream);

/* K5a random start: out = clip(x + eps*(2u-1), 0, 1).          semseg/attacker.py:293-294, 308 */
/* K1' -- the L2 branch of apgd_train (semseg/attacker.py:412-436; per-image norm = autoattack.other_utils.L2_norm, attacker.py:6):
 * normalised gradient step, projection onto the eps-ball (L2) around x and [0, 1], momentum combination, projection again.
 * Four streaming passes that recompute the element-wise chain; the three per-image norms are sums of per-block partials in a
 * fixed order (double), no atomics.  workspace: sea_apgd_l2_workspace_bytes(B) bytes, 8-byte aligned.  No shipped entry point
 * of the reference reaches this branch: it completes the drop-in surface. */
int64_t sea_apgd_l2_workspace_bytes(int B);
int sea_apgd_l2_step(const float* x, const float* x_adv, const float* x_old, const float* grad, const float* step_b, float eps,
                     float a, float* out, void* workspace, int B, int64_t n_per_img, void* stream);
int sea_linf_random_start(const float* x, const float* u, float eps, float* out, int64_t n,
                          void* stream);
/* K5b stage re-projection: out = clip(x + clip(z-x,-eps,eps), 0, 1). semseg/attacker.py:683-690 */
int sea_linf_project(const float* z, const float* x, float eps, float* out, int64_t n,
                     void* stream);

/* K6  PIR-AT PGD step on the perturbation.                      semseg/val.py:209-214 (168-172)
 *   d = delta + alpha*sign(grad) ; d = clip(X+d,0,1) - X ; delta_out = clip(d,-eps,eps)
 * If x_in_out != NULL it also receives the next model input: X+delta_out (clamp_input=0,
 * Pgd_Attack_1, val.py:200) or clip(X+delta_out,0,1) (clamp_input=1, Pgd_Attack val.py:150 and the
 * final x_adv of both, val.py:177, 217).  delta_out may alias delta. */
int sea_pgd_linf_step(const float* X, const float* delta, const float* grad, float alpha,
                      float eps, float* delta_out, float* x_in_out, int clamp_input, int64_t n,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2  fused per-pixel loss forward + logit gradient + tracking loss + accuracy + argmax.
 * replaces  masked_cross_entropy / _balanced / js_loss (semseg/attacker.py:143-173, 187-234),
 *           pixel_to_img_loss (237-240), the autograd of lines 347-350 / 462-469 down to the logits,
 *           the tracking loss (359-361, 473-475) and accuracy / argmax (370-373, 485-495);
 *           also the val.py losses (semseg/val.py:104-127) through `mode` + `grad_scale`.
 *
 *   logits   (B,C,H,W) [layout 0] or (B,H,W,C) [layout 1], dtype per `dtype`
 *   y        (B,H,W) labels, y_bytes wide
 *   w        (C) float32 class weights, required for mode/track_mode 1, else may be NULL
 *   grad_scale  upstream gradient per valid pixel (1/(H*W) for the per-image mean of the reference)
 *   dlogits  same shape/dtype/layout as logits, or NULL to skip the gradient (last APGD step,
 *            attacker.py:467, and clean/adversarial evaluation)
 *   pred     (B,H,W) argmax map (first maximum wins), pred_bytes wide (8/4/2/1) or NULL
 *   loss_px  (B,H,W) float32 per-pixel attack loss (reduction="none" of the reference functions) or NULL
 *   workspace  sea_loss_workspace_bytes(B,HW) bytes of scratch
 *   loss_sum / track_sum (B) float32: SUM over the image's pixels of the attack / tracking loss
 *            (the caller divides by H*W; attacker.py:240 divides by all pixels incl. ignored)
 *   n_correct (B) int32: #valid pixels with argmax == label
 *   Passing loss_sum = track_sum = n_correct = NULL defers the second reduction stage: the per-block
 *   records stay in `workspace` and sea_apgd_track (K7) sums them itself (`loss_workspace` argument), which
 *   saves one launch per APGD iteration.
 * Reductions are two-stage with a fixed order (no float atomics): results are run-to-run
 * deterministic.
 */
size_t sea_loss_workspace_bytes(int B, int64_t HW);
int sea_loss_fwd_bwd(const void* logits, int dtype, int layout, const void* y, int y_bytes,
                     const float* w, int mode, int track_mode, int B, int C, int64_t HW,
                     float grad_scale, void* dlogits, void* pred, int pred_bytes, float* loss_px,
                     void* workspace, size_t workspace_bytes, float* loss_sum, float* track_sum,
                     int32_t* n_correct, void* stream);
/* Benchmark hook: identical, but pins the pixels-per-lane (1/2/4, 0 = heuristic) of the register kernel. */
int sea_loss_fwd_bwd_tuned(const void* logits, int dtype, int layout, const void* y, int y_bytes,
                           const float* w, int mode, int track_mode, int B, int C, int64_t HW,
                           float grad_scale, void* dlogits, void* pred, int pred_bytes, float* loss_px,
                           void* workspace, size_t workspace_bytes, float* loss_sum, float* track_sum,
                           int32_t* n_correct, void* stream, int force_vec);

/* K2u  K2 fused with the model's final bilinear upsample.
 * replaces  F.interpolate(logits, size=(H,W), mode="bilinear", align_corners=False)
 *           (semseg/models/uperforseg.py:416-418, semseg/models/segmenter.py:228) + everything K2 replaces
 *           + the interpolate backward: the (B,C,H,W) logits and their gradient are never materialised.
 *   low   (B,C,h,w) float32 NCHW low-resolution logits;  dlow same shape (or NULL: no gradient)
 *   y     (B,H,W) labels; pred (B,H,W) argmax map at full resolution (or NULL)
 *   loss_sum / track_sum / n_correct as in K2 (sums over the H*W full-resolution pixels)
 * Any scale H/h, W/w >= 1 (ATen's align_corners=False source-index rule).  Gradients are gathered in a
 * fixed order (no atomics): deterministic.
 */
size_t sea_loss_upsampled_workspace_bytes(int B, int C, int h, int w, int H, int W);
int sea_loss_fwd_bwd_upsampled(const float* low, const void* y, int y_bytes, const float* w, int mode,
                               int track_mode, int B, int C, int h, int wl, int H, int W, float grad_scale,
                               float* dlow, void* pred, int pred_bytes, void* workspace,
                               size_t workspace_bytes, float* loss_sum, float* track_sum,
                               int32_t* n_correct, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  per-class integer statistics.
 * sea_class_counts replaces compute_iou_acc's loops (semseg/attacker.py:14-45), eval_performance
 * (tools/infer.py:90-116) and update_fn / update_fn_indiv (tools/worse_only.py:30-66).
 *   inter[c] += #{pred==y==c}; tgt_cnt[c] += #{y==c}; pred_cnt[c] += #{pred==c}
 *   mask_pred=1: pixels with ignored label do not count in pred_cnt (attacker.py:20, infer.py:90);
 *   mask_pred=0: they do (worse_only.py:42, 62).   union = tgt_cnt + pred_cnt - inter.
 *   per_image=1: outputs are (B,C), else (C).  Outputs are int64 and are ACCUMULATED into.
 * sea_confusion replaces Metrics.update's bincount (semseg/metrics.py:27-33): hist[t*C+p] += 1 for
 * every pixel whose label is valid.
 * Where only the argmax is needed (infer.py:88, metrics.py:28) call sea_loss_fwd_bwd with dlogits=NULL.
 */
int sea_class_counts(const void* pred, int pred_bytes, const void* y, int y_bytes, int B, int C,
                     int64_t HW, int mask_pred, int per_image, int64_t* inter, int64_t* pred_cnt,
                     int64_t* tgt_cnt, void* stream);
int sea_confusion(const void* pred, int pred_bytes, const void* y, int y_bytes, int64_t n, int C,
                  int64_t* hist, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4 + K7  device-resident APGD bookkeeping (no host round trip).
 *
 * sea_apgd_track: one tiny launch per iteration.  replaces semseg/attacker.py:370-383 (init=1),
 *   485-495 (best-adv tracking), 520-526 (best-loss tracking), 243-248 + 528-551 (oscillation
 *   check and step halving) and 568-569 (early stop, as a device flag).
 *     loss_sum/track_sum/n_correct  outputs of K2 for this iterate (or NULL + loss_workspace = the K2
 *               workspace of a deferred K2 call);  n_ignored (B) #ignored pixels
 *     iter      loop index i (ignored when init=1);  n_iter rows in loss_steps
 *     check_k   0, or the window k when this iteration is a checkpoint (schedule is data
 *               independent, attacker.py:528-551, so the host knows it)
 *   state (all (B) unless noted, updated in place):
 *     acc_cnt int32 (best = lowest #correct incl. ignored-as-correct), acc float = acc_cnt/HW,
 *     loss_best, loss_best_last, reduced_last, step (float32), loss_steps (n_iter,B) float32,
 *     flags uint8 (3,B): [0] copy x_adv->x_best_adv & pred->pred_best   (avg_acc <= acc)
 *                        [1] copy x_adv->x_best, grad->grad_best        (loss  >  loss_best)
 *                        [2] restart: x_adv<-x_best, grad<-grad_best    (step halved)
 *     done int32[1]: set to 1 when early_stop and every image has zero accuracy; once set, later
 *                    calls clear all flags and change nothing (the reference would have left the loop).
 * sea_select_copy: the conditional bulk copies selected by `flags` (attacker.py:494-495, 523-524,
 *   547-548), one launch.  pred/pred_best may be NULL.
 */
int sea_apgd_track(const float* loss_sum, const float* track_sum, const int32_t* n_correct,
                   const int32_t* n_ignored, int B, int64_t HW, int iter, int n_iter, int check_k,
                   int early_stop, int init, int32_t* acc_cnt, float* acc, float* loss_best,
                   float* loss_best_last, float* reduced_last, float* step, float* loss_steps,
                   uint8_t* flags, int32_t* done, const void* loss_workspace, void* stream);
/* Replayable forms for a caller that captures one loop iteration in a HIP graph (every per-iteration scalar in device
 * memory, fixed buffer addresses):
 *   sea_apgd_linf_step_graph: K1 IN PLACE (x_old <- x_adv, x_adv <- new iterate); a = 1 when *iter_dev == 0, else 0.75.
 *   sea_apgd_track_graph:     K7 (init = 0) with iter = *iter_dev and check_k = check_table[iter] (n_iter int32 entries,
 *                             0 = no checkpoint); advances *iter_dev by one at its end. */
int sea_apgd_linf_step_graph(const float* x, float* x_adv, float* x_old, const float* grad, const float* step_b,
                             float eps, const int32_t* iter_dev, int B, int64_t n_per_img, void* stream);
int sea_apgd_track_graph(const float* loss_sum, const float* track_sum, const int32_t* n_correct,
                         const int32_t* n_ignored, int B, int64_t HW, int32_t* iter_dev, const int32_t* check_table,
                         int n_iter, int early_stop, int32_t* acc_cnt, float* acc, float* loss_best,
                         float* loss_best_last, float* reduced_last, float* step, float* loss_steps, uint8_t* flags,
                         int32_t* done, const void* loss_workspace, void* stream);
/* The same two with the run's radius (one float) and length (one int32) in device memory as well: nothing run-specific is
 * left in the launch arguments, so ONE captured graph pair serves every stage, loss and batch of an evaluation (reference
 * semseg/attacker.py:691-728: three apgd_train calls per attack with eps 2e / 1.5e / e and 0.3n / 0.3n / 0.4n iterations;
 * tools/infer.py:338-370: three attacks per batch).  check_table and loss_steps must be sized for the longest run replayed. */
int sea_apgd_linf_step_graph_dev(const float* x, float* x_adv, float* x_old, const float* grad, const float* step_b,
                                 const float* eps_dev, const int32_t* iter_dev, int B, int64_t n_per_img, void* stream);
int sea_apgd_track_graph_dev(const float* loss_sum, const float* track_sum, const int32_t* n_correct,
                             const int32_t* n_ignored, int B, int64_t HW, int32_t* iter_dev, const int32_t* check_table,
                             const int32_t* n_iter_dev, int early_stop, int32_t* acc_cnt, float* acc, float* loss_best,
                             float* loss_best_last, float* reduced_last, float* step, float* loss_steps, uint8_t* flags,
                             int32_t* done, const void* loss_workspace, void* stream);
int sea_select_copy(const uint8_t* flags, float* x_adv, float* grad, float* x_best,
                    float* grad_best, float* x_best_adv, const void* pred, void* pred_best,
                    int pred_bytes, int B, int64_t n_per_img, int64_t HW, void* stream);

/* sea_count_ignored: n_ignored[b] = #{y[b] ignored}  (attacker.py:302-306, 489). */
int sea_count_ignored(const void* y, int y_bytes, int B, int64_t HW, int32_t* n_ignored,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8 + K9  worst-case bookkeeping over the attacks (HOST code, sequential by nature).
 * sea_worst_miou_greedy replaces evalSEA.worst_case_miou's greedy (tools/worse_only.py:279-334)
 * including _compute_miou / _compute_miou_subtraction (69-93) and Python's random.shuffle stream:
 *   ints/unions (A,N,C) float32 per-image tables (worse_only.py:200-234)
 *   mt_state    625 uint32: CPython `random.getstate()[1]` (624 words + position); advanced in
 *               place exactly as `random.shuffle` would, so the caller can `random.setstate` back
 *   outputs: *miou (fraction), selected (N) attack index per image, *rounds_run.
 * Arithmetic follows the reference bit for bit (float32 table differences, float32 rounding of the
 * running totals, float64 quotients, exactly rounded mean as in statistics.mean).
 */
int sea_worst_miou_greedy(const float* ints, const float* unions, int A, int N, int C,
                          uint32_t* mt_state, int n_rounds, double* miou, int32_t* selected,
                          int32_t* rounds_run);

/* ------------------------------------------------------------------------------------------------
 * M1  (model side) depthwise 7x7 convolution of the ConvNeXt block, stride 1, pad 3, fp32
 * NCHW: forward (flip=0, F.conv2d semantics incl. bias) and backward-data (flip=1: pass dy as x, bias
 * ignored).  Replaces the MIOpen/CK grouped convolution behind semseg/models/backbones/
 * convnext_orig.py:55-57, 75 - an HBM-bound stencil that the library runs ~25x below the roofline.
 *   x, y: (B*C, H, W) planes; w: (C,1,7,7); bias: (C) or NULL.
 */
int sea_dwconv7x7(const float* x, const float* w, const float* bias, float* y, int B, int C, int H,
                  int W, int flip, void* stream);
/* channels_last variant: x, y (B,H,W,C) contiguous, wt (49,C) taps-major (= w.view(C,49).T), C % 4 == 0.
 * With it the whole ConvNeXt block runs in NHWC: no permute / layout copy is left. */
int sea_dwconv7x7_nhwc(const float* x, const float* wt, const float* bias, float* y, int B, int C,
                       int H, int W, int flip, void* stream);
/* same, plus an optional addend of y's shape: y = conv(x) + addend (added after the taps: bitwise what a separate
 * element-wise add gives).  The backward of a ConvNeXt block (x + branch(x), convnext_orig.py:75-86) passes the skip
 * gradient here, which removes one element-wise pass per block.  bias and addend are mutually exclusive. */
int sea_dwconv7x7_nhwc_add(const float* x, const float* wt, const float* bias, const float* addend, float* y, int B,
                           int C, int H, int W, int flip, void* stream);

/* M2  (model side) bilinear up-sampling, align_corners=False, fp32 NCHW planes: forward and its
 * backward w.r.t. the input (gather formulation, deterministic; ATen scatters with atomics).
 * Replaces F.interpolate(..., mode="bilinear") in UperNet's FPN / PSP / final logits
 * (semseg/models/uperforseg.py:171-177, 236-262, 416-418) where ATen reaches ~0.3 TB/s.
 *   x / gx: (planes, h, w);  y / gy: (planes, H, W);  H >= h, W >= w, any (non-integer) scale.
 */
int sea_upsample_bilinear_fwd(const float* x, float* y, int64_t planes, int h, int w, int H, int W,
                              void* stream);
int sea_upsample_bilinear_bwd(const float* gy, float* gx, int64_t planes, int h, int w, int H, int W,
                              void* stream);
/* channels_last variants (x/gx: (B,h,w,C), y/gy: (B,H,W,C), C % 4 == 0, 16-byte aligned): lanes run
 * along C, so the head's NHWC tensors (what MIOpen's igemm convolutions return) need no layout copy.
 * y / gy may be a channel slice of a wider NHWC tensor (the FPN / PSP concatenation buffers,
 * uperforseg.py:171-177, 255-262): *_pixel_stride = floats between consecutive pixels (>= C, % 4 == 0),
 * batch stride = H*W*pixel_stride.  The up-sampled maps are then written straight into the buffer
 * torch.cat would have produced, and the gradient is gathered straight out of its gradient.
 * residual (NULL or dense (B,H,W,C)): y = residual + up(x), the FPN top-down add (uperforseg.py:243-250). */
int sea_upsample_bilinear_nhwc_fwd(const float* x, const float* residual, float* y, int B, int C, int h,
                                   int w, int H, int W, int64_t y_pixel_stride, void* stream);
int sea_upsample_bilinear_nhwc_bwd(const float* gy, float* gx, int B, int C, int h, int w, int H, int W,
                                   int64_t gy_pixel_stride, void* stream);

/* M4  (model side) Winograd F(m x m, 3 x 3) transforms, m = 2 or 4, for the 3x3 / stride 1 / pad 1
 * convolutions of the UperNet head (uperforseg.py:200-215 fpn_convs, 255-262 fpn_bottleneck), fp32 NHWC.
 * The convolution becomes  y = OUT( bmm( IN(x), FIL(w) ) ): the (m+2)^2 GEMMs in the middle are a plain
 * strided-batched fp32 GEMM (hipBLASLt); these three entry points are the HBM-bound transforms.
 *   T = sea_wino_tiles(B, H, W, m) = B * ceil(H/m) * ceil(W/m),  A = m + 2
 *   sea_wino_input_transform :  x (B,H,W,C) -> V (A*A, T, C); x may be a channel slice of a wider NHWC
 *                               tensor (x_pixel_stride floats between pixels, >= C, % 4 == 0: the gradient
 *                               of a concatenation buffer is read in place), and V a channel slice of a
 *                               wider (A*A, T, v_tile_stride) tensor (inputs that the reference concatenates
 *                               are transformed side by side, never concatenated); with gate (B,H,W,C, dense) the tiles are loaded
 *                               as  gate > 0 ? x * scale[c] : 0  (scale NULL = 1): the backward of the
 *                               fused epilogue below, applied to the incoming output gradient
 *   sea_wino_filter_transform:  w (Cout,Cin,3,3) -> U (A*A, Cin, Cout)             (flip = 0, forward)
 *                               w -> U (A*A, Cout, Cin) of the 180-degree rotated filters (flip = 1: the
 *                               convolution that yields the input gradient from the output gradient)
 *   sea_wino_output_transform:  M (A*A, T, C) -> y (B,H,W,C) = act(scale[c] * (conv + addend) + bias[c]);
 *                               addend (B,H,W,C) / scale / bias may be NULL, act = ReLU when relu != 0.  With scale/bias = the folded
 *                               eval-mode BatchNorm this is the whole ConvModule (uperforseg.py:119-146).
 * C % 4 == 0, 16-byte aligned pointers. */
int64_t sea_wino_tiles(int B, int H, int W, int m);
int sea_wino_input_transform(const float* x, int64_t x_pixel_stride, const float* gate, const float* scale,
                             float* V, int64_t v_tile_stride, int B, int C, int H, int W, int m, void* stream);
int sea_wino_filter_transform(const float* w, float* U, int Cout, int Cin, int m, int flip, void* stream);
/* sea_wino_input_transform that also max-accumulates, per tile t, the float bits of max|V[.][t][.]| into amax_out[t]
 * (sea_wino_tiles pre-zeroed words): the per-row fp16 x 2 scales of the Winograd-domain GEMMs (sea_gemm_split_f16 with
 * amax_rows = 1), each depending on its own tile only.  T < 2^31. */
int sea_wino_input_transform_amax(const float* x, int64_t x_pixel_stride, const float* gate, const float* scale, float* V,
                                  int64_t v_tile_stride, int B, int C, int H, int W, int m, uint32_t* amax_out, void* stream);
int sea_wino_output_transform(const float* M, const float* addend, const float* scale, const float* bias,
                              int relu, float* y, int B, int C, int H, int W, int m, void* stream);

/* M5  (model side) LayerNorm over the last dimension of a (rows, C) fp32 tensor with few channels (ConvNeXt:
 * C = 48..768, eps 1e-6; convnext_orig.py:19-40), forward and input gradient (w, b frozen).  A row is owned
 * by 16/32/64 lanes instead of a whole workgroup.  C % 4 == 0, C <= 1024; mean / rstd: (rows) saved statistics. */
int sea_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                      int64_t rows, int C, float eps, void* stream);
int sea_layernorm_bwd(const float* g, const float* x, const float* w, const float* mean, const float* rstd,
                      float* dx, int64_t rows, int C, void* stream);

/* M1w (model side, training) weight and bias gradient of the NHWC depthwise 7x7: gw (C,7,7), gb (C) or NULL from x and gy
 * (B,H,W,C) dense fp32 (PIR-AT's outer backward; convnext_orig.py:55-57).  Deterministic two-pass sum (per-tile partial
 * sums in `ws`, sea_dwconv7x7_nhwc_wgrad_workspace(B, C, H) floats, then the tiles in index order).  C % 4 == 0. */
int64_t sea_dwconv7x7_nhwc_wgrad_workspace(int B, int C, int H);
int sea_dwconv7x7_nhwc_wgrad(const float* x, const float* gy, float* gw, float* gb, float* ws, int B, int C, int H, int W,
                             void* stream);

/* M2'' (model side) adaptive average pooling of an NHWC fp32 map to oh x ow bins with ATen's bin rule (the pyramid pooling
 * of the head, uperforseg.py:150-177: 16 x 16 -> 1, 2, 3, 6), forward and input gradient; one block per bin x 16 channel
 * groups (ATen's NHWC kernel runs these maps on 8 blocks).  C % 4 == 0, oh <= H, ow <= W, dense tensors. */
int sea_adaptive_avg_pool_nhwc_fwd(const float* x, float* out, int B, int C, int H, int W, int oh, int ow, void* stream);
int sea_adaptive_avg_pool_nhwc_bwd(const float* g, float* dx, int B, int C, int H, int W, int oh, int ow, void* stream);

/* M9  (model side) the convolutional stem of the robust ConvNeXt backbones (backbones/convnext_orig.py:17-38:
 * Conv2d(3,48,3,s2,p1) -> LayerNorm(channels_first, eps 1e-6) -> GELU -> Conv2d(48,96,3,s2,p1) -> LayerNorm -> GELU), for
 * frozen parameters: forward and input gradient.  fp32 FMA in a fixed order (bitwise reproducible).  The image and its
 * gradient are NCHW; the tensors between the two convolutions are NHWC (the library's fast layout for the 48 -> 96
 * convolution, which stays a library call).
 *   sea_stem_conv1_ln_gelu: x (B,3,H,W) NCHW -> y (B,Ho,Wo,CO) NHWC = conv(x) + bias, and, unless a == NULL, a = GELU(LN_c(y))
 *                           (NHWC); Ho = (H-1)/2+1, Wo = (W-1)/2+1; w (CO,3,3,3); bias may be NULL; CO == 48.
 *   sea_stem_conv1_bwd    : dy (B,Ho,Wo,CO) NHWC -> dx (B,3,H,W) NCHW, the input gradient of that convolution.
 *   sea_ln_gelu_cl_fwd/bwd: a = GELU(LN over C of y) for y (B,HW,C) NHWC, C in {48, 96}; a / da are NHWC, or NCHW (B,C,HW)
 *                           when a_nchw != 0; dy (NHWC) from da with the statistics recomputed from y (nothing else is saved
 *                           by the forward).
 * Anything else (other CO / C) returns 1 (invalid argument): the caller keeps its library path. */
int sea_stem_conv1_ln_gelu(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                           float* y, float* a, int B, int CO, int H, int W, float eps, void* stream);
int sea_stem_conv1_bwd(const float* dy, const float* w, float* dx, int B, int CO, int H, int W, void* stream);
int sea_ln_gelu_cl_fwd(const float* y, const float* gamma, const float* beta, float* a, int a_nchw, int B, int C, int64_t HW,
                       float eps, void* stream);
int sea_ln_gelu_cl_bwd(const float* da, int a_nchw, const float* y, const float* gamma, const float* beta, float* dy, int B,
                       int C, int64_t HW, float eps, void* stream);

/* M6  (model side) the FPN bottleneck without up-sampling its coarse inputs (uperforseg.py:255-262).  Channel
 * mixing commutes with bilinear up-sampling, so for an input that is an xs up-sampling (s >= 3) the nine 3x3
 * taps are applied as one GEMM at the coarse resolution, G = f @ W -> (B,h,w,9,C), and only a gather is left
 * at the output resolution:
 *   sea_tap_gather_fwd: extra (B,H,W,C) (+)= sum_taps shift_tap(up(G[..., tap, :]))   (accumulate != 0: +=)
 *   sea_tap_gather_bwd: gz (B,H,W,C) -> dG (B,h,w,9,C), the exact adjoint (then df = dG @ W^T)
 *   sea_gate_scale    : out = gate > 0 ? g * scale[c] : 0, the backward of relu(scale * z + shift) (NHWC)
 * `extra` enters sea_wino_output_transform as `addend`.  C % 4 == 0, 16-byte aligned. */
int sea_tap_gather_fwd(const float* G, float* extra, int accumulate, int B, int C, int h, int w, int H, int W,
                       void* stream);
int sea_tap_gather_bwd(const float* gz, float* dG, int B, int C, int h, int w, int H, int W, void* stream);
int sea_gate_scale(const float* g, const float* gate, const float* scale, float* out, int64_t pixels, int C,
                   void* stream);

/* M3  (model side) NCHW <-> NHWC layout changes of the ConvNeXt block through LDS-tiled transposes,
 * fused with the per-channel layer scale and the residual add (convnext_orig.py:75-86: the two
 * `permute`s, `gamma * x` and `input + x`).  C % 4 == 0 and HW % 4 == 0.
 *   sea_nchw_to_nhwc: out[b,p,c] = scale[c] * in[b,c,p]                    (scale may be NULL)
 *   sea_nhwc_to_nchw: out[b,c,p] = residual[b,c,p] + scale[c] * in[b,p,c]  (scale / residual may be NULL)
 */
int sea_nchw_to_nhwc(const float* in, const float* scale, float* out, int B, int C, int64_t HW, void* stream);
int sea_nhwc_to_nchw(const float* in, const float* scale, const float* residual, float* out, int B, int C,
                     int64_t HW, void* stream);

/* 2x2 patch gather / scatter for the trunk's 2x2 / stride-2 down-sampling convolutions (convnext_orig.py:118-124):
 * pixels (B,H,W,C) channels_last <-> patch rows (B*H/2*W/2, 4*C) in (di, dj, c) order, so that the convolution is one GEMM
 * (bitwise reproducible, unlike the MIOpen kernel it replaces).  inverse != 0: patches -> pixels (the backward). */
int sea_patch2x2(const float* src, float* dst, int B, int H, int W, int C, int inverse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * M7  fp32 multi-head attention on the matrix cores (v_mfma_f32_32x32x2_f32), flash formulation.
 *     replaces the explicit softmax(q k^T * scale) v of semseg/models/backbones/vit_encoder.py:106-127 (fp32 operands,
 *     N = 1025 tokens x 6 heads x 64 in Segmenter ViT-S/16) and its autograd backward.
 * q, k, v: element (b, h, t, d) at ptr + b*sb + h*sh + t*st + d floats (d contiguous; the three may be slices of one
 *   packed (B,T,3,H,64) qkv tensor: sb = T*3*H*64, sh = 64, st = 3*H*64).  D must be 64.  Rows 16-byte aligned.
 * out (B,T,H*64) contiguous: the layout the output projection consumes.  lse (B,H,T): log-sum-exp of the scaled scores.
 * sea_attention_bwd: grad_out (B,T,H*64) contiguous; delta (B,H,T) scratch; dq/dk/dv are written with strides
 *   (gsb, gsh, gst) -- pass slices of one (B,T,3,H,64) gradient tensor to get d(qkv) without a concatenation.
 *   Deterministic (no atomics): S is recomputed in the dq kernel and in the dk/dv kernel.
 * Arithmetic (round 3, csrc/attention_bf16.hip): by default the products run on v_mfma_f32_32x32x16_bf16 with every fp32
 *   operand split into bf16 terms, fp32 accumulate: three terms (= the fp32 operands exactly, six products) in the
 *   forward, two terms in the backward (the attack consumes only the sign of the input gradient).  Environment, read per
 *   call: SEA_ATTN_TERMS / SEA_ATTN_TERMS_BWD = 3 | 2 | 0 (0 = the fp32 MFMA kernels of csrc/attention.hip).
 */
int sea_attention_fwd(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                      int T, int D, float scale, float* out, float* lse, void* stream);
int sea_attention_bwd(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                      int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                      float* delta, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst,
                      void* stream);
/* sea_attention_bwd with the number of bf16 terms of the backward products given by the caller (3, 2, or 0 = fp32 MFMA):
 * 3 when the weights are trained through this backward, 2 when only the sign of the input gradient is consumed. */
int sea_attention_bwd_terms(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                            int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                            float* delta, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst,
                            int terms, void* stream);
/* The forward and the backward with fp16 x 2 operands: 22 significant bits per operand in THREE MFMA products per pair (the accuracy of the
 * three-term bf16 mode at the cost of the two-term one).  Power-of-two scales: the exact row maximum for a lane's own
 * Q / K / V / dO row, one per (image, head) for the staged tiles (from a pre-pass this call launches), analytic bounds for the
 * soft-max operands P and dS.  amax_ws: 4 B H uint32 words of device scratch. */
/* sea_attention_fwd with the number of bf16 terms of the products given by the caller (3, 2, or 0 = fp32 MFMA) */
int sea_attention_fwd_terms(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                            int T, int D, float scale, float* out, float* lse, int terms, void* stream);
int sea_attention_fwd_f16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                          int T, int D, float scale, uint32_t* amax_ws, float* out, float* lse, void* stream);
int sea_attention_bwd_f16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                          int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                          float* delta, uint32_t* amax_ws, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh,
                          int64_t gst, void* stream);

/* ------------------------------------------------------------------------------------------------
 * M8  fp32 GEMM with frozen weights on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) by operand splitting:
 *     every fp32 operand is the exact sum of `terms` bf16 numbers (3: all 24 significant bits, six products, fp32-level
 *     accuracy; 2: 16 bits, three products), fp32 accumulation, fixed order (bitwise reproducible).
 *     replaces the hipBLASLt fp32 GEMMs behind the frozen-weight layers of the attacked model: the Winograd-domain
 *     products of the UperNet head's 3x3 convolutions (semseg/models/uperforseg.py:200-215, 255-262), its 1x1
 *     ConvModules (uperforseg.py:119-146) and the point-wise layers of the ConvNeXt blocks, forward and input gradient.
 *   C[g] (M x N, row stride ldc) = A[g] (M x K fp32, row stride lda) * W[g]^T + bias[n], optional ReLU; g < batch,
 *   batch strides in elements (A, C) / bytes (packed W).  K % 32 == 0, lda % 4 == 0, A 16-byte aligned.
 * sea_gemm_split_pack: W (N x K row-major, or K x N with trans = 1; row stride ldw) -> the packed, pre-split image the
 *   kernel reads ([K/32][terms][ceil128(N)][32] bf16, sea_gemm_split_packed_bytes bytes).  Done once per weight.
 * terms = 1 keeps one bf16 term per operand (the operands of a bf16-autocast GEMM, one product, fp32 accumulate and fp32
 *   in / out): the attack forward / backward of PIR-AT's inner PGD under TRAIN.AMP (BASELINE configs[3]).
 * terms = 22 selects fp16 x 2 operands instead (hi = fp16(x*s), mid = fp16(x*s - hi): 22 significant bits, three
 *   products on v_mfma_f32_32x32x16_f16): fp16 has 5 exponent bits, so the weights are packed with a power-of-two scale per
 *   output row and the activations are scaled by a power of two derived from max|A|, which the caller obtains on the device
 *   with sea_absmax_bits (one 4-byte word, no host round trip) and hands to sea_gemm_split_f16; the epilogue undoes both
 *   scales exactly.  Elements more than 2^28 below the tensor's maximum flush to zero.
 */
int64_t sea_gemm_split_packed_bytes(int N, int K, int terms);
int sea_absmax_bits(const float* A, int64_t lda, int M, int K, int batch, int64_t strideA, int rows_per_word,
                    uint32_t* out_bits, void* stream);
int sea_gemm_split_f16(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias, int relu, int M,
                       int N, int K, int batch, int64_t strideA, int64_t strideW_bytes, int64_t strideC,
                       const uint32_t* amax_bits, int amax_rows, uint32_t* out_amax, void* stream);
/* The activation scale is a power of two PER ROW of A, taken from amax_bits[row / amax_rows] (amax_rows = 0: one word for
 * the whole tensor): a word is the float bits of any upper bound of max|A| over its rows (all batch entries).  With one
 * word per image (amax_rows = rows of an image) or per row, an image's result does not depend on the images it shares a
 * batch with -- a per-tensor scale moves the sub-normal cut-off of the low fp16 term with the batch maximum, which changes
 * last bits and was measured to break the sharded evaluation's bitwise 1-rank == 2-rank property.
 * sea_absmax_bits(rows_per_word) computes the words exactly (rows_per_word = 0: one word; 1: one word per row, one writer
 * per word -- the scales of a GRADIENT operand, whose rows span many orders of magnitude); producers can supply them for
 * free: sea_wino_input_transform_amax (one word per tile), analytic bounds (LayerNorm output: sqrt(C) max|w| + max|b|; a
 * GEMM's output: bound(input) * max_n ||W_n||_1 + max|bias|), or the out_amax word (whole-tensor max|C|) of the GEMM whose
 * output feeds a non-expanding element-wise function.  A loose bound costs nothing up to a factor ~2^10 (fp16 is floating
 * point: only the sub-normal cut-off of the low term moves; error <= looseness * 2^-40 of the bound). */
int sea_gemm_split_pack(const float* W, int64_t ldw, int trans, int N, int K, int terms, void* out, void* stream);
/* sea_gemm_splitk_reduce: second pass of a split-K product.  A GEMM whose 128 x 128 tile grid cannot fill the chip (few
 * rows, long K: the point-wise layers of the last ConvNeXt stages, the FPN taps' input gradient) is launched as a BATCH of
 * `splits` GEMMs over K slices (batch strides of sea_gemm_split: A + s K/splits, packed slices of W) into a dense
 * (splits, M, N) workspace; this kernel adds the slices in the fixed order 0, 1, .. (bitwise reproducible), then bias,
 * ReLU and the optional max|C| word.  N % 4 == 0, ldc % 4 == 0, 16-byte aligned pointers. */
int sea_gemm_splitk_reduce(const float* partial, int splits, int M, int N, const float* bias, const float* addend,
                           int64_t ld_addend, int relu, float* C, int64_t ldc, uint32_t* out_amax, void* stream);
/* sea_gemm_split_fused: the same GEMM (terms 2 / 3 / 22; amax_bits / out_amax as above, 22 only) with the element-wise
 * neighbours of the MLP of a ConvNeXt / ViT block (convnext_orig.py:38-58, vit_encoder.py:41-60) in its epilogue:
 *   v = A W^T + bias + addend;  relu;  v *= GELU'(gelu_grad_of);  C = v;  gelu_out = GELU(v)     (exact erf GELU)
 * addend: the residual of the block (forward of the second projection); gelu_out: forward of the first projection (C keeps
 * the pre-activation the backward needs); gelu_grad_of: backward of the second projection (the gradient of the first
 * projection's output, without a separate GELU-backward pass).  gelu_out / gelu_grad_of have C's layout (ldc, strideC).
 * epi may be NULL (= sea_gemm_split / sea_gemm_split_f16). */
typedef struct SeaGemmEpilogue {
  const float* addend;
  int64_t ld_addend, stride_addend;
  float* gelu_out;
  const float* gelu_grad_of;
  const float* a_gelu_grad_of; /* PROLOGUE instead of epilogue (exclusive with the fields above; terms 2 or 22): A is read as
                                * A * GELU'(a_gelu_grad_of), same layout as A (lda, strideA): the GELU backward in front of the
                                * first projection's input-gradient GEMM, applied while the tile is staged */
  int a_gate;                  /* with a_gelu_grad_of: the tensor is a ReLU gate instead, A is read as (gate > 0 ? A : 0): the
                                * backward of a fused GEMM + ReLU in front of its input-gradient GEMM */
  int a_gelu;                  /* PROLOGUE (exclusive with everything above): A is read as GELU(A): the activation in front of the
                                * second projection's forward GEMM, without materialising GELU(A) */
  float a_amax_mul;            /* terms 22, > 0: the amax_bits words bound max|A| only after multiplication by this constant (a
                                * producer-side per-row bound carried through the GEMM in between: rowmax(g) * max_n ||W_n||_1,
                                * times max|GELU'| = 1.13 for the a_gelu_grad_of prologue); 0 = 1: the words as they are */
  const float* a_amax_mul_dev; /* the same constant as ONE float in device memory (takes precedence when non-NULL): a caller whose
                                * weights change every step (PIR-AT) derives it on the device without a host round trip */
} SeaGemmEpilogue;
int sea_gemm_split_fused(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias, int relu, int M,
                         int N, int K, int terms, int batch, int64_t strideA, int64_t strideW_bytes, int64_t strideC,
                         const uint32_t* amax_bits, int amax_rows, uint32_t* out_amax, const SeaGemmEpilogue* epi,
                         void* stream);
int sea_gemm_split(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias, int relu, int M,
                   int N, int K, int terms, int batch, int64_t strideA, int64_t strideW_bytes, int64_t strideC,
                   void* stream);
/* ------------------------------------------------------------------------------------------------
 * M10  the decode head's classifier, a 1 x 1 convolution onto a SMALL number of classes (semseg/models/uperforseg.py:262
 *      `cls_seg`; reference autograd for the input gradient), forward and input gradient for frozen weights, on
 *      v_mfma_f32_32x32x2_f32 (exact fp32 products).  Replaces torch.matmul (hipBLASLt) on these shapes.
 *   y / gy: (B P, K) fp32 NHWC rows, contiguous, 16-byte aligned; W: (cls, K) contiguous, 16-byte aligned; bias (cls) or NULL;
 *   out / g: (B, cls, P) fp32 NCHW, contiguous.  Shapes: sea_classifier_supported (P % 32 == 0, K % 64 == 0, K <= 1024,
 *   cls <= 32); anything else -> SEA_ERR_ARG.
 *   backward, optional: gate (layout of gy) and gate_scale (K): gy = gate > 0 ? gy * gate_scale[k] : 0, the backward of the
 *   ReLU(scale z + shift) that produced y (uperforseg.py:296-304 with the eval-mode BatchNorm folded) -- bit for bit
 *   sea_gate_scale applied to the stored gradient, without the 0.8 GB pass. */
int sea_classifier_supported(int P, int K, int cls);
int sea_classifier_fwd(const float* y, const float* W, const float* bias, float* out, int B, int P, int K, int cls, void* stream);
int sea_classifier_bwd(const float* g, const float* W, float* gy, int B, int P, int K, int cls, const float* gate,
                       const float* gate_scale, void* stream);
/* ------------------------------------------------------------------------------------------------
 * M8f  the MLP of a ConvNeXt block, y = res + W2 GELU(W1 x + b1) + b2 (semseg/models/backbones/convnext_orig.py:77-79:
 *      pwconv1 -> act -> pwconv2 with the layer scale folded into W2; reference autograd for the input gradient), as ONE
 *      kernel per direction.  Replaces, bit for bit, the pair of sea_gemm_split_fused launches (a_gelu / a_gelu_grad_of
 *      prologues), the residual add and -- backward -- the sea_absmax_bits(rows_per_word = 1) pass: a wave owns 32 rows for
 *      the whole MLP, the 4C-wide hidden tensor lives in accumulators and operand registers only (it is the HBM traffic that
 *      bounds the two-GEMM form at C = 96 / 192).  The backward recomputes t = W1 x + b1: the forward saves nothing but x.
 *   x, res, y / g, dx: (M, C) fp32 rows (row strides in elements, % 4 == 0, 16-byte aligned); H = 4 C; C in {96, 192}
 *     (sea_mlp_fused_supported).  W1p = sea_gemm_split_pack(w1: N = H, K = C, terms 22), W2p = pack(w2: N = C, K = H);
 *     backward: W2tp = pack(w2, trans = 1: N = H, K = C), W1tp = pack(w1, trans = 1: N = C, K = H).
 *   amax_x / amax_h: ONE device word each, float bits of an upper bound of max|x| / max|GELU(W1 x + b1)| (the analytic
 *     bounds of sea_gemm_split_f16's comment); amax_mul_dev: ONE float, rowmax|g[r]| * it bounds row r of (g W2) GELU'
 *     (SeaGemmEpilogue.a_amax_mul_dev).  b2 / res may be NULL. */
int sea_mlp_fused_supported(int C, int H);
/* The same pair with the block's LayerNorm (over the C channels, affine ln_w / ln_b, eps) in front of the MLP
 * (convnext_orig.py:75-77: norm -> pwconv1): x is the LayerNorm's INPUT, amax_x bounds its OUTPUT; the backward returns the
 * gradient w.r.t. x (frozen affine parameters).  The channel sums follow sea_layernorm_fwd / _bwd's summation order: replaces
 * those two launches and the round trip of the normalised tensor without changing a bit. */
int sea_ln_mlp_fused_fwd(const float* x, int64_t ldx, const float* ln_w, const float* ln_b, float ln_eps, const void* W1p,
                         const float* b1, const void* W2p, const float* b2, const float* res, int64_t ldres, float* y,
                         int64_t ldy, int M, int C, int H, const uint32_t* amax_x, const uint32_t* amax_h, void* stream);
int sea_ln_mlp_fused_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* ln_w, const float* ln_b,
                         float ln_eps, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* dx,
                         int64_t lddx, int M, int C, int H, const uint32_t* amax_x, const float* amax_mul_dev, void* stream);
/* test probe: out[0], out[1] (two pre-zeroed 64-bit device words) += the number of fp32 bit patterns for which the branch-free
 * GELU / GELU' evaluation inside the fused kernels differs from the one of sea_gemm_split's prologues (must stay 0, 0) */
int sea_probe_gelu_mismatches(unsigned long long* out, void* stream);
/* test probe: the LayerNorm of the fused kernels' prologue on its own (x dense (M, C), C in {96, 192}): yn, mean, rstd must equal
 * sea_layernorm_fwd's bit for bit */
int sea_probe_ln_rows(const float* x, const float* ln_w, const float* ln_b, float eps, int M, int C, float* yn, float* mean,
                      float* rstd, void* stream);
/* diagnostic build of the C = 96 fused kernels with s_memtime stamps around the segments of a loop iteration
 * (devtools/mlp_fused_stamps.py); dbg: 8 uint64 per wave.  W2p: pack(w2) forward, pack(w2, trans) backward. */
int sea_mlp_fused_stamps(int bwd, const float* g, int64_t ldg, const float* x, int64_t ldx, const void* W1p, const float* b1,
                         const void* W2p, const void* W1tp, const float* b2, const float* res, float* y, int M, int C,
                         const uint32_t* amax_x, const uint32_t* amax_h, const float* amax_mul_dev, unsigned long long* dbg,
                         void* stream);
int sea_mlp_fused_fwd(const float* x, int64_t ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                      const float* res, int64_t ldres, float* y, int64_t ldy, int M, int C, int H, const uint32_t* amax_x,
                      const uint32_t* amax_h, void* stream);
int sea_mlp_fused_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const void* W1p, const float* b1,
                      const void* W2tp, const void* W1tp, float* dx, int64_t lddx, int M, int C, int H,
                      const uint32_t* amax_x, const float* amax_mul_dev, void* stream);
/* Tuning knob of the sea_gemm_split* kernels: MFMA fragment shape, 32 (v_mfma_f32_32x32x16_{f16,bf16}; default) or 16
 * (v_mfma_f32_16x16x32_*; also env SEA_GEMM_SHAPE=16).  Any other argument only queries.  Returns the previous shape. */
int sea_gemm_split_mfma_shape(int shape);
/* K-loop pipeline of the sea_gemm_split* kernels at one or two terms per operand (32x32x16 fragments): 0 = the single-stage
 * loop (three blocks per CU), 1 = ping-pong (two LDS stages, one barrier per K step, the operand split in the shadow of the
 * wave's own MFMAs, loads two K steps ahead, two blocks per CU), 2 = chosen per launch (default; also env SEA_GEMM_PIPE=0|1|2):
 * ping-pong for launches with a prologue on A whose grid fills two blocks per CU.  Same split and same MFMA order: the kernels give the
 * same bits.  Any other argument only queries.  Returns the previous setting. */
int sea_gemm_split_pipeline(int pipe);

/* ------------------------------------------------------------------------------------------------
 * Measurement probes (bench.py / tools/kernel_bench.py only; nothing on the product path calls them).
 * sea_probe_stream_copy: dst[0:bytes] = src[0:bytes] with 16-byte-per-lane accesses (non_temporal != 0: nt loads and
 *   stores): the HBM copy ceiling the roofline fractions are also quoted against ("report against a
 *   measured stream ceiling on the box").  sea_probe_stream_read: read-only stream (sink: >= 2048 floats, untouched).
 * bytes % 16 == 0, 16-byte aligned pointers. */
int sea_probe_stream_copy(const void* src, void* dst, size_t bytes, int non_temporal, void* stream);
int sea_probe_stream_read(const void* src, float* sink, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEA_HIP_H */
