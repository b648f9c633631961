/*
 * abr_iod_hip.h -- C ABI of libabr_iod_hip.so (hand-written HIP for gfx950 / MI355X).
 *
 * Drop-in boundary for the Faster R-CNN + ARD training hot path of YuyangSunshine/ABR_IOD.
 * The reference crosses into native code through the pybind11 module `maskrcnn_benchmark._C`
 * (maskrcnn_benchmark/csrc/vision.cpp:9-25); everything else on the path is ATen (cuDNN/cuBLAS/
 * elementwise).  This library exports BOTH: section 1 mirrors `_C` one-to-one, sections 2-5 replace
 * the ATen work the path does per step.  Citations are file:line under /root/reference/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; no torch types anywhere
 *   - outputs are caller-allocated (the Python host allocates them with torch so allocator / stream
 *     semantics match the reference's at::empty, csrc/cuda/ROIAlign_cuda.cu:271,316)
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call returns
 *     without synchronising (reference: at::cuda::getCurrentCUDAStream(), ROIAlign_cuda.cu:273)
 *   - return 0 on success, <0 on error (ABR_E_*); abr_last_error() gives the message (the reference
 *     raises C++ exceptions -> Python RuntimeError; the Python host re-raises RuntimeError)
 *   - empty inputs are legal and return 0 without launching (ROIAlign_cuda.cu:278-281, nms.h:15-16)
 *   - layouts: ABR_NCHW is the reference's tensor layout (drop-in); ABR_NHWC is the library's native
 *     layout (channel-contiguous => 4 KB coalesced rows at C=1024)
 */
#ifndef ABR_IOD_HIP_H
#define ABR_IOD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABR_OK 0
#define ABR_E_INVALID (-1) /* bad argument (AT_ASSERTM in the reference) */
#define ABR_E_LAUNCH (-2)  /* hipGetLastError() != success after a launch (THCudaCheck) */
#define ABR_E_WORKSPACE (-3)

#define ABR_NCHW 0
#define ABR_NHWC 1

const char* abr_last_error(void);
int abr_version(void);
/* device properties the host side needs for grid sizing: out[0]=CU count, out[1]=LDS bytes/CU, out[2]=wave size */
int abr_device_info(int32_t* out_host);

/* Per-launch timing of the conv kernels with HIP events on the launch stream (bench.py's roofline leg; off by default).
 * abr_prof_end: out[id*6+{0,1,2}] = {launches, total ms, total flops} of the launches that had the device to themselves,
 * out[id*6+{3,4,5}] = the same for launches made while abr_prof_mark_overlap(1) was in force (the host runs weight-gradient
 * kernels on a second stream next to the dgrad chain: those launches share CUs and their event-bracketed duration is not a
 * property of the kernel).  id = 0 igemm 128x128, 1 igemm 128x64, 2 igemm 64x64, 3 igemm small-C (stem), 4 wgrad, 5/6 ROIAlign
 * fwd/bwd, 7 igemm bf16, 8 wgrad bf16 / bf16x6, 9 / 10 / 11 the bf16x6 implicit GEMM's 128x128 / 128x64 / 64x64 tile instances, 12 / 13 / 14 the same tiles of its weights-direct form.  Synchronises on the recorded events and stops profiling. */
int abr_prof_begin(void);
int abr_prof_mark_overlap(int on);
/* Order everything queued on `waiter` from now on behind everything queued on `signaller` so far (an event recorded on `signaller`, waited for by
 * `waiter`: torch's side.wait_stream(cur) in ONE call of ~2 us instead of ~15 us of Python -- the training step orders ~60 weight gradients so). */
int abr_stream_wait_stream(void* waiter, void* signaller);
/* bit id set = time that kernel (default all); every_nth = n > 1: bracket launch i of a kernel in step s iff (i + s) % n == 0 (an
 * event pair costs a ~6 us pipeline bubble per launch).  The caller marks step boundaries with abr_prof_step_begin(); over n
 * consecutive steps every launch position of the step's fixed launch sequence is sampled exactly once. */
int abr_prof_set_mask(uint32_t id_mask, int every_nth);
int abr_prof_step_begin(void);
int abr_prof_end(double* out_host, int n_ids);
/* out[id*2+{0,1}] = {launches, total flops} of EVERY launch of kernel id since abr_prof_begin, event-bracketed or not (call before
 * abr_prof_end or after: the totals survive until the next abr_prof_begin) */
int abr_prof_totals(double* out_host, int n_ids);
/* out[id] = ALGORITHMIC HBM bytes (every operand and the output once, + a fused residual / mask read) of every launch of kernel id since
 * abr_prof_begin: next to the PMC traffic of the same kernel it says how much of the traffic is re-reads (bench.py: roofline.traffic_algorithmic) */
int abr_prof_bytes(double* out_host, int n_ids);
/* out[id*2 + {0,1}] = {shader cycles (s_memtime), milliseconds (s_memrealtime)} that workgroup 0 of the sampled self-stamping launches of kernel
   id ran, summed by the last abr_prof_end: cycles / ms / 1e6 = the clock in GHz the chip sustained under that kernel (it clocks to its power budget) */
int abr_prof_clocks(double* out_host, int n_ids);
/* median duration (ms) of an event pair around an EMPTY kernel on a busy stream: the share of an event-bracketed duration that is
 * dispatch gap, not kernel; bench.py subtracts it from its live per-launch durations (synchronises the stream) */
int abr_prof_event_overhead_ms(double* out_host, void* stream);

/* Range guard of the bf16x6 arithmetic (abr_conv_desc::math == ABR_MATH_BF16X6).  The three-way bf16 split x = x0 + x1 + x2 is
 * EXACT -- and the contraction then meets the fp32 error bound -- for operands that are zero or have 2^-110 <= |x| < 2^128 (finite);
 * below 2^-110 the low planes leave bf16's normal range and precision decays towards bf16's, and an inf operand yields NaN
 * (inf - inf in the split) where an fp32 multiply-add chain yields inf.  Every bf16x6 kernel inspects each operand element once per
 * GEMM and ORs ABR_X6_FLAG_* into a device word; abr_x6_range_flags copies it to *out_host (synchronising `stream`) and, if `reset`,
 * clears it.  A caller that needs the fp32 result for out-of-domain data re-runs the operation with ABR_MATH_F32 when a flag is up
 * (the Python host does: ops.x6_range_flags, engine/trainer.py). */
#define ABR_X6_FLAG_TINY 1u
#define ABR_X6_FLAG_NONFINITE 2u
/* f16x3 arithmetic (ABR_MATH_F16X3), same flag word: STALE = an amax word did not carry the epoch the caller named (a caller bug: the kernel
 * then ran with scale 1); a non-finite amax raises ABR_X6_FLAG_NONFINITE.  Operand elements more than 18 binades below their tensor's amax
 * (they keep an ABSOLUTE accuracy of 2^-40 amax instead of 2^-22 relative) are ordinary and are COUNTED, not flagged: abr_h3_range_stats. */
#define ABR_H3_FLAG_STALE 8u
int abr_x6_range_flags(uint32_t* out_host, int reset, void* stream);
/* the same word copied to PINNED host memory on `stream` without synchronising (the trainer polls it one step later) */
int abr_x6_range_flags_async(uint32_t* out_pinned_host, void* stream);
/* the same word copied to DEVICE memory on `stream` (data-parallel training MAX-reduces it over the ranks with RCCL before reading it,
 * so that every rank leaves the bf16x6 arithmetic at the same step: engine/trainer.py::_x6_guard) */
int abr_x6_range_flags_to_device(uint32_t* out_device, void* stream);

/* =====================================================================================================
 * 1. maskrcnn_benchmark._C  (csrc/vision.cpp:10-16)
 * ===================================================================================================== */

/* _C.roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio)
 *   csrc/ROIAlign.h:11-25 -> csrc/cuda/ROIAlign_cuda.cu:65-122,258-300 / csrc/cpu/ROIAlign_cpu.cpp:113-256
 * feat [B,C,H,W] (NCHW) or [B,H,W,C] (NHWC); rois [K,5] = (batch_idx,x1,y1,x2,y2) fp32;
 * out  [K,C,PH,PW] (NCHW) or [K,PHo,PWo,C] (NHWC) where PHo=ceil(PH/bin_step).
 * bin_step: 1 = every bin (reference behaviour).  2 = only bins with even (ph,pw): what a following
 *   stride-2 1x1 conv (ResNetHead.layer4.0, modeling/backbone/resnet.py:278) actually reads; NHWC only. */
int abr_roi_align_forward(const float* feat, const float* rois, int K, int B, int C, int H, int W,
                          float spatial_scale, int PH, int PW, int sampling_ratio, int bin_step,
                          int layout, float* out, void* stream);

/* _C.roi_align_backward(grad, rois, spatial_scale, ph, pw, B, C, H, W, sampling_ratio)
 *   csrc/ROIAlign.h:27-45 -> csrc/cuda/ROIAlign_cuda.cu:178-254,304-346 (CPU: "Not implemented").
 * grad_feat is zero-filled by the callee unless accumulate!=0 (reference: at::zeros, :316). */
int abr_roi_align_backward(const float* grad, const float* rois, int K, int B, int C, int H, int W,
                           float spatial_scale, int PH, int PW, int sampling_ratio, int bin_step,
                           int layout, int accumulate, float* grad_feat, void* stream);

/* The float64 instantiations of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES: csrc/cuda/ROIAlign_cuda.cu:283,329, csrc/cpu/ROIAlign_cpu.cpp:242):
 * NCHW tensors, every T of the reference's templates = double.  grad_feat is zero-filled by the callee (at::zeros, :316). */
int abr_roi_align_forward_f64(const double* feat, const double* rois, int K, int B, int C, int H, int W, double spatial_scale, int PH, int PW,
                              int sampling_ratio, double* out, void* stream);
int abr_roi_align_backward_f64(const double* grad, const double* rois, int K, int B, int C, int H, int W, double spatial_scale, int PH, int PW,
                               int sampling_ratio, double* grad_feat, void* stream);

/* Same result as abr_roi_align_backward(layout=ABR_NHWC) without atomics: every feature pixel gathers from the RoIs that
 * cover it (two kernels: per-RoI separable weight tables, then one coalesced write per dFeat element, deterministic order).
 * workspace: abr_roi_align_backward_ws_bytes(...) bytes.  This is the form the training step uses. */
int64_t abr_roi_align_backward_ws_bytes(int K, int B, int H, int W, int PH, int PW, int bin_step);
int abr_roi_align_backward_gather(const float* grad, const float* rois, int K, int B, int C, int H, int W,
                                  float spatial_scale, int PH, int PW, int sampling_ratio, int bin_step, int accumulate,
                                  float* grad_feat, void* workspace, int64_t workspace_bytes, void* stream);

/* Integer tap table of the forward kernel's OWN indexing code (parity instrument, not on the hot path):
 * idx [K,PH*PW,max_s,4] int32 flat y*W+x (-1 = sample rejected, -2 = unused slot), grid [K,2]. */
int abr_roi_align_taps(const float* rois, int K, int H, int W, float spatial_scale, int PH, int PW,
                       int sampling_ratio, int max_s, int32_t* idx, int32_t* grid, void* stream);

/* _C.nms(dets, scores, threshold)   csrc/nms.h:10-27 -> csrc/cpu/nms_cpu.cpp:5-75 / csrc/cuda/nms.cu:70-131
 * Batched, boxes already sorted by descending score (the reference sorts first: nms_cpu.cpp:24, nms.cu:73).
 *   boxes   [N, n_max, 4] xyxy fp32, image i uses its first counts[i] rows
 *   strict_gt : 0 -> suppress when IoU >= thr (CPU rule, nms_cpu.cpp:60; the canonical one here)
 *               1 -> suppress when IoU >  thr (CUDA rule, nms.cu:60)
 *   keep    [N, max_keep] int32 : positions (in sorted order) of survivors, ascending; n_keep [N] int32
 *   The greedy sweep runs ON DEVICE (no 18 MB D2H + host loop as nms.cu:99-123) and stops at max_keep.
 *   workspace: abr_nms_workspace_bytes(N, n_max) bytes. */
int64_t abr_nms_workspace_bytes(int N, int n_max);
int abr_nms_sorted_batched(const float* boxes, const int32_t* counts, int N, int n_max, float thr,
                           int strict_gt, int max_keep, int32_t* keep, int32_t* n_keep,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* _C.nms on UNSORTED boxes in one call (round 5; abr_iod_amd/_C.py::nms): descending stable score sort (abr_sort_scores_desc: the proposal
 * ranking's own kernels on order-preserving keys), greedy suppression as above, survivors' ORIGINAL indices in ascending order (nms.cu:127-130).
 *   dets [n,4] xyxy fp32 (16-byte aligned), scores [n] fp32, n <= abr_sort_scores_max_n() (15360); keep_out [n] int64, *n_keep_out int32 (device);
 *   workspace: abr_nms_unsorted_workspace_bytes(n) bytes. */
int64_t abr_nms_unsorted_workspace_bytes(int n);
int abr_nms(const float* dets, const float* scores, int n, float thr, int strict_gt, int64_t* keep_out, int32_t* n_keep_out, void* workspace,
            int64_t workspace_bytes, void* stream);
/* order[i] = index of the i-th largest score (ties: ascending index -- torch.sort(descending=True, stable=True)); any finite fp32 */
int64_t abr_sort_scores_max_n(void);
int abr_sort_scores_desc(const float* scores, int n, int64_t* order, void* stream);

/* _C.sigmoid_focalloss_forward / _backward   csrc/SigmoidFocalLoss.h:10-41 -> csrc/cuda/SigmoidFocalLoss_cuda.cu:20-101
 * logits [N,C] fp32, targets [N] int32 (0 = background, -1 = ignore, c>=1 = class c), losses/d_logits [N,C]. */
int abr_sigmoid_focal_forward(const float* logits, const int32_t* targets, int N, int C, float gamma,
                              float alpha, float* losses, void* stream);
int abr_sigmoid_focal_backward(const float* logits, const int32_t* targets, const float* d_losses, int N,
                               int C, float gamma, float alpha, float* d_logits, void* stream);
/* the float64 instantiations (AT_DISPATCH_FLOATING_TYPES, SigmoidFocalLoss_cuda.cu:128,172): T = double with the template's own float gamma /
 * alpha and expf / powf / logf calls */
int abr_sigmoid_focal_forward_f64(const double* logits, const int32_t* targets, int N, int C, float gamma, float alpha, double* losses, void* stream);
int abr_sigmoid_focal_backward_f64(const double* logits, const int32_t* targets, const double* d_losses, int N, int C, float gamma, float alpha,
                                   double* d_logits, void* stream);

/* =====================================================================================================
 * 2. Distillation + detector losses (ATen elementwise/reduction chains in the reference)
 *    Every loss kernel writes its scalar to loss_out[0] (device, loss_out must hold >= 4 floats) and, when grad pointers are non-NULL,
 *    the gradient for upstream-gradient `gscale` in the same launch sequence.
 * ===================================================================================================== */

/* calculate_attentive_roi_feature_distillation(f_map_s=SOURCE, f_map_t=TARGET, gamma)
 *   maskrcnn_benchmark/distillation/distillation.py:86-130, call site tools/train_incremental.py:115.
 * f_src,f_tgt [N,HW,C] (NHWC) or [N,C,HW] (NCHW).
 * coef [N,2,HW] fp32 scratch: a_src and dPad/dm_tgt per position, written by forward, read by backward.
 * loss_out[0] = afd + gamma*pad ; loss_out[1] = afd ; loss_out[2] = pad. */
int abr_ard_forward(const float* f_src, const float* f_tgt, int N, int C, int HW, float gamma, int layout,
                    float* coef, float* loss_out, void* stream);
/* grad_tgt = gscale * d(afd+gamma*pad)/d f_tgt ; gscale_dev (optional device scalar) multiplies gscale. */
int abr_ard_backward(const float* f_src, const float* f_tgt, const float* coef, int N, int C, int HW,
                     float gamma, int layout, float gscale, const float* gscale_dev, float* grad_tgt,
                     void* stream);

/* smooth_l1_loss(input, target, beta, size_average)   maskrcnn_benchmark/layers/smooth_l1_loss.py:6-17
 * Optional row gather: when rows!=NULL only rows[i] (int64) of x/t take part and, per row, the 4 columns
 * starting at col0[i] of x (box_head/loss.py:166-172 advanced indexing); x_cols = row length of x.
 * loss_out[0] = sum * scale.  grad (optional, same shape as x, must be pre-zeroed when rows!=NULL). */
int abr_smooth_l1(const float* x, const float* t, int64_t n, float beta, float scale, float* loss_out,
                  float gscale, float* grad, void* stream);
int abr_smooth_l1_rows(const float* x, int x_cols, const float* t, const int64_t* rows, const int64_t* col0,
                       const int64_t* trows /* rows of t, NULL = rows */, int n_rows, float beta, float scale,
                       const float* denom_dev /* optional device scalar: scale /= max(*denom_dev,1); rows < 0 are skipped */,
                       float* loss_out, float gscale, float* grad, void* stream);

/* FastRCNNLossComputation classification term   modeling/roi_heads/box_head/loss.py:151-162
 *   inclusive != 0 : "Inclusive Classification Loss" (dist_type=='id'), n_old = number of old classes
 *   else           : F.cross_entropy.   labels int64 [n]; label<0 rows are ignored as F.nll_loss does (-100)
 * loss_out[0] = mean over counted rows.  d_logits optional [n,K]. */
int abr_softmax_ce(const float* logits, int ld_logits /* row pitch, <=0: K */, const int64_t* labels, int n, int K,
                   int inclusive, int n_old, float* loss_out, float gscale, float* d_logits, int ld_dlogits,
                   void* stream);

/* calculate_feature_distillation_loss(..., loss='normalized_filtered_l1')  distillation.py:133-161 (ablation DIST.FEAT='std'), one
 * feature level: loss = mean(max((s - mean s) - (t - mean t), 0)) over all n elements (any memory order).  stats3: 3-float scratch.
 * d_tgt (optional, n floats) = gscale * dloss/dt. */
int abr_feat_distill(const float* src, const float* tgt, int64_t n, float* loss_out, float* stats3, float gscale, float* d_tgt,
                     void* stream);
/* calculate_rpn_distillation_loss(cls_loss='filtered_l2', bbox_loss='l2'|'None', bbox_threshold)  distillation.py:18-84 (ablation
 * DIST.RPN), one level, NHWC head outputs: anchor a of row r has objectness obj[r*ld_obj + a] and deltas reg[r*ld_reg + 4a .. +3].
 * loss = mean_anchors max(o_s-o_t,0)^2 + mean_anchors [o_s-o_t > thr] sum_4 (d_s-d_t)^2.  d_obj_t / d_reg_t optional (same pitches
 * ld_d_obj / ld_d_reg), = gscale * dloss/d(target). */
int abr_rpn_distill(const float* obj_s, const float* reg_s, int ld_obj_s, int ld_reg_s, const float* obj_t, const float* reg_t,
                    int ld_obj_t, int ld_reg_t, int64_t rows, int A, float bbox_threshold, int use_bbox, float* loss_out,
                    float gscale, float* d_obj_t, float* d_reg_t, int ld_d_obj, int ld_d_reg, void* stream);

/* calculate_roi_distillation_losses(soften, target, dist)   distillation/distillation.py:164-240
 *   dist_id!=0 -> unbiased cross-entropy + L2 boxes ; else mean-centred L2 + L2 boxes
 * z_s [n,K_old], b_s [n,K_old,4], z_t [n,K_all], b_t [n,K_all,4]; d_zt/d_bt optional. */
/* ld_host: optional 6 row pitches (z_s, b_s, z_t, b_t, d_zt, d_bt) in floats, <=0 = dense; lets the scores / boxes
 * be column slices of the fused predictor output. */
int abr_roi_distill(const float* z_s, const float* b_s, const float* z_t, const float* b_t, int n, int K_old,
                    int K_all, const int32_t* ld_host, int dist_id, float* loss_out, float gscale, float* d_zt,
                    float* d_bt, void* stream);

/* F.binary_cross_entropy_with_logits(x[idx], y[idx]).mean()   modeling/rpn/loss.py:145-146
 * idx int64 [n_idx] into the flattened logits; grad optional, pre-zeroed, same shape as x. */
int abr_bce_logits_gather(const float* x, const float* y, const int64_t* idx, const int64_t* yidx /* NULL = idx */,
                          int n_idx, const float* denom_dev /* optional device scalar replacing n_idx as the mean's
                          denominator; idx < 0 entries are skipped (fixed-size, -1 padded index lists) */,
                          float* loss_out, float gscale, float* grad, void* stream);

/* =====================================================================================================
 * 3. Convolution as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), NHWC / OHWI.
 *    Replaces cuDNN fwd/dgrad/wgrad + FrozenBatchNorm2d + ReLU + residual add of
 *    modeling/backbone/resnet.py:327-346,363-368, layers/batch_norm.py:19-31, modeling/rpn/rpn.py:114-121,
 *    roi_box_predictors.py:27-32 (Linear = 1x1 conv on a 1x1 map).
 * ===================================================================================================== */
#define ABR_MATH_F32 0
#define ABR_MATH_BF16 1
#define ABR_MATH_BF16X6 2
#define ABR_MATH_F16X3 3

typedef struct {
    int B, H, W, Cin;        /* input  [B,H,W,Cin]  (Cin % 4 == 0) */
    int Cout, R, S;          /* weight [Cout,R,S,Cin] (OHWI)       */
    int stride, pad;
    int Ho, Wo;              /* output spatial size                 */
    /* epilogue: y = acc*scale[c] + bias[c] (+ residual) ; relu ; y *= (mask>0) */
    const float* scale;      /* [Cout] or NULL (=1)                 */
    const float* bias;       /* [Cout] or NULL (=0)                 */
    const float* residual;   /* same geometry as out, or NULL       */
    const float* mask;       /* same geometry as out, or NULL: multiply by (mask>0) (ReLU backward) */
    int relu;
    /* output placement: row (b,ho,wo) is stored at pixel (b, ho*out_sh, wo*out_sw) of [B,out_H,out_W,Cout]
       (out_sh=out_sw=1, out_H=Ho, out_W=Wo for an ordinary conv; 2 for the dgrad of a stride-2 1x1 conv) */
    int out_H, out_W, out_sh, out_sw;
    /* arithmetic of the contraction: ABR_MATH_F32 = fp32 MFMA (exact fp32, the default and the parity path);
       ABR_MATH_BF16 = both operands rounded to bf16 (RNE) inside the kernel, bf16 MFMA, fp32 accumulate, fp32 tensors in
       memory (BASELINE.json configs[4]); layers whose Cin is not a multiple of 64 (the stem) compute in fp32 either way;
       ABR_MATH_BF16X6 = fp32-ACCURATE arithmetic on the bf16 matrix cores: each fp32 operand split exactly into three bf16
       terms, the six cross products with i + j <= 2 accumulated in fp32 (same error bound as an fp32 FMA chain; opt-in) */
    int math;
    /* Winograd-domain input reuse between a convolution's forward pass and its weight gradient (both transform the SAME input
       with B^T d B).  abr_conv_forward: if wino_v is non-NULL and the conv takes the Winograd path, the transformed input V
       (abr_conv_wino_v_floats floats) is written THERE instead of scratch.  abr_conv_wgrad: if wino_v is non-NULL and the
       gradient takes the Winograd path, V is read from there and the input transform is skipped (x is not touched).  NULL =
       self-contained calls. */
    float* wino_v;
    /* ABR_MATH_BF16X6, abr_conv_forward only: the weights as FRAGMENT-PACKED bf16x3 planes made by abr_conv_pack_weights(w, Cout,
       R*S*Cin) (abr_conv_packed_bytes bytes): the weights-direct kernel then loads every weight fragment straight from these planes
       into the matrix-core registers -- no per-workgroup split, no LDS traffic for the weights.  NULL with w_version != 0: the library
       packs (w, w_version) itself on first use and keeps the planes; NULL with w_version == 0: the weight tile is split in every workgroup. */
    const void* w_planes;
    /* abr_conv_forward: non-zero = "the weight tensor at address w has not changed since the last call that passed this same
       (w, w_version) pair": the library then keeps data derived from it -- the Winograd-domain weights U = G g G^T (36*Cout*Cin
       floats, or their packed bf16x3 planes under ABR_MATH_BF16X6) and the packed bf16x3 planes of w -- and skips the derivation on later calls (the frozen source model: once; a trainable conv: once per optimiser
       step, shared by its forward passes).  0 = derive it again on every call. */
    int64_t w_version;
    /* ABR_MATH_F16X3 (round 5): fp32-ACCURATE contractions with THREE products per multiply-add.  Each operand is written x = s (h0 + h1) with
       h0 = fp16(x / s), h1 = fp16(x / s - h0) and s = the power of two that puts the operand's largest magnitude in [2^14, 2^15) -- per tensor
       for activations / gradients, per output channel for weights (folded into the epilogue) -- and x w = s_x s_w (h0 g0 + h0 g1 + h1 g0) runs on
       v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Representation error <= 2^-22 |x| for |x| >= 2^-18 amax (2^-40 amax below), random-signed:
       the result meets the fp32-MFMA error bound on tensors whose reductions are not made of such elements only (tests/test_gpu_f16x3_admission.py).
       The amax of an activation tensor lives in an AMAX WORD (abr_h3_amax_alloc): the producing kernel writes it (out_amax: any abr_conv_forward,
       whatever its math), the consuming kernel reads it (x_amax; gy_amax for abr_conv_wgrad).  A NULL x_amax / gy_amax makes the call reduce the
       operand itself first (one extra pass over it: correct, slower). */
    const uint64_t* x_amax;   uint32_t x_amax_epoch;
    const uint64_t* gy_amax;  uint32_t gy_amax_epoch;
    uint64_t* out_amax;       uint32_t out_amax_epoch;
} abr_conv_desc;

/* An amax word of the library's zero-initialised device ring and the epoch to use with it (see abr_conv_desc).  A word is handed out again after
 * ABR_H3_AMAX_RING allocations; by then its tensor must be gone (a reader naming an older epoch raises ABR_H3_FLAG_STALE). */
#define ABR_H3_AMAX_RING 65536
int abr_h3_amax_alloc(uint64_t** word_out, uint32_t* epoch_out);
/* n consecutive allocations at once (a host wrapper hands them out itself: one library call per n tensors instead of one per tensor):
 * allocation k of the block (0 <= k < n) is word ring_base + (first_count + k) % ABR_H3_AMAX_RING with epoch (uint32_t)(first_count + k + 1). */
int abr_h3_amax_alloc_block(int n, uint64_t** ring_base_out, uint64_t* first_count_out);
/* Range statistics of the f16x3 kernels since the last reset: out_host[0] = operand elements seen more than 18 binades below their tensor's amax
 * (non-zero), out_host[1] = operand elements inspected (every element of an activation / gradient operand once per GEMM).  Synchronises `stream`. */
int abr_h3_range_stats(uint64_t* out_host, int reset, void* stream);
/* the same two numbers written to DEVICE memory on `stream` without synchronising (a training loop copies them out asynchronously, after a SUM
 * over the ranks under data parallelism: engine/trainer.py::_x6_guard) */
int abr_h3_range_stats_to_device(uint64_t* out_device, int reset, void* stream);
/* *word = (epoch << 32) | bits(max |x[i]|) for n floats on `stream` (what a producer kernel's epilogue writes for free) */
int abr_h3_amax(const float* x, int64_t n, uint64_t* word, uint32_t epoch, void* stream);

int abr_conv_forward(const abr_conv_desc* d_host, const float* x, const float* w, float* out, void* stream);
/* The tail of a 64-wide bottleneck WITHOUT a backward pass (maskrcnn_benchmark/modeling/backbone/resnet.py:327-346 for the frozen layer1:
 * conv2 3x3 -> bn2 -> relu -> conv3 1x1 -> bn3 -> += identity -> relu) as ONE launch: d2 = the 3x3 conv (64 -> 64, stride 1, pad 1, scale / bias /
 * relu as in abr_conv_forward), d3 = the 1x1 conv (64 -> 256, scale / bias / residual / relu), both ABR_MATH_BF16X6 with w_version != 0 (or
 * caller-packed w_planes).  out [B,H,W,256] is bit-identical to abr_conv_forward(d2) followed by abr_conv_forward(d3); the intermediate tensor
 * never leaves the compute unit. */
int abr_conv_tail64_forward(const abr_conv_desc* d2_host, const abr_conv_desc* d3_host, const float* x, const float* w2, const float* w3, float* out,
                            void* stream);
/* Derive, on `stream`, whatever abr_conv_forward would derive from the weight tensor (w, w_version != 0) of a conv with this geometry and
 * arithmetic -- the Winograd-domain weights of a wide stride-1 3x3 conv, the packed bf16x3 planes of any bf16x6 conv -- so that the next abr_conv_forward with the same
 * (w, w_version) finds it ready (a consumer on another stream is ordered behind it by the library).  Lets a
 * caller move the per-step weight preparation of its trainable convs off the critical stream (solver/build.py). */
int abr_conv_prepare_weights(const float* w, int Cout, int R, int S, int Cin, int stride, int pad, int math, int64_t w_version, void* stream);
/* The same for a whole model in a handful of launches.  Item i: what abr_conv_prepare_weights derives from w; and, when wt != NULL, first the
 * dgrad copy wt = abr_conv_dgrad_weights(w, scale) and then what abr_conv_prepare_weights(wt, Cin -> Cout swapped, stride 1, pad R-1-pad) derives
 * from THAT.  All transposes go out as one launch, all Winograd weight transforms as one, all bf16x3 packings as one (a model's ~190 per-tensor
 * launches after every optimiser step were a host-paced train of tiny kernels and ~3 ms of host time).  Results and cache entries are
 * the ones the per-tensor calls produce.  Items whose shape the batched kernels do not take (Cin % 4 != 0 Winograd weights) go the per-tensor way. */
typedef struct abr_prep_item {
    const float* w;       /* [Cout][R][S][Cin] */
    const float* scale;   /* FrozenBN scale folded into the dgrad copy, or NULL */
    float* wt;            /* [Cin][R][S][Cout] dgrad copy (flipped taps), or NULL: forward derivation only */
    int32_t Cout, R, S, Cin, stride, pad, math;
    int64_t w_version;
} abr_prep_item;
int abr_conv_prepare_batch(const abr_prep_item* items_host, int n, void* stream);

/* A table of conv calls in ONE library call (round 5: the host side of a bottleneck's forward or backward pass is four to ten of the calls above,
 * ~12 us of interpreter and binding time each; a host wrapper keeps the table of a (block, input shape) with everything that does not change from
 * step to step filled in and writes only this step's pointers).  Ops run in table order, each on ITS stream:
 *   ABR_OP_FORWARD      abr_conv_forward(&desc, a = x, b = w, out, stream)
 *   ABR_OP_WGRAD        abr_conv_wgrad(&desc, a = x, b = gy, out = dw, stream)
 *   ABR_OP_STREAM_WAIT  abr_stream_wait_stream(stream, other)            -- `stream` waits for what `other` has queued so far
 * Stops at the first failing op and returns its status (abr_last_error names it).  The reference's counterpart: one cuDNN call per conv from
 * ATen, driven by the Python of modeling/backbone/resnet.py:327-346. */
#define ABR_OP_FORWARD 0
#define ABR_OP_WGRAD 1
#define ABR_OP_STREAM_WAIT 2
typedef struct abr_conv_op {
    int32_t kind;
    int32_t reserved;
    abr_conv_desc desc;
    const float* a;
    const float* b;
    float* out;
    void* stream;
    void* other;
} abr_conv_op;
int abr_conv_run(const abr_conv_op* ops, int n);
/* The library keeps this derived data per (weight address, kind, w_version) -- 36/9 of each wide 3x3 weight, 1.5x of every other bf16x6 weight.  The cache is bounded
 * (least-recently-used entries go when it exceeds ABR_WINO_CACHE_MB, default 4096); abr_conv_cache_clear drops every entry after waiting
 * for the streams that use them (call it when a model's parameter storage is released or rebuilt), abr_conv_cache_bytes reports its size. */
int abr_conv_cache_clear(void);
/* the same for the entries derived from tensors that live inside [base, base + bytes) only (a model's flat parameter storage that is being
 * released): entries of other models -- and conv calls in flight on other host threads that hold them -- are not touched.  Both calls skip
 * entries whose fill is in progress on another thread. */
int abr_conv_cache_drop_range(const void* base, int64_t bytes);
int64_t abr_conv_cache_bytes(void);
/* floats of the Winograd-domain input V = 36 * B*ceil(H/4)*ceil(W/4) * Cin if BOTH abr_conv_forward and abr_conv_wgrad take the
 * Winograd F(4x4,3x3) path for this descriptor (wide stride-1 pad-1 3x3, no residual / scatter, fp32 or bf16x6 math), else 0 */
int64_t abr_conv_wino_v_floats(const abr_conv_desc* d_host);
/* Fragment-packed bf16x3 planes of an fp32 matrix w [rows][K] (K % 16 == 0; a conv weight: rows = Cout, K = R*S*Cin) for
 * abr_conv_desc::w_planes.  Exact three-way split x = p0 + p1 + p2 (p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1), RNE), stored in
 * the order the bf16 MFMA consumes it: chunk (nb, ks, pl) = 64 lanes x 16 B at byte (((nb * K/16 + ks) * 3 + pl) * 64 + lane) * 16, lane l
 * holding row nb*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +8 of plane pl; rows are zero-padded to a multiple of 32.  The elements are
 * range-checked for the bf16x6 arithmetic as they are packed (abr_x6_range_flags). */
int64_t abr_conv_packed_bytes(int64_t rows, int64_t K);
int abr_conv_pack_weights(const float* w, int64_t rows, int K, void* planes, void* stream);

/* dW[Cout,R,S,Cin] (+)= sum_m gy[m,Cout]^T * im2col(x)[m,RSCin], columns scaled by d->scale (FrozenBN).
 * Accumulates with fp32 atomics into dw (caller zeroes it once per step). */
int abr_conv_wgrad(const abr_conv_desc* d_host, const float* x, const float* gy, float* dw, void* stream);

/* w [Cout,R,S,Cin] -> wt [Cin,R,S,Cout] spatially flipped and scaled by scale[Cout]: the weight tensor
 * that turns dgrad into abr_conv_forward(gy, wt). */
int abr_conv_dgrad_weights(const float* w, const float* scale, int Cout, int R, int S, int Cin, float* wt,
                           void* stream);
/* db[c] += sum_m gy[m,c] (bias gradient of nn.Conv2d / nn.Linear) */
int abr_bias_grad(const float* gy, int64_t M, int C, float* db, void* stream);

/* =====================================================================================================
 * 4. Pointwise / pooling / layout
 * ===================================================================================================== */
int abr_nchw_to_nhwc_pad(const float* x, int B, int C, int H, int W, int Cpad, float* out, void* stream);
int abr_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* out, void* stream);
int abr_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* out, void* stream);
int abr_maxpool3x3s2(const float* x, int B, int H, int W, int C, float* out, void* stream); /* resnet.py:367 */
/* AdaptiveAvgPool2d(1)  roi_box_predictors.py:28 : x [N,HW,C] -> out [N,C] ; backward spreads g/HW */
int abr_avgpool_forward(const float* x, int N, int HW, int C, float* out, void* stream);
int abr_avgpool_backward(const float* g, int N, int HW, int C, float* gx, void* stream);
/* the same fused with the backward of the ReLU that produced the pooled tensor y [N, HW, C]: gx = y > 0 ? g / HW : 0 */
int abr_avgpool_relu_backward(const float* g, const float* y, int N, int HW, int C, float* gx, void* stream);
/* ... which also writes max |gx| into an amax word (abr_h3_amax_alloc) for the f16x3 convs that consume gx; amax == NULL: as above */
int abr_avgpool_relu_backward_amax(const float* g, const float* y, int N, int HW, int C, float* gx, uint64_t* amax, uint32_t amax_epoch, void* stream);
/* out[row] = mean_c x[row, c] for x [rows, C] -- per-(RoI, bin) channel mean of NHWC pooled features
 * (tools/prototype_box_selection.py:84 `torch.mean(roi_align_features, dim=1)`) */
int abr_channel_mean(const float* x, int64_t rows, int C, float* out, void* stream);
/* out = g * (y > 0)  (ReLU backward); out == g is allowed (in place) */
int abr_relu_backward(const float* g, const float* y, int64_t n, float* out, void* stream);
int abr_add_inplace(float* a, const float* b, int64_t n, void* stream);
/* The training loop's loss arithmetic (tools/train_incremental.py:91,101-128: sum of the detector losses, alpha * ID + beta * ARD, their sum)
 * in one launch: terms_host [n <= 8] HOST array of DEVICE pointers to the scalar losses, weights_host / groups_host [n] host arrays
 * (group 0 = detector losses, 1 = distillation); *total = out[0] = sum w_i l_i, out[1] / out[2] = the two groups' sums.  The backward hands
 * every term its gradient w_i * *g (g: device scalar) in one launch. */
int abr_loss_sum(const float* const* terms_host, const float* weights_host, const int32_t* groups_host, int n, float* total, float* out,
                 void* stream);
int abr_loss_sum_backward(const float* weights_host, int n, const float* g, float* grads, void* stream);
/* x *= s * (s_dev ? *s_dev : 1): applies an upstream (device-resident) loss gradient without a host sync */
int abr_scale_inplace(float* x, int64_t n, float s, const float* s_dev, void* stream);

/* =====================================================================================================
 * 5. RPN / RoI-head glue (integer + fp32 index work; maskrcnn_benchmark/modeling/rpn/, matcher.py, box_coder.py)
 * ===================================================================================================== */
/* AnchorGenerator grid  anchor_generator.py:84-110 : out [H*W*A,4], vis [H*W*A] uint8 */
int abr_grid_anchors(const float* cell, int A, int H, int W, int stride, int img_h, int img_w, int straddle,
                     float* out, uint8_t* vis, void* stream);
/* sigmoid + top-k (sorted, descending; ties by ascending index) of the RPN objectness, per image, in one launch
 * (rpn/inference.py:87-96).  Anchor j of image i is logits[i*img_stride + (j/A)*ld + j%A] (A=15, ld=76 on the fused NHWC head
 * output; A=1, ld=1 for a plain [N,n] matrix).  scores [N,k] fp32, idx [N,k] int64.  k <= 15360. */
int abr_topk_sigmoid(const float* logits, int64_t img_stride, int N, int n, int A, int ld, int k, float* scores,
                     int64_t* idx, void* stream);
/* Test-time PostProcessor, stage 1 (roi_heads/box_head/inference.py:55-70, box_coder.py:52-95, bounding_box.py:214-225):
 * prob = softmax(logits[:, :C]); boxes[r,j] = clip_to_image(decode(deltas[r, delta_col0 + 4j .. +3], rois[r,1:5])).
 * rois [K,5] = (image index, x1,y1,x2,y2); img_hw [N,2] int32 (h,w).  agnostic_col >= 0 (CLS_AGNOSTIC_BBOX_REG): every class
 * reads the 4 columns at delta_col0 + agnostic_col instead.  prob [K,C], boxes [K,C,4]. */
int abr_det_softmax_decode(const float* logits, int ld_logits, const float* deltas, int ld_deltas, int delta_col0,
                           int agnostic_col, const float* rois, int K, int C, const int32_t* img_hw, float wx, float wy,
                           float ww, float wh, float* prob, float* boxes, void* stream);
/* Test-time PostProcessor, stage 2 = filter_results (inference.py:106-151) for the whole batch: per image and class
 * score > score_thresh, sort, NMS (`>=` as the CPU _C.nms), classes 1..C-1 concatenated, and when more than
 * detections_per_img (> 0) remain only those with score >= the detections_per_img-th largest (ties kept, order kept).
 * row_offsets [N+1] int32 (device): rows of image i in prob/boxes; r_max >= every image's row count (<= 16384).
 * out_* [N,cap,...] (cap = (C-1)*r_max holds every case), out_labels int64, out_count [N] int32.
 * bg_* [N,r_max,...] / bg_count [N]: class 0's NMS survivors (the reference's `results_background`); NULL to skip. */
int64_t abr_det_select_workspace_bytes(int N, int C, int r_max);
int abr_det_select(const float* prob, const float* boxes, const int32_t* row_offsets, int N, int C, int r_max,
                   float score_thresh, float nms_thresh, int detections_per_img, int cap, float* out_boxes,
                   float* out_scores, int64_t* out_labels, int32_t* out_count, float* bg_boxes, float* bg_scores,
                   int32_t* bg_count, void* workspace, int64_t workspace_bytes, void* stream);
/* BoxCoder.decode + clip_to_image on gathered rows  (rpn/inference.py:96-112, box_coder.py:52-95, bounding_box.py:214-225)
 * For image i and rank j<k: a = idx[i,j]; out[i,j,:] = clip(decode(reg[i,a,:], anchors[a,:])).
 * reg [N,n_anchor/A,reg_stride]: anchor a = loc*A + a' reads columns reg_col0 + 4a' .. +3 of row loc
 * (A=1: one row per anchor; A=15, reg_stride=76, reg_col0=15: the fused NHWC RPN head output). */
int abr_rpn_decode_clip(const float* reg, int reg_stride, int reg_col0, int A, const float* anchors,
                        const int64_t* idx, int N, int n_anchor, int k, const int32_t* img_hw, float wx,
                        float wy, float ww, float wh, float* out, void* stream);
/* BoxCoder.encode row-wise (box_coder.py:22-50): out[i] = encode(gt[i], ex[i]); img_hw==NULL above = decode without clip */
int abr_box_encode(const float* gt, const float* ex, int n, float wx, float wy, float ww, float wh, float* out,
                   void* stream);
/* boxlist_iou + Matcher (+ RPN labels / box-head labels) + BoxCoder.encode in one pass per box.
 *   boxes [n,4], gt [G,4], gt_labels [G] int64 (or NULL for RPN), vis [n] uint8 (or NULL)
 *   matched [n] int64 (-1 / -2 / gt index), labels_out [n] (fp32 for RPN: 1/0/-1; int64 for head: class/0/-1),
 *   reg_targets [n,4].   matcher.py:42-112, rpn/loss.py:66-102, box_head/loss.py:56-84 */
int abr_match_encode(const float* boxes, int n, const float* gt, const int64_t* gt_labels, int G,
                     const uint8_t* vis, float hi, float lo, int allow_low_quality, float wx, float wy,
                     float ww, float wh, int64_t* matched, float* labels_f32, int64_t* labels_i64,
                     float* reg_targets, void* workspace, int64_t workspace_bytes, void* stream);
int64_t abr_match_workspace_bytes(int n, int G);

/* BalancedPositiveNegativeSampler (balanced_positive_negative_sampler.py:19-77) for N images in one launch, no host sync:
 * labels [N, n] (row pitch `stride`; fp32 for the RPN, int64 for the box head): >=1 positive, ==0 negative, else ignored.
 * Draws min(#pos, max_pos) positives and min(#neg, batch_size - that) negatives uniformly at random (counter-based keys from
 * `seed`, image first_image+i, index) and writes them ASCENDING: pos_idx [N, max_pos], neg_idx [N, batch_size], padded
 * with -1; every index has i*index_offset_per_image added (batch-flattened indices); counts [N,2] = (#pos, #neg) taken. */
int abr_sample_pos_neg(const void* labels, int labels_are_int64, int N, int n, int64_t stride, int batch_size,
                       int max_pos, uint64_t seed, int first_image, int64_t index_offset_per_image,
                       int64_t* pos_idx, int64_t* neg_idx, int32_t* counts, void* stream);

/* Box-head training targets of a whole batch in three launches and NO host round trip -- what RPNPostProcessor.add_gt_proposals
 * (rpn/inference.py:53-74) + FastRCNNLossComputation.subsample (box_head/loss.py:56-120: match, label, encode, sample, cut) +
 * Pooler.convert_to_roi_format (poolers.py:73-86) do per image with a dozen small ops and two device synchronisations each:
 *   props [N,k_pre,4] decoded score-sorted boxes, keep [N,post] / n_keep [N] = abr_nms_sorted_batched's output (device);
 *   gt_ptrs / gt_label_ptrs [N] device arrays of device pointers to each image's GT boxes [G_i,4] / labels [G_i] int64, n_gt [N];
 *   candidates of image i = kept proposals then GT (Pmax = post + g_max rows per image):
 *     cand [N,Pmax,4], labels_all [N,Pmax] int64 (rows past the count = -1), regt_all [N,Pmax,4], n_cand [N];
 *   sampler (abr_sample_pos_neg on labels_all): pos_idx [N,max_pos], neg_idx [N,batch_size], counts [N,2];
 *   sampled rows, ascending candidate order per image, batch_size rows per image (rows past the number drawn: label -1, zero box):
 *     rois [N*batch_size,5] = (image, x1,y1,x2,y2), labels [N*batch_size], reg_targets [N*batch_size,4], sampled_idx [N,batch_size];
 *   obj [N*batch_size] the rows' objectness (scores [N,k_pre] for proposals, 1 for GT; obj_all [N,Pmax] scratch);
 *   pos_rows [N*batch_size] = row index where label > 0 else -1, col0 = num_classes + 4*label (or + 4 when cls_agnostic): the rows
 *   and columns of the fused predictor output that enter the box-regression loss (box_head/loss.py:166-171);
 *   n_valid [1] fp32 = total number of drawn rows (the denominator of the box-regression loss, box_head/loss.py:179). */
int abr_roi_head_targets(const float* props, const int32_t* keep, const int32_t* n_keep, int N, int k_pre, int post,
                         const float* const* gt_ptrs, const int64_t* const* gt_label_ptrs, const int32_t* n_gt, int g_max, float hi,
                         float lo, float wx, float wy, float ww, float wh, int batch_size, int max_pos, uint64_t seed, float* cand,
                         int64_t* labels_all, float* regt_all, int32_t* n_cand, int64_t* pos_idx, int64_t* neg_idx, int32_t* counts,
                         float* rois, int64_t* labels, float* reg_targets, int64_t* sampled_idx, float* n_valid, const float* scores,
                         float* obj_all, float* obj, int64_t* pos_rows, int64_t* col0, int num_classes, int cls_agnostic, void* stream);
/* RPN training targets of a whole batch (rpn/loss.py:66-102 per image: boxlist_iou, Matcher with low-quality matches, labels with the
 * visibility / between-thresholds discards, BoxCoder.encode) in two launches: anchors [n,4] shared by the images, gt_ptrs / vis_ptrs [N]
 * device arrays of device pointers ([G_i,4] fp32 / [n] uint8), n_gt [N]; labels [N,n] fp32 (1 / 0 / -1), reg_targets [N,n,4];
 * workspace >= N*g_max*4 bytes. */
int abr_rpn_targets_batched(const float* anchors, int n, int N, const float* const* gt_ptrs, const int32_t* n_gt, int g_max,
                            const uint8_t* const* vis_ptrs, float hi, float lo, float wx, float wy, float ww, float wh, float* labels,
                            float* reg_targets, void* workspace, int64_t workspace_bytes, void* stream);
/* RPN loss bookkeeping in one launch: pos [n_pos] / neg [n_neg] = abr_sample_pos_neg's batch-flattened, -1 padded lists, counts [n_img,2];
 * samp [n_pos+n_neg] = their concatenation, obj_flat = position of each sampled anchor's objectness logit in the fused NHWC head output
 * ([rows, Cf], anchor j -> row j / A, column j % A), pos_row / pos_col [n_pos] = row and first delta column (A + 4 (j % A)) of the
 * positives, denom [1] fp32 = number of sampled anchors (rpn/loss.py:136,146). */
int abr_rpn_loss_indices(const int64_t* pos, int n_pos, const int64_t* neg, int n_neg, const int32_t* counts, int n_img, int A, int Cf,
                         int64_t* samp, int64_t* obj_flat, int64_t* pos_row, int64_t* pos_col, float* denom, void* stream);
/* rois [N*P,5] = (i, props[i, keep[i, picks[i*P+j]]]), obj [N*P] (or NULL) = the matching scores: the P picked distillation
 * proposals per image of the source model (generalized_rcnn.py:140-158) straight from the NMS output */
int abr_gather_proposals(const float* props, const float* scores, const int32_t* keep, int N, int k_pre, int post,
                         const int64_t* picks, int P, float* rois, float* obj, void* stream);

/* =====================================================================================================
 * 6. Optimiser (solver/build.py:7-21, torch.optim.SGD semantics, one fused launch over all tensors)
 *    p,g,m flat fp32 buffers of `total` elements; seg_end[i] = exclusive end offset of tensor i;
 *    per-tensor lr[i], wd[i] (host arrays copied by the call).  m = mu*m + (g + wd*p); p -= lr*m.
 *    gscale multiplies g first (1/world_size after an RCCL sum all-reduce).
 * ===================================================================================================== */
int abr_sgd_momentum(float* p, const float* g, float* m, int64_t total, const int64_t* seg_end_dev,
                     const float* lr_dev, const float* wd_dev, int n_seg, float momentum, float gscale,
                     int first_step, void* stream);

/* =====================================================================================================
 * 7. ABR data path pixel work: uint8 HWC RGB images resident on the device.
 *    voc_abr.py:512-816 (box-crop resize, mixup blend, mosaic paste), transforms.py:64-165 (Resize, flip, ToTensor, Normalize),
 *    image_list.py:57-70 (zero-padded batch).  All results are bit-identical to the Pillow / numpy / torch-CPU originals.
 * ===================================================================================================== */
/* Pillow's 8-bit separable resampler (Image.resize).  bounds_* [out,2] int32 = (first tap, tap count), coeffs_* [out,ksize] int32 =
 * 22-bit fixed-point weights, both computed by the host exactly as Resample.c precompute_coeffs does (abr_iod_amd/data/resample.py);
 * horizontal pass first; tmp [H,OW,3] is needed when both sizes change. */
int abr_img_resample_u8(const uint8_t* src, int H, int W, uint8_t* dst, int OH, int OW, const int32_t* bounds_h,
                        const int32_t* coeffs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeffs_v, int ksize_v,
                        uint8_t* tmp, void* stream);
/* img[y0:y0+rh, x0:x0+rw] = (uint8)(lam*img[...] + (1-lam)*crop[off_y:off_y+rh, off_x:off_x+rw]) in float64 (voc_abr.py:664-683) */
int abr_img_blend_paste_u8(uint8_t* img, int H, int W, const uint8_t* crop, int CH, int CW, int x0, int y0, int rw, int rh,
                           int off_x, int off_y, double lam, void* stream);
/* dst[dy:dy+rh, dx:dx+rw] = src[sy:sy+rh, sx:sx+rw]  (mosaic tiles, voc_abr.py:765) */
int abr_img_copy_rect_u8(uint8_t* dst, int DH, int DW, const uint8_t* src, int SH, int SW, int dx, int dy, int sx, int sy,
                         int rw, int rh, void* stream);
int abr_img_fill_u8(uint8_t* dst, int64_t n, int value, void* stream);
/* ColorJitter (transforms.py:132-150 -> torchvision.transforms.ColorJitter -> Pillow's ImageEnhance / HSV conversions), one op in place:
 * op 0 brightness, 1 contrast, 2 saturation: img = Image.blend(degenerate, img, factor) with Pillow's float32 arithmetic (degenerate = black /
 * the constant int(mean(L) + 0.5) / the pixel's L); op 3 hue: H += uint8(factor * 255) (mod 256) through Pillow's RGB <-> HSV conversions,
 * factor in [-0.5, 0.5].  scratch8: 8 bytes of device memory (used by contrast). */
int abr_img_color_jitter_u8(uint8_t* img, int H, int W, int op, double factor, void* scratch8, void* stream);
/* one [3,HP,WP] fp32 slot of the batch tensor: (optional hflip) -> /255 -> [2,1,0]*255 (to_bgr255) -> (x-mean)/std, zeros outside
 * [h,w] (transforms.py:108-165 + to_image_list).  mean/std are HOST pointers to 3 floats. */
int abr_img_normalize_to_batch(const uint8_t* src, int h, int w, int flip, int to_bgr255, const float* mean3_host,
                               const float* std3_host, float* out_slot, int HP, int WP, void* stream);

/* =====================================================================================================
 * 8. Data-parallel gradient exchange: what DistributedDataParallel's reducer does for the reference
 *    between backward and optimizer.step() (tools/train_incremental.py:231-235; reduce in maskrcnn_benchmark/engine/trainer.py:15-37),
 *    as an in-place sum all-reduce of element ranges of the FLAT fp32 gradient buffer over RCCL (csrc/comm.hip).  One process per GPU.
 *    librccl.so.1 is loaded at first use (a copy the process already holds is reused).  Rendezvous: rank 0 calls abr_comm_unique_id and
 *    distributes the 128 bytes out of band; every rank then calls abr_comm_init (collective) on its own device.
 * ===================================================================================================== */
/* RCCL's version code (ncclGetVersion), 0 when librccl cannot be loaded */
int abr_comm_rccl_version(void);
/* id128_host: 128 bytes (ncclUniqueId) */
int abr_comm_unique_id(void* id128_host);
/* *comm_out = a communicator of `world` ranks on the CURRENT device (hipSetDevice first); collective over the ranks */
int abr_comm_init(int world, int rank, const void* id128_host, void** comm_out);
/* out[0] = ranks, out[1] = this rank, out[2] = device ordinal the communicator was created on */
int abr_comm_info(void* comm, int32_t* out_host);
int abr_comm_destroy(void* comm);
/* flat[a_i : b_i) = sum over ranks, in place, for the n_ranges ranges ranges_host = {a_0, b_0, a_1, b_1, ...} (element offsets, HOST array):
 * ONE RCCL group enqueued on `stream`; returns without synchronising.  Every rank must pass the same ranges in the same order. */
int abr_allreduce_flat(void* comm, float* flat, const int64_t* ranges_host, int n_ranges, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ABR_IOD_HIP_H */
