/*
 * mi3d.h — C ABI of libmi3d.so: the MI355X-native (gfx950, hand-written HIP) 3D U-Net training hot path.
 *
 * The reference (fransiskusbudi/multimodal_segmentation_project) has NO FFI/plugin boundary: its hot path is
 * reached through Python objects (models/unet.py UNet3D, models/unet_dann.py UNet3D, utils/metrics.py losses and
 * metrics, train_dann.py GradientReversal/DomainDiscriminator) whose arithmetic is delegated to PyTorch.  This
 * header is therefore the boundary a maintainer binds with ctypes underneath those objects (INTEGRATION.md shows
 * the stub); every entry point names the reference interface (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - the library NEVER allocates, frees or synchronises: every buffer (inputs, outputs, saved activations,
 *     scratch) is caller-owned device memory; sizes come from the *_workspace_bytes() queries.
 *   - return 0 on success; < 0 argument/shape/dtype error; > 0 hipError_t.  mi3d_last_error() (thread-local)
 *     describes the last failure.  No C++ exception crosses the boundary.
 *   - callable from any host thread (PyTorch runs backward on its autograd thread); launches go to the stream
 *     given; no thread-local device state is kept.
 *   - external tensor layout = PyTorch's: NCDHW contiguous, float32 data, int64 labels.  Internally activations
 *     are channels-last in `dtype` (MI3D_F32 exact path, MI3D_BF16 fast path with fp32 accumulation/statistics).
 */
#ifndef MI3D_H
#define MI3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI3D_DTYPE_F32 0
#define MI3D_DTYPE_BF16 1
#define MI3D_MAX_LEVELS 6

const char* mi3d_last_error(void);
int mi3d_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Whole-network plan: UNet3D forward / backward.
 * Replaces models/unet.py:64-90 (UNet3D.forward) and models/unet_dann.py:65-98 (forward with return_features),
 * i.e. DoubleConv blocks (unet.py:6-22: Conv3d k3 p1 -> BatchNorm3d -> ReLU -> Dropout3d, twice), MaxPool3d(2,2)
 * (:40,71), ConvTranspose3d(k2,s2) + cat(skip, x) (:56-58,79-84) and the final 1x1x1 conv (:62,87), plus their
 * autograd backward (train_unet.py:225 accelerator.backward).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct mi3d_unet_desc {
    int32_t in_channels, out_channels;      /* UNet3D(in_channels, out_channels, ...)  unet.py:34 */
    int32_t n_levels;                       /* len(features), <= MI3D_MAX_LEVELS */
    int32_t features[MI3D_MAX_LEVELS];      /* default [16,32,64,128] */
    int32_t N, D, H, W;                     /* per-GPU batch and volume; D,H,W divisible by 2^n_levels */
    int32_t dtype;                          /* internal activation dtype */
    float bn_momentum, bn_eps;              /* 0.1, 1e-5 (nn.BatchNorm3d defaults) */
    int32_t prepacked_from;                 /* 0 (default): a training forward packs every MFMA weight image itself.  k > 0: the images
                                             * of the DoubleConv blocks with index >= k (order: encoder.0..L-1, bottleneck, decoder.0..L-1)
                                             * and of all transposed convs in `workspace` are current -- the caller ran
                                             * mi3d_unet_pack_from(first_block = k) on the same workspace after its last parameter
                                             * update -- and the forward packs only blocks < k.  Read by mi3d_unet_forward* only. */
} mi3d_unet_desc;

/* Parameter / buffer pointer tables follow nn.Module.parameters() / .buffers() order of the reference model:
 *   params : for each DoubleConv in [encoder.0..L-1, bottleneck] : conv0.w, conv0.b, bn0.w, bn0.b, conv1.w, conv1.b,
 *            bn1.w, bn1.b ; then upconvs.0..L-1 : w, b ; then decoder.0..L-1 (8 each) ; then final_conv.w, .b
 *   buffers: for each DoubleConv in [encoder..., bottleneck, decoder...] : bn0.running_mean, bn0.running_var,
 *            bn0.num_batches_tracked(int64), bn1.(same three)
 * grads uses the params order; entries may be NULL to skip a gradient. */
int mi3d_unet_num_params(const mi3d_unet_desc* d);
int mi3d_unet_num_buffers(const mi3d_unet_desc* d);
size_t mi3d_unet_workspace_bytes(const mi3d_unet_desc* d);
/* number of Dropout3d scale entries (sum over the 2*(2L+1) dropout layers of N*C); layout = block order
 * [encoder..., bottleneck, decoder...], half 0 then half 1, each [N][C] */
int64_t mi3d_unet_dropout_count(const mi3d_unet_desc* d);

/* training != 0: batch statistics + running-stat update (BN train mode); == 0: running stats (eval mode).
 * training == 2: batch statistics with the running-stat update DEFERRED -- buffers[i] of every running_mean slot then points
 * at a device double[2*C] side buffer (the running_var / num_batches_tracked slots are ignored) into which the forward
 * publishes (batch mean, unbiased batch variance); mi3d_unet_bn_apply_deferred applies them to the real buffers later.  For
 * two forwards of ONE model on two streams (train_dann.py:268-272: source then target): the second one's updates of the
 * shared buffers are applied after both have run, in the reference's order, bit-identically to the serial execution.
 * training | 4 (with 1 or 2): another forward runs beside this one on a second stream -- the BatchNorm apply passes stay thin
 * (a finalize launch per layer at levels 0-1) instead of the wide whole-CU consumers, which slow each other down there.
 * drop_scales: device float[mi3d_unet_dropout_count] holding 0 or 1/(1-p), or NULL (p = 0 / eval).
 * logits: device float (N,out_channels,D,H,W), or NULL: the 1x1x1 head is not run (DANN target pass, whose logits nobody reads;
 * or the caller runs head + loss with mi3d_unet_head_loss_forward).  gap_out: device float (N, 2*features[L-1]) or NULL
 * (unet_dann.py:77-79).  The workspace keeps everything backward needs until the next forward. */
int mi3d_unet_forward(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers,
                      const float* drop_scales, int training, float* logits, float* gap_out, void* workspace,
                      size_t workspace_bytes, void* stream);

int mi3d_unet_bn_apply_deferred(const mi3d_unet_desc* d, void* const* buffers, const void* const* side, void* stream);

/* Inference forward (eval mode, no backward possible afterwards).  Replaces model.eval() + forward of
 * train_unet.py:259-305 (evaluate), test_model.py:242-251 and the frozen teacher of distill_unet.py:109-111,216-220:
 * eval-mode BatchNorm3d is folded into the preceding Conv3d (filter * gamma/sqrt(running_var+eps) at pack time, bias
 * replaced), ReLU runs in the conv epilogue, Dropout3d is the identity, and each conv writes the activated tensor
 * directly -- no raw conv outputs, statistics or saved activations.  Same workspace size as mi3d_unet_forward; buffers
 * (running statistics) are read-only here.  logits = NULL: as for mi3d_unet_forward. */
int mi3d_unet_infer(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers, float* logits,
                    float* gap_out, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the last mi3d_unet_forward on this workspace.  dlogits (N,out,D,H,W) float or NULL (target pass of
 * DANN: only the GAP branch carries gradient, train_dann.py:271-285); dgap (N,2*features[L-1]) float or NULL, its
 * contribution is scaled by gap_scale (gradient reversal folds in as gap_scale = -lambda, train_dann.py:29).
 * accumulate != 0: grads += ; else grads = .
 * Segments allow the caller to interleave gradient all-reduces: segment 0 = final_conv,
 * 1..L = decoder.L-1..0 (+ its upconv), L+1 = bottleneck, L+2..2L+1 = encoder.L-1..0.  Run [seg_begin, seg_end). */
int mi3d_unet_num_segments(const mi3d_unet_desc* d);
int mi3d_unet_backward(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* grads,
                       const float* drop_scales, const float* dlogits, const float* dgap, float gap_scale,
                       int accumulate, int seg_begin, int seg_end, void* workspace, size_t workspace_bytes,
                       void* stream, void* aux_stream, void* const* events, int aux_join);
/* aux_stream/events (both NULL = single stream): a second hipStream_t and 4 hipEvent_t handles (mi3d_event_create).
 * The backward's critical path is then the input-gradient chain alone (train_unet.py:225 fixes no order between the two
 * gradients of a layer): the dy of every decoder conv at the full-resolution levels and of every conv at the deep levels is
 * kept in its own buffer, and those weight gradients run on aux_stream -- the decoder's under the latency-bound deep-level
 * chain, the deep levels' under the bandwidth-bound encoder backward -- behind at most three forks (events[0..2], recorded
 * on `stream`) per call.  events[3] is recorded on aux_stream when it has finished the call's work; aux_join != 0 makes
 * `stream` wait for it before the call returns (the call is then stream-ordered for the caller), aux_join == 0 leaves that
 * to the caller: whatever consumes the gradients (optimizer, gradient exchange) and the end of a hipGraph capture must wait
 * for events[3] / aux_stream.  Results are bit-identical to the single-stream route. */
/* Optimizer tail beside the end of the backward (round 4).  With an aux stream the weight gradients of the leading encoder
 * blocks (the full-resolution levels) are the LAST thing the compute stream produces, everything else is complete on the aux
 * stream earlier: the caller can run the optimizer update of the other parameters and re-pack their MFMA weight images THERE,
 * and only the update of the leading blocks behind the join (train_unet.py:226 fixes no order between parameters).
 *   mi3d_unet_chain_tail_blocks: k = number of leading encoder blocks whose gradients come last (0: do not split).  Valid
 *     when mi3d_unet_backward*(seg 0 .. all, aux_stream, aux_join = 0) has returned: aux_stream is then ordered after every
 *     gradient of the blocks >= k, of the transposed convs and of the head (routes as they are NOW: ask before every step).
 *   mi3d_unet_pack_from: the weight packs of blocks >= first_block and of all transposed convs, one launch on `stream`
 *     (what the forward does for all blocks when desc.prepacked_from == 0). */
int mi3d_unet_chain_tail_blocks(const mi3d_unet_desc* d);
int mi3d_unet_pack_from(const mi3d_unet_desc* d, const void* const* params, void* workspace, size_t workspace_bytes,
                        int first_block, void* stream);
/* Exchange marks (data-parallel step, round 4): events[i] is recorded on the backward's stream as soon as every gradient of the
 * segments <= segs[i] is complete -- set for the NEXT mi3d_unet_backward / _backward_loss call of the calling thread (n <= 4,
 * n = 0 clears).  The backward then stays ONE call over all segments (a call cut at an exchange costs a slab-sum launch and a
 * host round trip); the exchange stream waits for the mark (mi3d_stream_wait_event) and all-reduces the bucket while the
 * remaining segments run.  Marks are consumed by that call. */
int mi3d_unet_backward_marks(const int* segs, void* const* events, int n);
/* Utilities: stream-to-stream ordering through a counter in device memory (flag = device int64[2], zero-initialised by the caller).
 * mi3d_flag_set: flag[0] = value once everything enqueued on `stream` so far has finished (one 1-thread kernel).
 * mi3d_flag_wait: `stream` continues when flag[0] >= value (one 1-wave kernel that polls; values must grow monotonically).  After
 * timeout_us without it the waiter gives up and stores `value` in flag[1].  Round 4 measured them as a replacement for the hardware
 * cross-queue join of the data-parallel step: no gain; the step does not use them, tools/queue_probe.py does. */
int mi3d_flag_set(int64_t* flag, int64_t value, void* stream);
int mi3d_flag_wait(int64_t* flag, int64_t value, int64_t timeout_us, void* stream);
int mi3d_stream_wait_event(void* stream, void* event);
int mi3d_event_create(void** event_out);
int mi3d_event_destroy(void* event);
/* A non-blocking hipStream_t of a priority class: -1 = the device's highest, 0 = middle, +1 = lowest.  torch.cuda.Stream only
 * offers high / normal; the aux stream of the deferred weight gradients wants the LOWEST class, so that the wave dispatcher
 * gives a free workgroup slot to the data-gradient chain first.  The caller owns the stream (mi3d_stream_destroy). */
int mi3d_stream_create(int priority_class, void** stream_out);
/* A non-blocking hipStream_t whose kernels may only run on `cus_per_xcd` (1..31) of the 32 compute units of every XCD
 * (hipExtStreamCreateWithCUMask; mask bit i = CU i / 8 of XCD i % 8, measured with tools/micro/cu_mask_probe.hip): from_top = 0
 * takes CUs 0..n-1 of each XCD, 1 the last n.  Round 4: a CU partition for the aux stream of the deferred weight gradients, so
 * that they stop taking workgroup slots from the data-gradient chain on every CU. */
int mi3d_stream_create_masked(int cus_per_xcd, int from_top, void** stream_out);
int mi3d_stream_destroy(void* stream);
/* Route switches: every kernel-selection switch of the library ("no_persist", "no_fused_bwd", "ks_target", ... -- the table
 * in INTEGRATION.md) is read from the environment (MI3D_<NAME>=<int>) ONCE, when the library is first used; afterwards only
 * these calls change one.  Debug / test interface: do not call it concurrently with launches; a captured hipGraph keeps the
 * routes it was captured with.  Unknown names fail. */
int mi3d_debug_set_route(const char* name, int value);
int mi3d_debug_get_route(const char* name, int* value_out);
int mi3d_debug_route_count(void);
int mi3d_debug_experiments(void);      /* 1: built with make EXPERIMENTS=1 (default-off experiment kernels compiled in) */
const char* mi3d_debug_route_name(int index);
/* Measurement hook (bench.py `roofline`: HIP events around ONE kernel on the stream it is launched on, also the aux stream).
 * mi3d_time_next_conv3_kernel arms it for the calling thread: the next launch of `kind` for the layer (Cin, Cout as that
 * launcher sees them) records start/stop -- timing events from mi3d_timing_event_create -- tightly around its kernel.
 *   kind 0 = fused full-resolution conv backward (input + weight gradient in one launch; Cin, Cout of the layer)
 *        1 = stand-alone conv weight gradient (Cin, Cout of the layer)
 *        2 = persistent full-resolution conv with BatchNorm partial sums (the training forward; Cin, Cout of the conv)
 *        3 = persistent full-resolution conv without (the input gradient: Cin = the layer's Cout, Cout = the layer's Cin)
 * One-shot.  mi3d_time_hook_fired() returns 1 when the armed launch happened and disarms the hook either way (call it before
 * reading the events: events that were never recorded cannot be waited for); NULL events disarm.
 * mi3d_event_elapsed_ms synchronises on `stop`. */
/* mi3d_debug_occupy_cus: a stand-in for a resident collective kernel (RCCL all-reduce) on a 1-GPU box: `workgroups` x 512
 * threads x 128 VGPRs hold their CU slots for `microseconds` on `stream` and read through buf[0..n) meanwhile
 * (bench.py --emulate-comm: what a collective beside the encoder backward costs the persistent grids). */
int mi3d_debug_occupy_cus(int workgroups, int microseconds, float* buf, int64_t n, void* stream);
/* mi3d_set_cu_budget: CUs (0..128) the persistent conv grids launched from the calling thread leave free for a collective
 * kernel that is resident beside them (TrainStep sets it for the backward segments that overlap a gradient exchange).
 * THREAD-LOCAL: it applies to launches made by the thread that set it (a backward run by another thread, e.g. the autograd engine's,
 * does not see it), and a hipGraph capture freezes the value that was active while it was captured.  The grids are sized from the
 * device's compute-unit count (hipDeviceAttributeMultiprocessorCount, read once per process). */
int mi3d_set_cu_budget(int cus);
int mi3d_timing_event_create(void** event_out);
int mi3d_time_next_conv3_kernel(void* start_event, void* stop_event, int kind, int Cin, int Cout);
int mi3d_time_hook_fired(void);
int mi3d_event_elapsed_ms(void* start_event, void* stop_event, float* ms_out);
/* params-table index ranges whose gradients segment `seg` produces: ranges = {first0, last0, first1, last1}
 * (half-open; the second range is the segment's upconv for decoder segments, otherwise {-1,-1}) */
int mi3d_unet_segment_params(const mi3d_unet_desc* d, int seg, int* ranges);

/* ------------------------------------------------------------------------------------------------------------
 * Losses and metrics.  Replace utils/metrics.py:14-40 (combined_loss), :137-156 (tversky_loss), :158-167
 * (combined_ce_tversky_loss), :169-190 (distillation_loss), train_unet.py:186-198 ('dice' variant) and
 * utils/metrics.py:65-129 (calculate_iou / calculate_dice / calculate_accuracy, incl. its class-loop bound).
 *   loss = w_ce*CE_mean + w_reg*mean_{c>=1} region_c + w_kd*T^2*mean_{n,c,v} KL(teacher||student)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct mi3d_loss_cfg {
    float w_ce;
    int32_t region_kind;        /* 0 none, 1 soft Dice (eps 1e-5), 2 Tversky(alpha,beta) (eps 1e-6) */
    float w_reg, alpha, beta, eps;
    float w_kd, temperature;    /* distillation: w_ce,w_reg pre-multiplied by alpha_kd; w_kd = 1-alpha_kd */
} mi3d_loss_cfg;
#define MI3D_LOSS_COEF_FLOATS 20
size_t mi3d_seg_loss_workspace_bytes(int C);
/* logits/teacher (N,C,V) float, labels (N,V) int64; loss_out: device float[1]; coef: device float[20] kept for bwd */
int mi3d_seg_loss_forward(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V,
                          const mi3d_loss_cfg* cfg, float* loss_out, float* coef, void* workspace, void* stream);
/* grad_out: device float[1] upstream gradient (NULL = 1) */
int mi3d_seg_loss_backward(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V,
                           const mi3d_loss_cfg* cfg, const float* coef, const float* grad_out, float* dlogits,
                           void* stream);
/* The training step needs the loss AND the metrics of the same logits (train_unet.py:224-232): one pass over
 * logits + labels for both.  metrics_out as mi3d_seg_metrics (device float[3]), workspaces as the two calls. */
int mi3d_seg_loss_metrics_forward(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int D,
                                  int64_t V, const mi3d_loss_cfg* cfg, float* loss_out, float* coef, float* metrics_out,
                                  void* loss_workspace, void* metrics_workspace, void* stream);
/* The training step (train_unet.py:224-232, train_dann.py:268-281, distill_unet.py:107-115) uses the logits only inside the loss,
 * so the 1x1x1 head (models/unet.py:62,87) and the loss can be one pass each way and neither logits nor dlogits ever reach memory:
 *   mi3d_unet_forward_loss      = mi3d_unet_forward + mi3d_seg_loss_metrics_forward   (logits_opt: NULL, or where to keep them)
 *   mi3d_unet_head_loss_forward = the head + loss part alone, on the decoder output that mi3d_unet_forward / mi3d_unet_infer called
 *                                 with logits = NULL left in the workspace (distillation: the teacher's forward runs on another
 *                                 stream and is joined in between; evaluation: infer, then this)
 *   mi3d_unet_backward_loss     = mi3d_seg_loss_backward + mi3d_unet_backward         (grad_scale: device float[1] or NULL = 1)
 * teacher_logits: (N,C,V) float, required iff cfg->w_kd != 0 (distillation_loss, utils/metrics.py:169-190), else NULL.
 * Every logit has the bits of the unfused head; the forward sums add the voxels in another order (loss equal to a few ulp,
 * the integer counts behind the metrics exactly), the backward is bit-identical to the unfused pair given the same `coef`.
 * mi3d_unet_head_loss_supported: 1 if the configuration has the fused kernels (bf16 activations, features[0] == 16,
 * <= 4 classes), else 0 and the calls fail with MI3D_EINVAL. */
int mi3d_unet_head_loss_supported(const mi3d_unet_desc* d, const mi3d_loss_cfg* cfg);
/* The two fused passes as operators (z: the decoder output, channels-last bf16 (N,V,zcs); w (C,Cin), bias (C) float;
 * workspace of mi3d_head_loss_backward: mi3d_conv1_workspace_bytes(Cin, C)). */
int mi3d_head_loss_supported(int dtype, int Cin, int C, const mi3d_loss_cfg* cfg);
int mi3d_head_loss_forward(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels,
                           const float* teacher, int N, int C, int D, int64_t V, const mi3d_loss_cfg* cfg, float* loss_out, float* coef,
                           float* metrics_out, void* loss_workspace, void* metrics_workspace, float* logits_opt, void* stream);
int mi3d_head_loss_backward(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels,
                            const float* teacher, int N, int C, int64_t V, const mi3d_loss_cfg* cfg, const float* coef,
                            const float* grad_scale, void* dz, int dzcs, float* dW, float* db, int accumulate, void* workspace,
                            size_t workspace_bytes, void* stream);
int mi3d_unet_forward_loss(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* buffers,
                           const float* drop_scales, int training, const int64_t* labels, const float* teacher_logits,
                           const mi3d_loss_cfg* cfg, float* loss_out, float* coef, float* metrics_out, void* loss_workspace,
                           void* metrics_workspace, float* logits_opt, float* gap_out, void* workspace, size_t workspace_bytes,
                           void* stream);
int mi3d_unet_head_loss_forward(const mi3d_unet_desc* d, const void* const* params, const int64_t* labels, const float* teacher_logits,
                                const mi3d_loss_cfg* cfg, float* loss_out, float* coef, float* metrics_out, void* loss_workspace,
                                void* metrics_workspace, float* logits_opt, void* workspace, size_t workspace_bytes, void* stream);
int mi3d_unet_backward_loss(const mi3d_unet_desc* d, const float* x, const void* const* params, void* const* grads,
                            const float* drop_scales, const int64_t* labels, const float* teacher_logits, const mi3d_loss_cfg* cfg,
                            const float* coef, const float* grad_scale, const float* dgap, float gap_scale, int accumulate, int seg_begin,
                            int seg_end, void* workspace, size_t workspace_bytes, void* stream, void* aux_stream, void* const* events,
                            int aux_join);
size_t mi3d_seg_metrics_workspace_bytes(int C);
/* out: device float[3] = {iou, dice, accuracy}; D = first spatial dim (reference loop bound, metrics.py:74,101) */
int mi3d_seg_metrics(const float* logits, const int64_t* labels, int N, int C, int D, int64_t V, float* out,
                     void* workspace, void* stream);
/* Evaluation.  Replaces the per-class mask/sum loops of test_model.py:255-276: exact counts of
 * pred = argmax(logits) against labels in one pass.  counts: device int64[3*C + 1] =
 * { n_inter[C] (pred==c && label==c), n_pred[C], n_label[C], n_correct };  same workspace as mi3d_seg_metrics. */
int mi3d_seg_class_counts(const float* logits, const int64_t* labels, int N, int C, int64_t V, int64_t* counts,
                          void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * DANN head.  Replaces train_dann.py:34-49 (DomainDiscriminator MLP 256-256-128-64-2, ReLU, Dropout 0.2) and
 * :283 (nn.CrossEntropyLoss on the concatenated domain predictions).  GradientReversal (:22-32) is the
 * gap_scale argument of mi3d_unet_backward plus gx_scale here.
 * ---------------------------------------------------------------------------------------------------------- */
int mi3d_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
                        const float* drop, void* stream);
/* y = the forward's output (ReLU mask); gx = gx_scale * dL/dx (gx_scale = -lambda folds the gradient reversal);
 * workspace: M*Nout floats */
int mi3d_linear_backward(const float* x, const float* w, const float* y, const float* gy, int M, int K, int Nout,
                         int relu, const float* drop, float* gx, float* gw, float* gb, int accumulate,
                         float gx_scale, float* workspace, void* stream);
/* y = alpha * (alpha_dev ? *alpha_dev : 1) * x over n floats.  GradientReversal.backward (train_dann.py:27-29:
 * grad_output.neg() * lambda_) is alpha = -lambda; the row-CE backward multiplies by the upstream gradient alpha_dev. */
int mi3d_scale(const float* x, float* y, int64_t n, float alpha, const float* alpha_dev, void* stream);
/* mean CE over M rows; dlogits = scale * d(loss)/d(logits) (may be NULL) */
int mi3d_softmax_ce_rows(const float* logits, const int64_t* labels, int M, int C, float* loss, float* dlogits,
                         float scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer and RNG helpers on the step path.
 * mi3d_adamw_step replaces torch.optim.AdamW(...).step() (train_unet.py:378,226) over one flat fp32 arena.
 * step_dev: device int64 holding the number of steps taken so far (incremented by the call).
 * ---------------------------------------------------------------------------------------------------------- */
int mi3d_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float grad_scale, int64_t* step_dev, void* stream);
/* Same update over ONE contiguous range of the arena without (increment = 0) or with the step increment: a step over
 * several trainable ranges (frozen encoder, train_unet.py:31-43,413-431) = one call per range, increment on the last;
 * n = 0 with increment = 1 only advances the counter. */
int mi3d_adamw_apply(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float grad_scale, int64_t* step_dev, int increment, void* stream);
/* Dropout3d (unet.py:14,18) channel masks: out[i] = 0 w.p. p else 1/(1-p); state_dev = device uint64[2]
 * {seed, counter}, counter advanced by n.  Counter-based RNG: same distribution as torch, different stream. */
int mi3d_dropout_scales(float* out, int64_t n, float p, uint64_t* state_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input pipeline: the per-volume arithmetic of CombinedDataset.__getitem__ (utils/dataloader.py:148-200)
 * on the device.  in/out: device float (n voxels of ONE volume; in == out allowed); labels int64.
 * ---------------------------------------------------------------------------------------------------------- */
/* preprocess_ct, utils/dataloader.py:111-117: clip to [window_min, window_max] (reference: -160, 240), scale to [0,1] */
int mi3d_preprocess_ct(const float* in, float* out, int64_t n, float window_min, float window_max, void* stream);
/* preprocess_mri, utils/dataloader.py:128-144: z-score with the population std (+1e-8), clip to the
 * np.percentile([p_low, p_high]) values (reference: 1, 99; linear interpolation between exact order statistics),
 * (x - low) / (high - low + 1e-8).  workspace: mi3d_preprocess_mri_workspace_bytes(), 16-byte aligned. */
size_t mi3d_preprocess_mri_workspace_bytes(void);
int mi3d_preprocess_mri(const float* in, float* out, int64_t n, float p_low, float p_high, void* workspace, void* stream);
/* label remaps, utils/dataloader.py:162-181.  kind 0: identity (ts / btcv), 1: AMOS {0:0,1:1,2:3,3:3,6:2, else 0},
 * 2: CHAOS ranges {[55,70]:2, [110,135]:3, [175,200]:3, [240,255]:1, else 0} */
int mi3d_remap_labels(const int64_t* in, int64_t* out, int64_t n, int kind, void* stream);

/* Training-set augmentation, combined_transform() (utils/dataloader.py:223-262, used at train_unet.py:361): the
 * arithmetic of MONAI's RandBiasField -> RandGaussianNoise -> RandAdjustContrast -> RandHistogramShift ->
 * RandCoarseDropout on one (C, D, H, W) float volume, with the random parameters already drawn by the host
 * (augment.py draws them from numpy RandomState streams laid out like MONAI's Compose).  A transform whose do_* flag
 * (n_holes for the last) is 0 is skipped, as when MONAI's prob draw fails.
 *   bias      x * exp(field), field = Legendre series (degree <= 3; coefficient order i, j, k with i + j + k <= degree
 *             over the three spatial axes) on linspace(-1, 1, dim) float32 coordinates, evaluated in float64
 *   noise     x + noise[i] (noise != NULL: host-drawn tensor, same shape) or x + N(noise_mean, noise_std) from the
 *             library's counter-based generator seeded with noise_seed (same distribution, different stream)
 *   contrast  ((x - min) / (max - min + 1e-7)) ** gamma * (max - min) + min, min / max over the whole volume
 *   hist      piecewise-linear map through n_cp control points ref_cp -> flt_cp (both on [0, 1], scaled to [min, max])
 *   holes     n_holes boxes [hole_lo, hole_lo + hole_size) over all channels set to fill_value */
#define MI3D_AUG_MAX_COEFF 20
#define MI3D_AUG_MAX_CP 16
#define MI3D_AUG_MAX_HOLES 8
typedef struct mi3d_aug_params {
    int32_t do_bias, bias_degree;
    double bias_coeff[MI3D_AUG_MAX_COEFF];
    int32_t do_noise;
    float noise_mean, noise_std;
    uint64_t noise_seed;
    int32_t do_contrast;
    float gamma;
    int32_t do_hist, n_cp;
    float ref_cp[MI3D_AUG_MAX_CP], flt_cp[MI3D_AUG_MAX_CP];
    int32_t n_holes;
    int32_t hole_size[3];
    int32_t hole_lo[MI3D_AUG_MAX_HOLES][3];
    float fill_value;
} mi3d_aug_params;
size_t mi3d_augment_workspace_bytes(void);
/* in == out allowed.  workspace: mi3d_augment_workspace_bytes(), 16-byte aligned. */
int mi3d_augment(const float* in, float* out, const float* noise, int C, int D, int H, int W, const mi3d_aug_params* params,
                 void* workspace, size_t workspace_bytes, void* stream);
/* the label half of RandCoarseDropoutd: the same boxes set to `fill` in an int64 (C, D, H, W) label volume, in place */
int mi3d_fill_boxes_i64(int64_t* label, int C, int D, int H, int W, int n_holes, const int32_t* hole_lo,
                        const int32_t* hole_size, int64_t fill, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Per-operator entry points (channels-last activations; used by the parity tests and by stand-alone modules).
 * x/y: `dtype` tensors [N][D][H][W][C] with channel stride xcs/ycs (elements).
 * ---------------------------------------------------------------------------------------------------------- */
/* nn.Conv3d(k=3,p=1) unet.py:11,15.  w (Cout,Cin,3,3,3) float.  workspace: mi3d_conv3_workspace_bytes */
size_t mi3d_conv3_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W);
int mi3d_conv3_forward(int in_dtype, int out_dtype, const void* x, int xcs, int Cin, const float* w, const float* bias,
                       void* y, int ycs, int Cout, int N, int D, int H, int W, void* workspace, size_t workspace_bytes,
                       void* stream);
/* dx may be NULL (first layer); dW (Cout,Cin,3,3,3), db (Cout) float */
int mi3d_conv3_backward(int x_dtype, int dy_dtype, const void* x, int xcs, int Cin, const float* w, const void* dy,
                        int dycs, int Cout, void* dx, int dxcs, float* dW, float* db, int accumulate, int N, int D,
                        int H, int W, void* workspace, size_t workspace_bytes, void* stream);
/* BatchNorm3d(train) + ReLU + Dropout3d fused; stat: device float[4*C] saved for backward */
size_t mi3d_bn_workspace_bytes(int C);
int mi3d_bn_relu_drop_forward(int dtype, const void* y, int ycs, int C, int64_t M, int64_t V, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float momentum, float eps, int training, const float* drop, void* z, int zcs, float* stat,
                              void* workspace, void* stream);
int mi3d_bn_relu_drop_backward(int dtype, const void* dz, int dzcs, const void* y, int ycs, int C, int64_t M, int64_t V,
                               const float* stat, const float* drop, void* dy, int dycs, float* dgamma, float* dbeta,
                               int accumulate, void* workspace, void* stream);
/* the second half of an encoder block as the training step runs it (unet.py:16-18 then :71): train-mode BatchNorm3d + ReLU +
 * Dropout3d writing z AND pooled = MaxPool3d(2,2)(z) in one pass.  Even D, H, W; pooled [N][D/2][H/2][W/2][C], stride pcs */
int mi3d_bn_relu_drop_pool_forward(int dtype, const void* y, int ycs, int C, int N, int D, int H, int W, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float momentum, float eps, const float* drop, void* z, int zcs, void* pooled, int pcs,
                                   float* stat, void* workspace, void* stream);
int mi3d_maxpool2_forward(int dtype, const void* z, int zcs, int C, int N, int D, int H, int W, void* p, int pcs,
                          void* stream);
int mi3d_maxpool2_backward(int dtype, const void* dp, int dpcs, const void* z, int zcs, const void* dskip, int dskipcs,
                           void* dz, int dzcs, int C, int N, int D, int H, int W, void* stream);
/* ConvTranspose3d(k2,s2) unet.py:56-58.  w (Cin,Cout,2,2,2) float; geometry = INPUT volume */
size_t mi3d_upconv2_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W);
int mi3d_upconv2_forward(int dtype, const void* x, int xcs, int Cin, const float* w, const float* bias, void* y, int ycs,
                         int Cout, int N, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
int mi3d_upconv2_backward(int dtype, const void* x, int xcs, int Cin, const float* w, const void* gy, int gycs, int Cout,
                          void* dx, int dxcs, float* dW, float* db, int accumulate, int N, int D, int H, int W,
                          void* workspace, size_t workspace_bytes, void* stream);
/* final nn.Conv3d(C0, n_classes, 1) unet.py:62,87.  z: channels-last `dtype` [N][V][Cin] (stride zcs); w (Cout,Cin,1,1,1) float;
 * logits / dlogits: NCDHW float (the API layout).  backward: dz (channels-last, stride dzcs), dW (Cout,Cin), db (Cout) float;
 * workspace: mi3d_conv1_workspace_bytes.  The bf16 path runs the MFMA kernels of the training step. */
size_t mi3d_conv1_workspace_bytes(int Cin, int Cout);
int mi3d_conv1_forward(int dtype, const void* z, int zcs, int Cin, const float* w, const float* bias, float* logits, int Cout,
                       int N, int64_t V, void* stream);
int mi3d_conv1_backward(int dtype, const void* z, int zcs, int Cin, const float* w, const float* dlogits, int Cout, void* dz,
                        int dzcs, float* dW, float* db, int accumulate, int N, int64_t V, void* workspace,
                        size_t workspace_bytes, void* stream);
/* layout helpers: NCDHW float <-> channels-last `dtype` */
int mi3d_ncdhw_to_ndhwc(int dtype, const float* src, void* dst, int dcs, int C, int N, int64_t V, void* stream);
int mi3d_ndhwc_to_ncdhw(int dtype, const void* src, int scs, float* dst, int C, int N, int64_t V, void* stream);

/* hipGraph capture helpers (launch-bound step loops): capture everything launched on `stream` between begin/end */
int mi3d_graph_begin(void* stream);
int mi3d_graph_end(void* stream, void** graph_exec_out);
int mi3d_graph_launch(void* graph_exec, void* stream);
int mi3d_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* MI3D_H */
