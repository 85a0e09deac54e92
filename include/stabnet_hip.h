/* libstabnet_hip.so -- C ABI of the MI355X (gfx950) StabNet hot path.
 *
 * The reference (cxjyxxme/deep-online-video-stabilization, TensorFlow 1.3) has no FFI: its boundary is the
 * TF1 session -- Python op signatures at graph-build time plus a named-tensor contract at run time
 * (SURVEY.md section 8b).  Each entry point below replaces the TF-op cluster behind one of those Python
 * signatures / fetched tensors; the citation names it (file:line in the reference tree).
 *
 * Conventions (all entry points):
 *   - plain pointers to CALLER-OWNED DEVICE memory, NHWC contiguous float32 unless stated; explicit shapes;
 *     `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - never allocates, frees or synchronises; work that needs scratch takes a caller-provided workspace whose
 *     size is queried up front; every call only enqueues kernels on `stream` (hipGraph-capturable);
 *   - returns 0 on success, negative on error (-1 bad argument, -2 launch failure, -3 workspace too small);
 *     the message is available from stabnet_last_error() (thread-local);
 *   - thread-safe per stream.
 */
#ifndef STABNET_HIP_H
#define STABNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* stabnet_last_error(void);
int stabnet_abi_version(void);

/* ---- mesh + multi-grid warp, forward ------------------------------------------------------------------- */

/* get_4_pts(theta, batch_size) -> pts2            s_net_bundle_nobm.py:29-71
 * + get_Hs(pts2) -> Hs                           spatial_transformer3.py:179-198 (get_H/pinv :144-175)
 * theta [N,(gh+1)(gw+1)*2] -> pts2 [N,gh+1,gw+1,2] (vertex = regular grid + offset, clipped to +-1/do_crop_rate),
 * Hs [N,gh,gw,9] (h = inv(A + 1e-4 I) b, last entry 1), pts1 [N,gh,gw,8] (optional, may be NULL): per cell
 * [x_TL,x_TR,x_BL,x_BR,y_TL,y_TR,y_BL,y_BR] (s_net_bundle_nobm.py:65-66). */
int stabnet_get_4_pts(const float* theta, int N, int grid_h, int grid_w, float do_crop_rate,
                      float* pts1, float* pts2, float* Hs, void* stream);

/* transformer(U, theta=pts2) -> (output, black_pix, img=[x_map,y_map])   spatial_transformer3.py:19,218-301,362-365
 * Fetched in deploy as output_img:0, black_pix:0, get_Hs/Hs:0, x_map:0, y_map:0 (deploy_bundle.py:48-56,286).
 * U [N,H,W,C]; pts2 [N,gh+1,gw+1,2]; out [N,H,W,C]; black,x_map,y_map [N,H,W]; Hs [N,gh,gw,9]. */
int stabnet_transformer_fwd(const float* pts2, const float* U, int N, int H, int W, int C, int grid_h, int grid_w,
                            float* out, float* black, float* x_map, float* y_map, float* Hs, void* stream);

/* Fused get_4_pts + transformer: what one deploy frame / training tower runs after the regressor
 * (s_net_bundle_nobm.py:304-307,332).  pts2 may be NULL. */
int stabnet_warp_fwd(const float* theta, const float* U, int N, int H, int W, int C, int grid_h, int grid_w,
                     float do_crop_rate, float* out, float* black, float* x_map, float* y_map, float* Hs,
                     float* pts2, void* stream);

/* _transform3 map stage + _interpolate from given homographies (spatial_transformer3.py:227-295). */
int stabnet_maps_from_hs_fwd(const float* Hs, const float* U, int N, int H, int W, int C, int grid_h, int grid_w,
                             float* out, float* black, float* x_map, float* y_map, void* stream);

/* interpolate(im, x, y, out_size) -> output      spatial_transformer.py:200-281 (used train_bundle_nobm.py:117-118)
 * im [N,H,W,C]; x,y [N,H,W] normalised coordinates; out [N,H,W,C]. */
int stabnet_interp_fwd(const float* im, const float* x, const float* y, int N, int H, int W, int C, float* out,
                       void* stream);

/* ---- regressor building blocks -------------------------------------------------------------------------- */

/* slim conv2d / conv2d_same (+ folded batch_norm + relu on the INPUT, + bias / residual / relu on the output):
 * the op cluster of resnet_v2_50's bottleneck units called at s_net_bundle_nobm.py:252-253.
 * x NHWC [N,H,W,Cin] (Cin % 16 == 0); w OHWI [Cout][KH][KW][Cin]; symmetric zero pad `pad`; y NHWC [N,Ho,Wo,Cout],
 * Ho = (H + 2 pad - KH)/stride + 1.   A-operand prologue (both or neither): a = relu(a*in_scale[c] + in_shift[c]),
 * applied to in-frame pixels only (padding stays 0).  Epilogue: + bias[n] (or NULL), + residual (or NULL; NHWC
 * [N,res_H,res_W,Cout] read at (oy*res_stride, ox*res_stride) -- slim's `subsample` identity shortcut), relu if
 * relu_out.  Arithmetic: exact float32 MFMA, fp32 accumulate.  workspace: stabnet_conv2d_workspace_bytes(). */
size_t stabnet_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
/* Tuning hook (tools/autotune.py): force tile (0 128x128, 1 128x64, 2 64x64; < 0 = built-in choice) and split-K. */
void stabnet_conv_tuning_override(int tile, int splitk);
/* Tuning hook (tools/tune_splitk.py): measured split-K of one convolution shape; ring = 1 for prologue-free launches.
 * splitk <= 0 removes the entry, M < 0 clears the table.  Applies to plans made afterwards. */
void stabnet_conv_tuning_table_set(int M, int Cout, int K, int KH, int ring, int splitk);
/* Which measured split-K table plans made AFTERWARDS use: 0 = the exact-f32-MFMA kernels (default), 1 = the packed split kernels
 * (operand mode 4 of stabnet_net_set_bf16_operands: two workgroups per CU, two-way K split inside the workgroup).  Set it around
 * stabnet_net_create() of a plan that will run in mode 4 (stabnet_amd.regressor does). */
void stabnet_conv_tuning_profile(int profile);
int stabnet_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, const float* in_scale,
                       const float* in_shift, const float* residual, int res_H, int res_W, int res_stride, float* y,
                       int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu_out,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Same, with the NEXT layer's folded batch_norm fused behind the sum: y = act((conv + bias + residual)*out_scale[n] +
 * out_shift[n]) (both or neither; NULL = plain epilogue).  This is how the inference plan runs the
 * conv1 -> bn -> relu -> conv2 -> bn -> relu -> conv3 chain of a bottleneck unit (resnet_v2 `bottleneck`, slim). */
int stabnet_conv2d_fwd_ex(const float* x, const float* w_ohwi, const float* bias, const float* in_scale,
                          const float* in_shift, const float* residual, int res_H, int res_W, int res_stride,
                          const float* out_scale, const float* out_shift, float* y, int N, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int relu_out, void* workspace,
                          size_t workspace_bytes, void* stream);
/* stabnet_conv2d_fwd_ex on the bf16 matrix pipe with float32-level results ("packed split", operand mode 4 of
 * stabnet_net_set_bf16_operands; replaces the same slim conv2d calls, s_net_bundle_nobm.py:252-253): every float32 operand is the
 * exact sum of three bf16 terms, the product is accumulated in float32 from six bf16 x bf16 partial products.  The weights are
 * split once: stabnet_conv_weight_split_image() writes the fragment-major image (stabnet_conv_weight_image_floats() floats, Cin a
 * multiple of 32) the kernel reads; w_ohwi must still be passed (geometries the packed kernel does not take -- Cin % 32 != 0,
 * other tiles -- run stabnet_conv2d_fwd_ex's exact-f32 kernels on it).  splitk > 0 forces the K split (2 with equal halves runs
 * inside the workgroup, others through `workspace` slabs + a reduce launch); 0 = the planned one.  workspace: at least
 * max(stabnet_conv2d_workspace_bytes(), splitk * N*Ho*Wo*Cout * 4). */
size_t stabnet_conv_weight_image_floats(int Cout, int KH, int KW, int Cin);
int stabnet_conv_weight_split_image(const float* w_ohwi, int Cout, int KH, int KW, int Cin, float* w_img, void* stream);
int stabnet_conv2d_fwd_packed(const float* x, const float* w_ohwi, const float* w_img, const float* bias, const float* in_scale,
                              const float* in_shift, const float* residual, int res_H, int res_W, int res_stride,
                              const float* out_scale, const float* out_shift, float* y, int N, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad, int relu_out, int splitk, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The tail of a slim bottleneck_v2 unit as ONE launch (resnet_v2 `bottleneck`, called at s_net_bundle_nobm.py:252-253; what the
 * inference plan runs for the block-1 / block-2 units of a frame): conv2 (3x3, pad 1, stride 1 | 2, C -> C channels, C = 64 | 128,
 * no bias) -> folded batch_norm (mid_scale, mid_shift) + ReLU -> conv3 (1x1, C -> Cout, Cout % C == 0) with conv2d_fwd_ex's
 * epilogue: y = act((conv3 + bias3 + residual) * out_scale + out_shift).  The C-channel intermediate never reaches memory (a
 * workgroup keeps its activated 64-pixel tile in LDS and feeds the 1x1 GEMM from there).  x [N,H,W,C] with x_ld floats between
 * pixels (0 = C); residual read at (oy*res_stride, ox*res_stride) with res_ld floats between pixels (0 = Cout); y [N,Ho,Wo,Cout].
 * No workspace.  STABNET_ERR_BAD_ARG for other geometries (use two stabnet_conv2d_fwd_ex calls). */
int stabnet_conv3x3_conv1x1_fwd(const float* x, int x_ld, const float* w2_ohwi, const float* mid_scale, const float* mid_shift,
                                const float* w3_ohwi, const float* bias3, const float* residual, int res_H, int res_W,
                                int res_stride, int res_ld, const float* out_scale, const float* out_shift, float* y, int N, int H,
                                int W, int C, int Cout, int stride, int relu_out, void* stream);

/* ---- the regressor as one plan --------------------------------------------------------------------------
 * get_resnet(x_tensor, reuse, is_training=False, x_batch_size) -> theta       s_net_bundle_nobm.py:250-264
 *   = slim resnet_v2_50(global_pool=False, output_stride=32) -> reduce_mean([1,2]) -> fully_connected 2048/1024/512
 *     -> output_layer (resnet.py:44-56).  The run-time feed/fetch it serves: x_tensor:0 -> (theta ->) the tensors of
 *     deploy_bundle.py:48-56,286.
 * A net handle is a HOST-ONLY description (layer list, buffer offsets); it owns no device memory.
 * Parameter buffer (floats): [weights/biases in network order][BN gammas][BN betas] | [moving means][moving vars];
 * the first stabnet_net_trainable_floats() floats are the trainables.  Conv weights are OHWI with Cin padded to 16,
 * FC weights [out][in]; stabnet_net_param_info() gives name (TF variable name under stable_net/resnet/), offset,
 * kind (0 conv w, 1 conv bias, 2 gamma, 3 beta, 4 moving_mean, 5 moving_variance, 6 FC w, 7 FC bias), dims, aux
 * (conv: un-padded Cin). */
int stabnet_net_create(void** net, int N, int H, int W, int in_ch, int n_theta, int keep_activations);
void stabnet_net_destroy(void* net);
/* Conv operand mode of the inference forward (tensors are float32 in memory and accumulation is float32 in every mode):
 *   0  exact f32 MFMA (v_mfma_f32_32x32x2_f32), the default;
 *   1  SECONDARY reduced-precision mode (never the headline): operands rounded to bf16 at fragment-read
 *      time.  Own, looser parity bar (3e-3 on theta);
 *   2 / 3  split operands: every f32 operand is decomposed EXACTLY into three bf16 terms (x = h + m + l) when its fragment is read
 *      and the f32 product is accumulated as six (3: all nine) bf16 x bf16 partial products on v_mfma_f32_32x32x16_bf16 -- f32-level
 *      results (theta within 2e-7 of the oracle, like mode 0) on the 16x faster matrix pipe; VALU-bound, slower than mode 0: kept
 *      as the reference form of mode 4;
 *   4  packed split: the weights are split once into a fragment-major image inside `fold` (stabnet_net_fold_bn), only the A
 *      fragments are split at run time (conv_ring_f32_kernel<MODE, 4, KG, PRO>).  Same f32-level parity bar as mode 0.
 * Training plans (keep_activations = 1) accept 0 and 4 only; 4 = the step's weight-operand launches (the prologue-carrying 1x1
 * forward pairs, the stride-1 dgrad launches) read images of the forward weights and of the re-packed dgrad weights that the step
 * writes itself, inside its workspace -- call it BEFORE stabnet_net_train_workspace_bytes().  Off by default in the Python
 * mirror (no sustained gain at 8 pairs per GPU on a power-limited part; it pays at larger batches). */
int stabnet_net_set_bf16_operands(void* net, int on);
int stabnet_net_num_params(const void* net);
int stabnet_net_param_info(const void* net, int idx, char* name, int name_cap, long* offset, int* kind, int* dims4,
                           int* aux);
size_t stabnet_net_param_floats(const void* net);
size_t stabnet_net_trainable_floats(const void* net);
size_t stabnet_net_bn_channels(const void* net);
size_t stabnet_net_workspace_bytes(const void* net);
double stabnet_net_flops(const void* net);
int stabnet_net_num_launches(const void* net);
/* Kernel launches of one stabnet_deploy_frame pass (refine = 1) on the CURRENT device: stack assembly + the regressor's
 * launches + (mesh, unless it rides with the output layer's launch: batch <= 8) + sampler-with-push; -1 on a null plan. */
int stabnet_deploy_frame_launches(const void* net, int grid_h, int grid_w);
int stabnet_net_activation_info(const void* net, const char* name, long* offset, int* dims4);

/* slim batch_norm(is_training=False) folded to per-channel (scale, shift): fold = [G scales][G shifts],
 * G = stabnet_net_bn_channels().  scale = rsqrt(var + eps) * gamma, shift = beta - mean * scale. */
int stabnet_net_fold_bn(const void* net, const float* params, float* fold, float eps, void* stream);
/* Floats the caller must allocate for `fold`: 2*bn_channels, plus the re-laid-out stem weights of an inference plan
 * (keep_activations = 0), which reads the 13-channel stack directly (no channel padding). */
size_t stabnet_net_fold_floats(const void* net);

/* x_tensor NHWC [N,H,W,in_ch] -> theta [N,n_theta]; BN in moving-average mode (s_net_bundle_nobm.py:302). */
int stabnet_backbone_fwd_infer(const void* net, const float* params, const float* fold, const float* x_tensor,
                               float* theta, void* workspace, size_t workspace_bytes, void* stream, void* prof);

/* ---- the online loop (deploy_bundle.py) ------------------------------------------------------------------ */

/* History ring initialisation: `depth` copies of the first frame and zero masks per stream
 * (deploy_bundle.py:216-224).  frames_ring, masks_ring: [S][depth][H*W]; first_frame [S][H*W]. */
int stabnet_ring_init(float* frames_ring, float* masks_ring, const float* first_frame, int S, int depth, int H, int W,
                      void* stream);

/* One iteration of the hot loop for S = net.N independent streams (deploy_bundle.py:259-296,319-332), the work of
 * one sess.run([output, black_pix, Hs, x_map, y_map], {x_tensor: in_x}) plus the NumPy stack assembly before it and
 * the feedback after it:  13-channel stack from the ring at the dilated `lags` (HOST int array, e.g. 1,2,4,8,16,32;
 * channel order masks, frames, current) -> regressor -> get_4_pts -> transformer -> frame = img - black ->
 * frames_ring[head] = frame, masks_ring[head] = black.  refine > 1 repeats the network on the refined frame
 * (:284-295).  `head` is a DEVICE int[2] = {ring slot of this frame's push, reserved (zero)}; the call
 * advances head[0] to (head+1) % depth on the device (refine = 1: by one thread of the mesh kernel, which sits between
 * the last reader of the head -- the stack assembly -- and the sampler, whose fused feedback push then writes slot
 * head - 1; else by a one-thread kernel at the end), so every argument is fixed across frames
 * and the call can be captured into a hipGraph once and replayed (copy the new frame into the fixed `cur_frame` buffer
 * before each replay).  all_black (optional, may be NULL): int32 [S,H,W] += round(black) once per refine pass
 * (deploy_bundle.py:291, inside the refine loop) -- the input of stabnet_crop_search.
 * Outputs: theta [S,n_theta]; out_img, black, x_map, y_map, frame_fb [S,H,W]; Hs [S,gh,gw,9]. */
int stabnet_deploy_frame(const void* net, const float* params, const float* fold, float* frames_ring,
                         float* masks_ring, int depth, int* head, const int* lags, int n_lags, const float* cur_frame,
                         int refine, int grid_h, int grid_w, float do_crop_rate, float* theta, float* out_img,
                         float* black, float* x_map, float* y_map, float* Hs, float* frame_fb, int* all_black,
                         void* workspace, size_t workspace_bytes, void* stream, void* prof);

/* ---- optional per-launch timing (bench.py roofline leg) --------------------------------------------------
 * A profiler handle owns HIP events (host objects).  Passing it as `prof` to a forward makes that call record an
 * event pair around each kernel launch; read the records after synchronising the stream.  NULL = no instrumentation. */
int stabnet_prof_create(void** prof, int max_records);
void stabnet_prof_destroy(void* prof);
int stabnet_prof_reset(void* prof);
int stabnet_prof_record_empty(void* prof, void* stream);   /* an event pair around nothing: the overhead to subtract */
int stabnet_prof_num_records(const void* prof);
int stabnet_prof_record(const void* prof, int idx, int* kind, float* ms, double* flops, double* bytes);
int stabnet_prof_record_shape(const void* prof, int idx, int* shape4);
const char* stabnet_prof_kind_name(int kind);

/* ---- measurement helpers (bench.py: empirical peaks of the box beside the vendor peaks; not on the path) ---- */
int stabnet_probe_mfma_f32(float* out /* blocks*256 floats */, int blocks, int iters,
                           unsigned long long* stamps /* optional, device [blocks][2]: {shader cycles, 100 MHz ticks} in the loop */,
                           void* stream);
double stabnet_probe_mfma_f32_flops(int blocks, int iters);
int stabnet_probe_hbm_copy(const float* src, float* dst, long n_floats, void* stream);
/* Stand-in for a collective's kernel on a one-GPU box: dst[i] += src[i] by `workgroups` long-lived
 * workgroups of 256 threads holding lds_bytes (<= 64 KiB) of LDS each; stamps (optional, device, 2 words per workgroup):
 * 100 MHz ticks at its entry and exit.  Not on the product path. */
int stabnet_probe_comm_proxy(const float* src, float* dst, long n_floats, int workgroups, int lds_bytes, unsigned long long* stamps,
                             void* stream);
/* CUs the persistent convolution kernels leave free (0 = none, the default): their grids are sized for (CUs - reserved) x
 * workgroups-per-CU, so that a collective's kernel on a communication stream finds CU slots without waiting for a kernel
 * boundary.  A process-wide setting read at launch time (train.Trainer sets it from STABNET_COMM_RESERVED_CUS when it has a
 * process group); returns the previous value. */
int stabnet_conv_reserve_cus(int reserved);

/* ---- training: backward of the warp / sampler and the loss kernels ---------------------------------------
 * These replace what TF autodiff generates for `opt.minimize(total_loss)` (train_bundle_nobm.py:160) over the ops
 * above.  floor / casts / comparisons carry no gradient (corners, black_pix, z sign, warp_pts indices are constants). */

/* Reproducibility: every reduction below is order-independent (64-bit fixed-point accumulation, scale 2^40, or block
 * partials added in a fixed order) -- two runs of a training step on the same inputs give the same bits.
 *
 * d transformer / d pts2 (pre-clip vertex gradient) [N,gh+1,gw+1,2] from d_out [N,H,W,C], d_xmap, d_ymap [N,H,W]
 * (each may be NULL); dmap_scale [N] (optional) multiplies d_xmap / d_ymap per sample (stabnet_feature_loss hands over
 * signed counts and that factor).  x_map, y_map, Hs: the forward's outputs.  workspace: N*gh*gw*8 + 1 8-byte words (8-B aligned).
 * A non-finite or out-of-range (|v| >= 2^22) contribution poisons the sums: the gradients then come out NaN, not finite garbage. */
int stabnet_transformer_bwd(const float* pts2, const float* Hs, const float* U, const float* x_map, const float* y_map,
                            const float* d_out, const float* d_xmap, const float* d_ymap, const float* dmap_scale, int N,
                            int H, int W, int C, int grid_h, int grid_w, float* d_pts2, void* workspace, void* stream);

/* d interpolate(im, x, y) / d im  (train_bundle_nobm.py:117-118): scatter-add of the four taps, d_im (+)= it.
 * workspace: N*H*W*C + 1 8-byte words (8-B aligned); non-finite contributions give NaN (see stabnet_transformer_bwd). */
int stabnet_interp_bwd(const float* x, const float* y, const float* d_out, int N, int H, int W, int C, float* d_im,
                       int accumulate, void* workspace, void* stream);

/* Re-packing helpers of the host mirror: out [npix] = x[npix][C][:, c] (x_tensor[..., 12:13] of s_net_bundle_nobm.py:281,
 * flow[..., 0]); out [n][2] = (a, b) interleaved (img = [x_map, y_map], spatial_transformer3.py:295). */
int stabnet_slice_channel(const float* x, long npix, int C, int c, float* out, void* stream);
int stabnet_interleave2(const float* a, const float* b, long n, float* out, void* stream);

/* y = a*x + b elementwise (1 - black_pix of train_bundle_nobm.py:118). */
int stabnet_axpb(const float* x, float a, float b, long n, float* y, void* stream);

/* masked MSE of img_loss (s_net_bundle_nobm.py:347-352, m2 = NULL) and temp_loss (train_bundle_nobm.py:110-125,
 * m2 = interp(1 - black2)): sums [N,2] = {sum((a-b)m)^2, sum m}, m = (1 - black) * m2;
 * loss = sum_n sums[n][0] / (sums[n][1] + 1e-8) / batch_size.  _grad: ga (+)= coef * dloss_unnormalised/da, gb = -that. */
size_t stabnet_masked_mse_workspace_bytes(int N);
int stabnet_masked_mse_sums(const float* a, const float* b, const float* black, const float* m2, int N, long hw,
                            float* sums, void* workspace, void* stream);
int stabnet_masked_mse_grad(const float* a, const float* b, const float* black, const float* m2, const float* sums,
                            float coef, int N, long hw, float* ga, int accumulate_a, float* gb, void* stream);

/* feature loss (s_net_bundle_nobm.py:215-230,335-343): value [N] = masked mean L1 between the maps gathered at the
 * rounded stable point and the unstable point; its gradient wrt the maps = d_xmap / d_ymap (signed counts +-mask scattered
 * at the rounded pixels; zeroed here; both NULL = forward only) times dscale [N] = gcoef / max(sum mask, 1);
 * warped [N,max_matches,2] optional (ret['stable_warpped']). */
int stabnet_feature_loss(const float* matches, const float* mask, const float* x_map, const float* y_map, int N, int H,
                         int W, int max_matches, float gcoef, float* value, float* d_xmap, float* d_ymap, float* dscale,
                         float* warped, void* stream);

/* losses4 = {id2_loss (= mean|theta| * id_mul, :263), black_pos mean (:139-146,312-317), distortion (:148-181),
 * consistency (:183-210)}; d_theta [N,n_theta] = clip-mask (:58) * (d_pts2_warp + w_dist d dist + w_cons d cons +
 * w_black d black) + w_id * d id2_loss (d_theta may be NULL: values only). */
int stabnet_mesh_losses(const float* theta, const float* d_pts2_warp, int N, int grid_h, int grid_w, float do_crop_rate,
                        float id_mul, float w_id, float w_dist, float w_cons, float use_black, float w_black,
                        float* losses4, float* d_theta, void* stream);

/* ---- training: convolution backward (autodiff of slim conv2d) --------------------------------------------- */

/* dW (OHWI, ACCUMULATED into, not zeroed) += d conv2d / d weights.  x: forward input; (in_scale,in_shift): the forward's
 * folded-BN + ReLU prologue (both or NULL); dy [N,Ho,Wo,Cout].  Exact float32 MFMA; the pixel range is split over
 * workgroups whose partial tiles go to slabs in `workspace` and are added in split order (reproducible, no atomics).
 * dw (and d_bias below) must be 16-byte aligned: the slab reduction updates them 16 B at a time (STABNET_ERR_BAD_ARG otherwise). */
size_t stabnet_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int stabnet_conv2d_wgrad(const float* x, const float* dy, float* dw, const float* in_scale, const float* in_shift,
                         int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* workspace,
                         size_t workspace_bytes, void* stream);

/* 1x1 stride-1 layers with Cout % 256 == 0: dW as above and d_bias [Cout] += column sums of dy from the same pass over dy
 * (what the training step does for the unit-closing convolutions).  Other geometries: STABNET_ERR_BAD_ARG. */
size_t stabnet_conv2d_wgrad_bias_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int stabnet_conv2d_wgrad_bias(const float* x, const float* dy, float* dw, float* d_bias, const float* in_scale, const float* in_shift,
                              int N, int H, int W, int Cin, int Cout, void* workspace, size_t workspace_bytes, void* stream);

/* The same for a channel count that is not a multiple of 4 (the 13-channel stem input): x [N,H,W,Cin] tight,
 * dw OHWI [Cout][KH][KW][CinPad] accumulated into (pad channels untouched).  x is embedded in a zero-bordered image inside
 * `workspace` and read as runs of KW*Cin contiguous floats per filter row -- the operand layout of the training step's stem. */
size_t stabnet_conv2d_wgrad_rowrun_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int stabnet_conv2d_wgrad_rowrun(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int CinPad, int Cout,
                                int KH, int KW, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);

/* dx [N,H,W,Cin] = d conv2d / d input (+ residual if given, may alias dx) from dy [N,Ho,Wo,Cout] and the forward
 * weights (OHWI).  Cout % 16 == 0.  workspace: stabnet_conv2d_dgrad_workspace_bytes() (re-packed weights + split-K). */
size_t stabnet_conv2d_dgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int stabnet_conv2d_dgrad(const float* dy, const float* w_ohwi, float* dx, const float* residual, int N, int H, int W,
                         int Cin, int Cout, int KH, int KW, int stride, int pad, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- training: the regressor tower (get_resnet(is_training=True) and its autodiff) -------------------------
 * The plan must be created with keep_activations = 1.  One workspace per tower (activations are kept between the
 * forward and the backward of the same tower; the siamese step runs two towers, train_bundle_nobm.py:107-108). */
size_t stabnet_net_train_workspace_bytes(const void* net);

/* Forward with batch-statistics BN (tf.nn.moments, biased variance); updates the moving averages stored in `params`
 * (slim UPDATE_OPS tied to the step, s_net_bundle_nobm.py:355-356).  x_tensor [N,H,W,in_ch] -> theta [N,n_theta]. */
int stabnet_tower_fwd_train(const void* net, float* params, const float* x_tensor, float* theta, void* workspace,
                            size_t workspace_bytes, float bn_eps, float bn_decay, void* stream, void* prof);

/* Both siamese towers of a step (train_bundle_nobm.py:107-108: two towers over the same weights) in LOCKSTEP, layer by layer:
 * per layer ONE convolution launch over both towers' batches where the tower's rows are a multiple of the 64-row tile (else the
 * convolution of tower 1, then of tower 2), and ONE batch-statistics reduction launch covering both.  The results are those of
 * tower_fwd_train(x1) then tower_fwd_train(x2) up to float32 summation order (the pair's launch may choose another split-K);
 * the moving averages receive tower 1's update, then tower 2's.  One workspace per tower. */
int stabnet_towers_fwd_train(const void* net, float* params, const float* x1, const float* x2, float* theta1, float* theta2,
                             void* workspace1, void* workspace2, size_t workspace_bytes, float bn_eps, float bn_decay,
                             void* stream, void* prof);

/* Backward from d_theta [N,n_theta]; parameter gradients are ACCUMULATED into grads (layout = trainable prefix of
 * params; zero once per step). */
int stabnet_tower_bwd(const void* net, const float* params, const float* d_theta, float* grads, void* workspace,
                      size_t workspace_bytes, void* stream, void* prof);
/* The same backward in stabnet_net_num_grad_stages() stages (0: FC head + block4, 1: block3, 2: block2, 3: block1 + stem;
 * call them in this order).  When stage k's kernels are done the gradients in [lo, hi) of stabnet_net_grad_bucket(k) are
 * final for this tower -- a data-parallel host hands that bucket to the collective while the earlier layers are still in
 * backward (reverse layer order).  The BN gamma / beta sections (stabnet_net_bn_grad_range) are final after the last stage. */
int stabnet_tower_bwd_stage(const void* net, const float* params, const float* d_theta, float* grads, void* workspace,
                            size_t workspace_bytes, int stage, void* stream, void* prof);
/* One backward stage of BOTH towers in lockstep (after stabnet_towers_fwd_train): the bucket of stage k then holds the sum of
 * both towers' gradients.  The dgrad weights are re-packed once, the wgrad slabs of both towers are reduced together.
 * PAIRING (checked, STABNET_ERR_BAD_ARG otherwise): the lockstep forward keeps the FC-head activations of both towers in
 * workspace1, the single-tower forward in its own workspace, so stabnet_towers_bwd_stage(ws1, ws2) must follow
 * stabnet_towers_fwd_train(ws1, ws2) on the same two workspaces, and stabnet_tower_bwd / _bwd_stage(ws) must follow
 * stabnet_tower_fwd_train(ws). */
int stabnet_towers_bwd_stage(const void* net, const float* params, const float* d_theta1, const float* d_theta2, float* grads,
                             void* workspace1, void* workspace2, size_t workspace_bytes, int stage, void* stream, void* prof);
int stabnet_net_num_grad_stages(void);
int stabnet_net_grad_bucket(const void* net, int stage, long* lo, long* hi);
int stabnet_net_bn_grad_range(const void* net, long* lo, long* hi);

/* Float offsets, inside a tower workspace, of the batch BN buffers [G] (scale, shift, mean, invstd) of the last forward. */
int stabnet_net_train_bn_offsets(const void* net, long* scale_off, long* shift_off, long* mean_off, long* invstd_off);

/* Debug view of a training workspace after a forward (tests read the forward's discrete decisions -- ReLU signs, max-pool
 * argmax -- back from it): what = "bn:<channel offset>" (the tensor that BN normalises: float offset, M*C elements),
 * "fcx0".."fcx3" (input of FC layer k of the PAIR, [2N][dims[k]], tower 0's workspace), "argmax" (float offset of the pool's
 * argmax bytes, byte count), "pool". */
int stabnet_net_train_debug_offset(const void* net, const char* what, long* off, long* count);

/* slim L2 regularisers (REGULARIZATION_LOSSES, s_net_bundle_nobm.py:324-325; resnet.py:35-37): *loss_out +=
 * sum_seg coef*0.5*sum w^2 (NULL to skip), grads[seg] += gscale*coef*w (NULL to skip).  seg_* are DEVICE arrays.
 * workspace: 64*nseg floats (block partials of the value; may be NULL when loss_out is). */
int stabnet_weight_decay(const float* params, float* grads, const long* seg_off, const long* seg_len,
                         const float* seg_coef, int nseg, float gscale, float* loss_out, float* workspace, void* stream);

/* tf.train.AdamOptimizer step (train_bundle_nobm.py:155-160): g = (grads + grads2) * gscale (grads2 may be NULL). */
int stabnet_adam_step(float* params, const float* grads, const float* grads2, float* m, float* v, long n, float lr,
                      float beta1, float beta2, float eps, int step, float gscale, void* stream);

/* ---- next to the path: colour-frame remap with smoothed maps --------------------------
 * warpRevBundle2(img, x_map, y_map) (deploy_bundle.py:136-146,303): cv2.resize of both maps down by `rate` and back
 * up (INTER_LINEAR), (m+1)/2*size, cv2.remap(img, ., ., INTER_LINEAR) of the uint8 BGR frame.  img, out uint8
 * [N,H,W,C]; x_map, y_map [N,H,W]; workspace 2*N*(H/rate)*(W/rate) floats; px_out, py_out optional [N,H,W]. */
int stabnet_warp_rev_bundle2(const unsigned char* img, const float* x_map, const float* y_map, int N, int H, int W, int C,
                             int rate, unsigned char* out, float* workspace, float* px_out, float* py_out, void* stream);

/* cvt_train2img (deploy_bundle.py:75): the network's grey output back to 8 bits, out[i] = uint8((x[i] + 0.5) * 255) clipped to
 * [0, 255].  x float [n], out uint8 [n].  The 16-byte path needs both pointers 16-byte aligned (any alignment is accepted). */
int stabnet_cvt_train2img(const float* x, unsigned char* out, long n, void* stream);

/* ---- next to the path: max-inscribed-rectangle crop --------------------------------
 * deploy_bundle.py:291: all_black += round(black) per frame;  :344-366: once per video, the largest black-free rectangle
 * whose top-left corner lies on the `step` (10) grid of the top-left quadrant, first-found-wins on ties. */
int stabnet_black_accumulate(const float* black, int* all_black, long n, void* stream);
size_t stabnet_crop_search_workspace_bytes(int H, int W, int step);
int stabnet_crop_search(const int* all_black, int H, int W, int step, int* ans5, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ---- next to the path: training sample assembly -------------------------------------
 * read_and_decode's tensor part (get_data_mini_after.py:229-253) for N pairs at once: warp_img (:14-31: bilinear up-scale
 * by 1/random_crop_rate, crop, flip, tf.image contrast + brightness, clip) on the 2*(before_ch+1) stable and 2 unstable
 * channels, add_mask (:93-147: random-homography black masks, masked pixels = -1), warp_flow (:33-51), warp_point
 * (:53-70).  The random draws are INPUTS (device arrays): para [N][3] = crop row, crop column, flip; jitter [N][2] =
 * contrast factor, brightness delta; Hs [N][2][before_ch][9] mask homographies (tower, channel).
 * stable [N,H,W,2*(before_ch+1)] (label, history x before_ch, per tower), unstable [N,H,W,2], flow [N,H,W,2] (or NULL),
 * matches [N,max_matches,4] + n [N] valid counts (or NULL).  Outputs NHWC: x1,x2 [N,H,W,2*before_ch+1] (masks, masked
 * history, current), y1,y2 [N,H,W,1], flow_out, fm [N,max_matches,4], mk [N,max_matches] (0/1 floats). */
size_t stabnet_augment_workspace_bytes(int N, int H, int W, int before_ch);
int stabnet_augment_pairs(const float* stable, const float* unstable, const float* flow_in, const float* matches1,
                          const int* n1, const float* matches2, const int* n2, const int* para, const float* jitter,
                          const float* Hs, int N, int H, int W, int before_ch, int max_matches, float random_crop_rate,
                          float* x1, float* y1, float* x2, float* y2, float* flow_out, float* fm1, float* mk1, float* fm2,
                          float* mk2, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STABNET_HIP_H */
