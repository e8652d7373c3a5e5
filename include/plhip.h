/*
 * plhip.h — C ABI of the MI355X (gfx950) INT8 conv / depthwise / fc / calib backend.
 *
 * This is the drop-in boundary underneath the TARGET(kHIP)/PRECISION(kInt8) kernel classes in
 * paddle-lite_amd/lite/kernels/hip/ (which mirror lite/kernels/arm/{conv,fc,calib}_compute.cc of the
 * reference).  Each entry point names the reference interface it replaces (paths relative to the
 * reference tree).  Conventions:
 *   - every data pointer is a DEVICE pointer unless the name says host; the caller owns all buffers,
 *     the library never frees caller memory and allocates nothing inside a launch function;
 *   - calls are asynchronous on the context's HIP stream; plhip_stream_sync() waits;
 *   - return 0 on success, a negative plhip_status otherwise; the library never aborts (the C++
 *     kernel layer turns non-zero into LOG(FATAL), like CUDA_CALL in lite/backends/cuda/cuda_utils.h);
 *   - one plhip_ctx per host thread and GPU, not shared between threads
 *     (shape of lite/backends/cuda/context.h:35-140).
 * Tensors are dense NCHW; activations/weights int8, bias/scale fp32.  No torch types here.
 */
#ifndef PLHIP_H_
#define PLHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  PLHIP_OK = 0,
  PLHIP_ERR_INVALID = -1,     /* bad argument / inconsistent descriptor */
  PLHIP_ERR_HIP = -2,         /* a HIP runtime call failed (see plhip_last_error) */
  PLHIP_ERR_UNSUPPORTED = -3, /* configuration outside the implemented paths */
  PLHIP_ERR_WORKSPACE = -4,   /* workspace missing or too small */
  PLHIP_ERR_NO_DEVICE = -5    /* no HIP device / device is not gfx950 */
} plhip_status;

/* Output kind.  I32_ACC exists for the bit-exact accumulator check only. */
/* PLHIP_OUT_F32_GAP: the fp32 output averaged over each output plane, y = [n][cout] floats: the instruction pair
 * conv2d[fp32_out] -> pool2d(avg, global_pooling) (lite/backends/arm/math/pooling.cc:1006- pooling_global_avg) in the launch that
 * produces the plane; accepted by plhip_dwpw_fused_int8 / plhip_dwpw_fused_supported only (every other entry refuses it). */
typedef enum { PLHIP_OUT_I32_ACC = 0, PLHIP_OUT_F32 = 1, PLHIP_OUT_I8 = 2, PLHIP_OUT_F32_GAP = 3 } plhip_out_kind;

/* Activation codes == lite_api::ActivationType, lite/api/paddle_place.h:101-105. */
typedef enum { PLHIP_ACT_NONE = 0, PLHIP_ACT_RELU = 1, PLHIP_ACT_RELU6 = 2, PLHIP_ACT_LEAKY_RELU = 4 } plhip_act;

/* The kernel's whole shape contract == the fields of operators::ConvParam the ARM kernels read
 * (lite/operators/op_params.h:446-502): x dims, filter dims (OIHW, I = cin/groups), paddings
 * {top,bottom,left,right}, strides {h,w}, dilations {h,w}, groups, activation_param. */
typedef struct {
  int n, cin, h, w;
  int cout, kh, kw;
  int pad[4];
  int stride[2];
  int dil[2];
  int groups;
  int act;         /* plhip_act */
  float act_alpha; /* relu6: clip coefficient (already divided by out_scale for int8-out,
                      conv_gemmlike.cc:259-263); leaky: negative slope */
} plhip_conv_desc;

typedef struct plhip_ctx plhip_ctx;

/* ---- context / memory: replaces TargetWrapper<kCUDA> + CUDAContext for the new target
 *      (lite/backends/cuda/target_wrapper.h:27-85, lite/backends/cuda/context.h:35-140,
 *       lite/core/memory.{h,cc} TargetMalloc/TargetFree/TargetCopy). ---- */
int plhip_device_count(void);
plhip_status plhip_ctx_create(int device_id, plhip_ctx** out);
/* Adopt an existing hipStream_t (e.g. the caller's framework stream); not destroyed with the ctx. */
plhip_status plhip_ctx_create_on_stream(int device_id, void* hip_stream, plhip_ctx** out);
void plhip_ctx_destroy(plhip_ctx* ctx);
void* plhip_ctx_stream(plhip_ctx* ctx);
const char* plhip_last_error(plhip_ctx* ctx);
plhip_status plhip_malloc(plhip_ctx* ctx, size_t bytes, void** dev_ptr);
plhip_status plhip_free(plhip_ctx* ctx, void* dev_ptr);
plhip_status plhip_memcpy_h2d(plhip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
plhip_status plhip_memcpy_d2h(plhip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
plhip_status plhip_memcpy_d2d(plhip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
plhip_status plhip_memset(plhip_ctx* ctx, void* dst_dev, int value, size_t bytes);
plhip_status plhip_stream_sync(plhip_ctx* ctx);
/* ---- launch graphs (no counterpart in the reference: its CUDA backend launches kernel by kernel) ----
 * Every compute entry point is asynchronous, allocation-free and argument-static, so a whole program step can be recorded
 * once and replayed: plhip_graph_begin puts the context's stream into capture (hipStreamBeginCapture, thread-local mode);
 * the caller issues its launches as usual; plhip_graph_end ends the capture and instantiates an executable graph;
 * plhip_graph_launch replays it on the context's stream (one submission instead of ~30, no per-kernel dispatch gap).
 * Pointers baked into the recorded launches must stay valid; anything that synchronises or allocates is illegal between
 * begin and end (the first, un-captured run of a program does the one-time work: weight packing, workspace, attributes). */
plhip_status plhip_graph_begin(plhip_ctx* ctx);
plhip_status plhip_graph_end(plhip_ctx* ctx, void** graph_exec);
plhip_status plhip_graph_launch(plhip_ctx* ctx, void* graph_exec);
plhip_status plhip_graph_destroy(plhip_ctx* ctx, void* graph_exec);
/* hipEvent timing on the context's stream (DeviceTimer<kCUDA> analogue, lite/core/profile/timer.h:127-158). */
plhip_status plhip_event_create(plhip_ctx* ctx, void** event);
plhip_status plhip_event_record(plhip_ctx* ctx, void* event);
plhip_status plhip_event_elapsed_ms(plhip_ctx* ctx, void* start, void* stop, float* ms);
plhip_status plhip_event_destroy(plhip_ctx* ctx, void* event);

/* ---- dense / grouped conv2d ----
 * Replaces: GemmLikeConv / DirectConv / WinogradConv <kInt8,*>::Run (lite/kernels/arm/conv_gemmlike.cc:
 * 325-462, conv_direct.cc:110-239, conv_winograd.cc:224-475) -> conv1x1s1_gemm_int8 / conv_im2col_gemm_int8
 * (lite/backends/arm/math/conv_impl.cc:260-331, 490-598) -> gemm_prepack_int8 (gemm_prepacked_int8.cc:
 * 5263-5457) with its fused epilogue (:643-796).
 *
 * Weight pre-pack replaces prepackA_int8 (gemm_prepacked_int8.cc:109-224) / trans_gemm_weights<kInt8>
 * (conv_block_utils.h:65-73): OIHW int8 -> per-group MFMA A-fragment order, zero padded. */
size_t plhip_conv_packed_weight_bytes(const plhip_conv_desc* d);
plhip_status plhip_pack_conv_weights(plhip_ctx* ctx, const plhip_conv_desc* d,
                                     const int8_t* w_oihw, void* w_packed);
/* Scratch (0 for 1x1 s1 p0 and the small-Cin 3x3 / 7x7 s2 stems): the zero-padded input copy of the implicit-GEMM route
 * (dense k x k, stride 1 or 2, wide enough GEMM: ~1.1x the input) or the im2col buffer (everything else: kh*kw x the
 * input); replaces ctx.workspace_data (conv_gemmlike.cc:131).  The caller passes at least this many bytes, 16-byte aligned
 * (4 suffices for every route but the dense 3x3 patch kernel's padded / phase-split copy). */
size_t plhip_conv_workspace_bytes(const plhip_conv_desc* d);
/* scale/bias: folded per-output-channel fp32 arrays of length cout; bias may be NULL
 * (treated as zeros).  y: int32 / float / int8 NCHW according to `out`.  For PLHIP_OUT_I32_ACC scale,
 * bias and activation are ignored.  Name of the path taken (kernel_func_name analogue,
 * conv_gemmlike.cc:384): plhip_conv_impl_name(). */
plhip_status plhip_conv2d_int8(plhip_ctx* ctx, const plhip_conv_desc* d, const int8_t* x,
                               const void* w_packed, const float* scale, const float* bias, void* y,
                               plhip_out_kind out, void* workspace, size_t workspace_bytes);
const char* plhip_conv_impl_name(const plhip_conv_desc* d);
/* fp32-output conv with a fused graph tail (graph-level fusion on this target).  Replaces the
 * instruction run  conv2d[fp32_out] -> elementwise_add | fusion_elementwise_add_activation(relu) -> calib[fp32_to_int8]
 * of the reference's ResNet50 / MobileNetV2 programs (lite/kernels/arm/{conv,elementwise,calib}_compute.cc), any suffix
 * of it:   v = act(fma(float(acc), scale, bias));   if (residual) v = v + residual[i];   if (residual_relu) v = max(v, 0);
 *          if (y_f32) y_f32[i] = v;   if (y_i8) y_i8[i] = sat8(round_half_away(v * (1.f / calib_scale)))
 * Every value is rounded exactly as the separate instructions round it, so the results are bit-identical to running
 * them one by one.  residual / y_f32 / y_i8 have the output's NCHW shape; y_f32 may be NULL when only the int8 copy has
 * consumers.  The field ConvParam::residualData of the reference (lite/operators/op_params.h:446-502) is the operand. */
plhip_status plhip_conv2d_int8_fused(plhip_ctx* ctx, const plhip_conv_desc* d, const int8_t* x, const void* w_packed,
                                     const float* scale, const float* bias, float* y_f32, const float* residual,
                                     int residual_relu, int8_t* y_i8, float calib_scale, void* workspace,
                                     size_t workspace_bytes);

/* ---- calib[fp32_to_int8] + conv2d in one launch (graph-level fusion on this target) ----
 * Replaces the instruction pair  calib[fp32_to_int8](scale) ; conv2d 3x3 s2 (Cin <= 3)  at the head of the MobileNet programs
 * (lite/kernels/arm/calib_compute.cc:25-40 -> type_trans.cc:34-187 ; lite/kernels/arm/conv_direct.cc): x_f32 is the calib's
 * fp32 input [n, cin, h, w], calib_scale its scale; every value is quantised exactly as plhip_calib_f32_to_i8 does and
 * convolved exactly as plhip_conv2d_int8 does: bit-identical to the two calls, the int8 image never exists.  w_packed: what
 * plhip_pack_conv_weights made for `d`.  plhip_conv2d_calib_supported: 1 where the fused kernel takes the conv (3x3 stride 2,
 * cin <= 3, left padding 1, top padding <= 1, w % 4 == 0, ow % 4 == 0), else the caller runs the two instructions. */
int plhip_conv2d_calib_supported(const plhip_conv_desc* d);
plhip_status plhip_conv2d_calib_int8(plhip_ctx* ctx, const plhip_conv_desc* d, const float* x_f32, float calib_scale,
                                     const void* w_packed, const float* scale, const float* bias, void* y, plhip_out_kind out);

/* ---- depthwise conv (groups == cin == cout) ----
 * Replaces: DepthwiseConv<kInt8,*>::Run (lite/kernels/arm/conv_depthwise.cc:357-446) ->
 * conv_depthwise_3x3_int8_{fp32,int8} / conv_depthwise_5x5_int8_{fp32,int8}
 * (lite/backends/arm/math/conv_impl.cc:798-1184).  Weights are the raw OIHW [C,1,kh,kw] filter
 * (no re-layout needed on this target; cf. conv_trans_weights_numc, conv_block_utils.h:89). */
plhip_status plhip_depthwise_conv_int8(plhip_ctx* ctx, const plhip_conv_desc* d, const int8_t* x,
                                       const int8_t* w_oihw, const float* scale, const float* bias,
                                       void* y, plhip_out_kind out);

/* ---- fused depthwise 3x3 [int8_out] -> pointwise 1x1 (graph-level fusion) ----
 * Replaces the instruction pair  depthwise_conv2d[int8_out] ; conv2d 1x1 s1 p0 g1  of the MobileNet programs
 *  when the depthwise output has no other consumer.  Result is bit-identical to running
 * plhip_depthwise_conv_int8 (PLHIP_OUT_I8) followed by plhip_conv2d_int8; the int8 intermediate never leaves the CU.
 * dw: descriptor of the depthwise conv (groups == cin == cout, 3x3, stride 1|2, dilation 1); dw_scale / dw_bias: its
 * folded int8-out scale / bias; pw_cout, pw_w_packed (from plhip_pack_conv_weights of the 1x1 conv), pw_scale / pw_bias,
 * pw_act / pw_alpha describe the pointwise conv; y: [n, pw_cout, oh, ow] of kind `out`.
 * Returns PLHIP_ERR_UNSUPPORTED when the shape is outside the fused path (caller falls back to the two calls):
 * 3x3, stride 1 | 2, dilation 1, <= 1024 channels, a 128..1024-column tile touching <= 4 images, and 4 K-steps of the
 * tile's staged input rows (32 channels each) fitting the CU's LDS — plhip_dwpw_fused_supported answers that up front. */
int plhip_dwpw_fused_supported(const plhip_conv_desc* dw, int pw_cout, plhip_out_kind out);
plhip_status plhip_dwpw_fused_int8(plhip_ctx* ctx, const plhip_conv_desc* dw, const int8_t* x, const int8_t* dw_w_oihw,
                                   const float* dw_scale, const float* dw_bias, int pw_cout, const void* pw_w_packed,
                                   const float* pw_scale, const float* pw_bias, int pw_act, float pw_alpha, void* y,
                                   plhip_out_kind out);

/* ---- fc ----
 * Replaces: FcCompute<kInt8,*>::Run (lite/kernels/arm/fc_compute.cc:229-344) -> gemm_s8 / gemv_int8
 * (lite/backends/arm/math/gemm_s8.cc:23-47, gemv_arm_int8.cc:701-760).
 * x [m,k] int8 row-major; w [k,n] int8 (Paddle "mul" layout); scale/bias per output column n.
 * Pre-pack (an opaque block of plhip_fc_packed_weight_bytes: a [k/4][n][4] copy for the dot4 kernels followed by the
 * MFMA A-fragment order, both zero padded) replaces the weight transpose of fc_compute.cc:53-62. */
/* `relu` is a flag word: bit 0 = fused relu; bit 1 (fp32 output only) = round twice, float(acc)*scale then + bias, as
 * the reference's gemm_s8 + fill_bias_fc route does (fc_compute.cc:250-266, taken there when m > 1 and the weight
 * scale is a single value); without it the epilogue is the single fused multiply-add of its gemv route. */
size_t plhip_fc_packed_weight_bytes(int k, int n);
plhip_status plhip_pack_fc_weights(plhip_ctx* ctx, int k, int n, const int8_t* w_kn, void* w_packed);
plhip_status plhip_fc_int8(plhip_ctx* ctx, int m, int k, int n, const int8_t* x, const void* w_packed,
                           const float* scale, const float* bias, int relu, void* y, plhip_out_kind out);

/* ---- calib (graph-edge precision casts) ----
 * Replaces: CalibComputeFp32ToInt8 / Int8ToFp32 (lite/kernels/arm/calib_compute.cc:25-57) ->
 * fp32_to_int8 / int8_to_fp32 (lite/backends/arm/math/type_trans.cc:34-187, 268-371), single scale. */
plhip_status plhip_calib_f32_to_i8(plhip_ctx* ctx, const float* x, int8_t* y, float scale, int64_t count);
plhip_status plhip_calib_i8_to_f32(plhip_ctx* ctx, const int8_t* x, float* y, float scale, int64_t count);

/* ---- fp32 glue ops of the MobileNet graph kept on device ----
 * Replaces: PoolCompute global-avg (lite/backends/arm/math/pooling.cc:1006-) and SoftmaxCompute
 * (lite/backends/arm/math/softmax.cc) for the tail pool2d -> calib -> fc -> softmax. */
plhip_status plhip_global_avg_pool_f32(plhip_ctx* ctx, const float* x, int nc, int spatial, float* y);
plhip_status plhip_softmax_f32(plhip_ctx* ctx, const float* x, int rows, int cols, float* y);

/* ---- fp32 glue ops of the ResNet50 / MobileNetV2 programs (both ops are fp32-only on the
 * reference's ARM target, so residual edges de/re-quantise through calib) ----
 * pool2d replaces PoolCompute::Run (lite/kernels/arm/pool_compute.cc:36-345) -> pooling_basic and its specialisations
 * (lite/backends/arm/math/pooling.cc:38-215): max, or avg with the exclusive flag, over windows clipped to the image;
 * paddings {top, bottom, left, right}; output dims are the caller's (PoolOutputSize, lite/operators/pool_op.h, incl.
 * ceil_mode).  x [n*c, h, w] -> y [n*c, oh, ow]. */
typedef struct {
  int planes, h, w, oh, ow;
  int kh, kw;
  int pad[4];
  int stride[2];
  int is_max;     /* 1 max, 0 avg */
  int exclusive;  /* avg: divide by the clipped window size (pooling.cc:160-164) */
} plhip_pool_desc;
plhip_status plhip_pool2d_f32(plhip_ctx* ctx, const plhip_pool_desc* d, const float* x, float* y);
/* int8 max pool: the kHIP graph fusion turns conv2d[fp32_out] -> pool2d(max) -> calib into conv+calib -> this (max
 * commutes with the monotonic quantiser of type_trans.cc:183-184, so y == calib(pool2d_f32(x_f32))). */
plhip_status plhip_pool2d_max_i8(plhip_ctx* ctx, const plhip_pool_desc* d, const int8_t* x, int8_t* y);
/* elementwise_add / fusion_elementwise_add_activation(relu), same-shape operands: replaces ElementwiseAddCompute /
 * ElementwiseAddActivationCompute (lite/kernels/arm/elementwise_compute.cc:85-140) -> elementwise_add{,_relu}<float>
 * (lite/backends/arm/math/elementwise.cc).  out may alias x or y. */
plhip_status plhip_elementwise_add_f32(plhip_ctx* ctx, const float* x, const float* y, float* out, int64_t count,
                                       int relu);

/* ---- introspection used by tests: operand-layout self-check of the MFMA tile on this device.
 * Runs a tiny known-answer GEMM through the MFMA path; returns PLHIP_OK iff bit-exact. ---- */
plhip_status plhip_selftest(plhip_ctx* ctx);

/* ---- diagnostics (no effect on results): switches of the shipped library are set HERE, never through the environment,
 * so that a stray variable cannot change which kernel a benchmark measures.  Keys: the A/B knobs of DESIGN.md 3.6 without
 * their former PLHIP_ prefix ("GEMM_WIDE", "CONV_PATCH", "DW_STAGE", "GEMM_DEBUG", ...), "fused_exp", and "fused_stamps" (1 = the fused
 * depthwise -> pointwise kernel records its in-kernel timeline, read back with plhip_debug_read_fw_stamps:
 * [tile][wave][16] shader-clock stamps).  Returns 0, or -1 for an unknown key. ---- */
int plhip_debug_set(const char* key, int value);
int plhip_debug_read_fw_stamps(void* dst_host, size_t bytes);
int plhip_debug_read_fs_stamps(void* dst_host, size_t bytes); /* the streaming fused kernel: [tile < 2048][wave 4][8] */
int plhip_debug_read_f7_stamps(void* dst_host, size_t bytes); /* the small-plane fused kernel: [block < 1024][wave 8][8] */

#ifdef __cplusplus
}
#endif
#endif /* PLHIP_H_ */
