/*
 * plref.h — CPU oracle for the INT8 conv / depthwise / fc / calib hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (chenjiaoAngel/Paddle-Lite, ARM int8 path).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product path (paddle-lite_amd/) never links or loads it.
 *
 * Parity pin: the accumulator / GEMM functions here are checked bit-for-bit
 * against the reference's own scalar oracle lite/tests/utils/naive_math_impl.h
 * compiled in place (oracle/_ref, see oracle/Makefile) and against the golden
 * vectors minted from it (tests/golden/).  The float epilogue follows the
 * scalar spec lite/backends/arm/math/conv_block_utils.h:3185-3225.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef PLREF_H_
#define PLREF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Activation codes = lite_api::ActivationType (lite/api/paddle_place.h:101-105). */
enum { PLREF_ACT_NONE = 0, PLREF_ACT_RELU = 1, PLREF_ACT_RELU6 = 2, PLREF_ACT_LEAKY = 4 };

typedef struct {
  int n, cin, h, w;      /* input NCHW */
  int cout, kh, kw;      /* filter OIHW, I = cin/groups */
  int pad[4];            /* {top, bottom, left, right}  (op_params.h:446, conv_op.h:149-161) */
  int stride[2];         /* {h, w} */
  int dil[2];            /* {h, w} */
  int groups;
} plref_conv_shape;

/* Output dims: lite/tests/math/conv_int8_compute_test.cc:67-88. */
void plref_conv_out_dims(const plref_conv_shape* s, int* oh, int* ow);

/* Direct int32 accumulator, zero padding by skipping OOB taps.
 * lite/tests/utils/naive_math_impl.h:393-425 (conv_basic<int8_t,int>, no bias/act). */
void plref_conv2d_i8_acc(const plref_conv_shape* s, const int8_t* x, const int8_t* w, int32_t* acc);

/* im2col for one (image, group) slab: col[(c*kh*kw + r*kw + q) * (oh*ow) + (y*ow + x)].
 * lite/backends/arm/math/conv_impl.cc:103-153. */
void plref_im2col_i8(const int8_t* x, int cin_g, int h, int w, int kh, int kw,
                     const int pad[4], const int stride[2], const int dil[2],
                     int oh, int ow, int8_t* col);

/* C[m,n] = sum_k A[m,k] * B[k,n]  (row-major, no transposes), int32.
 * lite/tests/utils/naive_math_impl.h:246-295 (basic_gemm<int8_t,int>, alpha=1 beta=0). */
void plref_gemm_i8_acc(int m, int n, int k, const int8_t* a, const int8_t* b, int32_t* c);

/* Same accumulator computed the way the reference kernel is structured:
 * for b, g: im2col -> GEMM (conv_impl.cc:490-598), OpenMP over the GEMM rows.
 * Used as the timed CPU baseline ("port"); result identical to plref_conv2d_i8_acc. */
void plref_conv2d_i8_acc_im2col_gemm(const plref_conv_shape* s, const int8_t* x, const int8_t* w,
                                     int32_t* acc, int8_t* workspace /* k*n bytes per thread-0 slab */);

/* Scale / bias folding (per output channel).
 * fp32-out: s_i = w_scale[i]*in_scale, b_i = bias[i]          conv_gemmlike.cc:208-226
 * int8-out: s_i = w_scale[i]*in_scale/out_scale, b_i = bias[i]/out_scale,
 *           relu6 alpha /= out_scale                            conv_gemmlike.cc:229-263,
 *                                                               conv_depthwise.cc:242-271
 * n_wscale is 1 (broadcast) or cout.  bias may be NULL (-> zeros).  Returns folded alpha. */
float plref_fold_scales(int int8_out, float in_scale, const float* w_scale, int n_wscale,
                        float out_scale, const float* bias, int cout, int act, float alpha,
                        float* scale_out, float* bias_out);

/* Epilogue, scalar spec conv_block_utils.h:3185-3225 (cvt_kernel<float>/<int8_t>), vector twin
 * gemm_prepacked_int8.cc:643-796.  y = fma(float(acc), scale, bias) then act; int8: round half
 * away from zero, saturate, floor at -127. */
float  plref_epilogue_f32(int32_t acc, float scale, float bias, int act, float alpha);
int8_t plref_epilogue_i8(int32_t acc, float scale, float bias, int act, float alpha);

/* Apply the epilogue to an NCHW accumulator tensor: channel stride = spatial. */
void plref_apply_epilogue_f32(const int32_t* acc, int n, int cout, int spatial, const float* scale,
                              const float* bias, int act, float alpha, float* y);
void plref_apply_epilogue_i8(const int32_t* acc, int n, int cout, int spatial, const float* scale,
                             const float* bias, int act, float alpha, int8_t* y);

/* FC: acc[m,n] = sum_k x[m,k] * w[k,n]  (w in Paddle "mul" layout [k,n]); per-column (n) scale.
 * lite/kernels/arm/fc_compute.cc:84-109,229-344.  Epilogue spec = SURVEY A.8 (single fma, relu). */
void plref_fc_i8_acc(int m, int n, int k, const int8_t* x, const int8_t* w, int32_t* acc);
void plref_fc_epilogue_f32(const int32_t* acc, int m, int n, const float* scale, const float* bias,
                           int relu, float* y);
/* The reference's fp32-out FC has two epilogue routes (fc_compute.cc:66-71, 229-290):
 *   route 0  gemv_int8 per row (m == 1 or a per-column weight scale): vmlaq_f32(bias, float(acc), scale)
 *            (gemv_arm_int8.cc:47-56), contracted to one fmla == plref_fc_epilogue_f32;
 *   route 1  gemm_s8 without bias, then fill_bias_fc (m > 1 and a single weight scale): y = float(acc)*s rounded,
 *            then y + b rounded again, then relu (fc_compute.cc:250-266, lite/backends/arm/math/funcs.cc:24-108;
 *            also the scalar tail of write_gemv_out, gemv_arm_int8.cc:85-87).
 * Route 1 differs from route 0 by at most half an ulp of the product plus one ulp of the result. */
void plref_fc_epilogue_f32_two_roundings(const int32_t* acc, int m, int n, const float* scale,
                                         const float* bias, int relu, float* y);
void plref_fc_epilogue_i8(const int32_t* acc, int m, int n, const float* scale, const float* bias,
                          int relu, int8_t* y);

/* calib: lite/backends/arm/math/type_trans.cc:34-47,180-185 (fp32->int8), :268-371 (int8->fp32). */
void plref_calib_f32_to_i8(const float* x, int8_t* y, float scale, int64_t count);
void plref_calib_i8_to_f32(const int8_t* x, float* y, float scale, int64_t count);

/* Glue ops of the MobileNet graph (fp32): global average pool and softmax over the last axis.
 * lite/backends/arm/math/pooling.cc (pooling_global_avg), lite/backends/arm/math/softmax.cc. */
void plref_global_avg_pool_f32(const float* x, int nc, int spatial, float* y);
void plref_softmax_f32(const float* x, int rows, int cols, float* y);

/* pool2d, fp32: lite/backends/arm/math/pooling.cc:38-215 (pooling_basic: windows clipped to the image, the first
 * element initialises the result; avg: exclusive -> clipped window size, else the divisor of :165-205 as written).
 * pad = {top, bottom, left, right}.  Output dims are the caller's (PoolOutputSize, lite/operators/pool_op.cc:44-61). */
void plref_pool2d_f32(const float* x, int planes, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw,
                      const int pad[4], int is_max, int exclusive, float* y);
int plref_pool_out_size(int in, int k, int pad0, int pad1, int stride, int ceil_mode);

/* elementwise_add / elementwise_add_relu <float>, same-shape operands: lite/backends/arm/math/elementwise.cc
 * (vaddq_f32, then vmaxq_f32 with 0 for the fused relu). */
void plref_elementwise_add_f32(const float* x, const float* y, float* out, int64_t count, int relu);

/* round-half-away + saturate helpers exposed for host-side bit tricks tests. */
int8_t plref_round_sat_i8(float v);


/* BASELINE config C1: the reference's x86 fp32 conv path (lite/kernels/x86/conv_compute.h:48-150: im2col + SGEMM, no
 * bias / activation in the kernel), batch_norm (+relu).  Timing / plumbing restatement; parity unpinned beyond 1e-5. */
void plref_im2col_f32(const float* x, int cin_g, int h, int w, int kh, int kw, int pt, int pl, int sh, int sw, int dh,
                      int dw, int oh, int ow, float* col);
void plref_sgemm_f32(int m, int n, int k, const float* a, const float* b, float* c);
void plref_conv2d_f32_x86(const plref_conv_shape* s, const float* x, const float* w, float* y, float* col);
void plref_batch_norm_f32(const float* x, float* y, int n, int c, int spatial, const float* scale, const float* bias,
                          const float* mean, const float* var, float eps, int relu);

#ifdef __cplusplus
}
#endif
#endif /* PLREF_H_ */
