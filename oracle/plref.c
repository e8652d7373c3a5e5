/*
 * plref.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see plref.h header comment).
 *
 * Plain-C restatement of the reference's INT8 conv/GEMM/FC/calib algorithm.  Not used by,
 * linked into, or loaded from the product path.  Each function cites the reference lines it
 * follows (paths relative to the reference tree root).
 */
#include "plref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lite/tests/math/conv_int8_compute_test.cc:67-88 */
void plref_conv_out_dims(const plref_conv_shape* s, int* oh, int* ow) {
  int keh = s->dil[0] * (s->kh - 1) + 1;
  int kew = s->dil[1] * (s->kw - 1) + 1;
  *oh = (s->h + s->pad[0] + s->pad[1] - keh) / s->stride[0] + 1;
  *ow = (s->w + s->pad[2] + s->pad[3] - kew) / s->stride[1] + 1;
}

/* lite/tests/utils/naive_math_impl.h:393-425 — same loop nest and index math; bias/activation
 * are applied later by the float epilogue (the ARM kernels never add bias in int32). */
void plref_conv2d_i8_acc(const plref_conv_shape* s, const int8_t* x, const int8_t* w, int32_t* acc) {
  int oh, ow;
  plref_conv_out_dims(s, &oh, &ow);
  const int g = s->groups;
  const int ocg = s->cout / g;
  const int icg = s->cin / g;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < s->n; ++n) {
    for (int oc_all = 0; oc_all < s->cout; ++oc_all) {
      const int grp = oc_all / ocg;
      for (int y = 0; y < oh; ++y) {
        for (int xo = 0; xo < ow; ++xo) {
          int32_t sum = 0;
          for (int ic = 0; ic < icg; ++ic) {
            for (int r = 0; r < s->kh; ++r) {
              const int ih = y * s->stride[0] - s->pad[0] + r * s->dil[0];
              if (ih < 0 || ih >= s->h) continue;
              for (int q = 0; q < s->kw; ++q) {
                const int iw = xo * s->stride[1] - s->pad[2] + q * s->dil[1];
                if (iw < 0 || iw >= s->w) continue;
                const int64_t iidx = (((int64_t)n * s->cin + grp * icg + ic) * s->h + ih) * s->w + iw;
                const int64_t widx = (((int64_t)oc_all * icg + ic) * s->kh + r) * s->kw + q;
                sum += (int32_t)x[iidx] * (int32_t)w[widx];
              }
            }
          }
          acc[(((int64_t)n * s->cout + oc_all) * oh + y) * ow + xo] = sum;
        }
      }
    }
  }
}

/* lite/backends/arm/math/conv_impl.cc:103-153 — K x N row-major, K = c*kh*kw + r*kw + q. */
void plref_im2col_i8(const int8_t* x, int cin_g, int h, int w, int kh, int kw, const int pad[4],
                     const int stride[2], const int dil[2], int oh, int ow, int8_t* col) {
  const int n = oh * ow;
  for (int c = 0; c < cin_g; ++c) {
    for (int r = 0; r < kh; ++r) {
      for (int q = 0; q < kw; ++q) {
        int8_t* dst = col + (int64_t)((c * kh + r) * kw + q) * n;
        for (int y = 0; y < oh; ++y) {
          const int ih = y * stride[0] - pad[0] + r * dil[0];
          for (int xo = 0; xo < ow; ++xo) {
            const int iw = xo * stride[1] - pad[2] + q * dil[1];
            int8_t v = 0;
            if (ih >= 0 && ih < h && iw >= 0 && iw < w) v = x[((int64_t)c * h + ih) * w + iw];
            dst[y * ow + xo] = v;
          }
        }
      }
    }
  }
}

/* lite/tests/utils/naive_math_impl.h:246-295 with trans_a = trans_b = false, alpha 1, beta 0. */
void plref_gemm_i8_acc(int m, int n, int k, const int8_t* a, const int8_t* b, int32_t* c) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; ++i) {
    int32_t* crow = c + (int64_t)i * n;
    memset(crow, 0, sizeof(int32_t) * (size_t)n);
    for (int l = 0; l < k; ++l) {
      const int32_t av = a[(int64_t)i * k + l];
      const int8_t* brow = b + (int64_t)l * n;
      for (int j = 0; j < n; ++j) crow[j] += av * (int32_t)brow[j];
    }
  }
}

/* lite/backends/arm/math/conv_impl.cc:490-598 (conv_im2col_gemm_int8) and :260-331
 * (conv1x1s1_gemm_int8): serial over batch and groups, im2col into a workspace unless the conv is
 * 1x1 s1 p0 (conv_gemmlike.cc:125), then one GEMM M = cout/g, K = cin/g*kh*kw, N = oh*ow. */
void plref_conv2d_i8_acc_im2col_gemm(const plref_conv_shape* s, const int8_t* x, const int8_t* w,
                                     int32_t* acc, int8_t* workspace) {
  int oh, ow;
  plref_conv_out_dims(s, &oh, &ow);
  const int g = s->groups;
  const int m = s->cout / g;
  const int icg = s->cin / g;
  const int k = icg * s->kh * s->kw;
  const int n = oh * ow;
  const int is_1x1 = s->kh == 1 && s->kw == 1 && s->stride[0] == 1 && s->stride[1] == 1 &&
                     s->pad[0] == 0 && s->pad[1] == 0 && s->pad[2] == 0 && s->pad[3] == 0;
  for (int b = 0; b < s->n; ++b) {
    for (int grp = 0; grp < g; ++grp) {
      const int8_t* xin = x + ((int64_t)b * s->cin + grp * icg) * s->h * s->w;
      const int8_t* bmat = xin;
      if (!is_1x1) {
        plref_im2col_i8(xin, icg, s->h, s->w, s->kh, s->kw, s->pad, s->stride, s->dil, oh, ow,
                        workspace);
        bmat = workspace;
      }
      plref_gemm_i8_acc(m, n, k, w + (int64_t)grp * m * k, bmat,
                        acc + ((int64_t)b * s->cout + grp * m) * n);
    }
  }
}

/* conv_gemmlike.cc:208-263, conv_depthwise.cc:146-158,242-271 — fp32 arithmetic, evaluated
 * left-to-right exactly as written there (ws * in / out; bias / out; coef / out). */
float plref_fold_scales(int int8_out, float in_scale, const float* w_scale, int n_wscale,
                        float out_scale, const float* bias, int cout, int act, float alpha,
                        float* scale_out, float* bias_out) {
  for (int i = 0; i < cout; ++i) {
    volatile float ws = w_scale[n_wscale == 1 ? 0 : i];
    volatile float t = ws * in_scale;
    if (int8_out) t = t / out_scale;
    scale_out[i] = t;
    float b = bias ? bias[i] : 0.f;
    if (int8_out && bias) {
      volatile float bb = b / out_scale;
      b = bb;
    }
    bias_out[i] = b;
  }
  if (int8_out && act == PLREF_ACT_RELU6) {
    volatile float a = alpha / out_scale;
    return a;
  }
  return alpha;
}

/* saturate_cast<int8_t>(roundf(v)) then floor at -127: lite/backends/arm/math/saturate.h,
 * conv_block_utils.h:3203-3225, type_trans.cc:183-184. */
int8_t plref_round_sat_i8(float v) {
  float r = roundf(v); /* ties away from zero (fcvtas) */
  if (r > 127.f) r = 127.f;
  if (r < -127.f) r = -127.f; /* sat to -128 then max(-127) == clamp at -127 */
  if (r != r) r = 0.f;
  return (int8_t)(int)r;
}

/* conv_block_utils.h:3188-3201; the multiply-add is a single fused fmla in the GEMM epilogue
 * (gemm_prepacked_int8.cc:655-694), so one rounding. */
float plref_epilogue_f32(int32_t acc, float scale, float bias, int act, float alpha) {
  float y = fmaf((float)acc, scale, bias);
  if (act == PLREF_ACT_RELU) {
    y = y > 0.f ? y : 0.f;
  } else if (act == PLREF_ACT_RELU6) {
    y = y > 0.f ? y : 0.f;
    y = y < alpha ? y : alpha;
  } else if (act == PLREF_ACT_LEAKY) {
    y = y > 0.f ? y : alpha * y;
  }
  return y;
}

int8_t plref_epilogue_i8(int32_t acc, float scale, float bias, int act, float alpha) {
  return plref_round_sat_i8(plref_epilogue_f32(acc, scale, bias, act, alpha));
}

void plref_apply_epilogue_f32(const int32_t* acc, int n, int cout, int spatial, const float* scale,
                              const float* bias, int act, float alpha, float* y) {
#pragma omp parallel for schedule(static)
  for (int64_t nc = 0; nc < (int64_t)n * cout; ++nc) {
    const int c = (int)(nc % cout);
    for (int i = 0; i < spatial; ++i)
      y[nc * spatial + i] = plref_epilogue_f32(acc[nc * spatial + i], scale[c], bias[c], act, alpha);
  }
}

void plref_apply_epilogue_i8(const int32_t* acc, int n, int cout, int spatial, const float* scale,
                             const float* bias, int act, float alpha, int8_t* y) {
#pragma omp parallel for schedule(static)
  for (int64_t nc = 0; nc < (int64_t)n * cout; ++nc) {
    const int c = (int)(nc % cout);
    for (int i = 0; i < spatial; ++i)
      y[nc * spatial + i] = plref_epilogue_i8(acc[nc * spatial + i], scale[c], bias[c], act, alpha);
  }
}

/* fc_compute.cc:84-109 (m = prod(x.dims[:ncol]), k, n = w.dims[1]); w is [k, n]. */
void plref_fc_i8_acc(int m, int n, int k, const int8_t* x, const int8_t* w, int32_t* acc) {
  plref_gemm_i8_acc(m, n, k, x, w, acc);
}

void plref_fc_epilogue_f32(const int32_t* acc, int m, int n, const float* scale, const float* bias,
                           int relu, float* y) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j)
      y[(int64_t)i * n + j] = plref_epilogue_f32(acc[(int64_t)i * n + j], scale[j],
                                                 bias ? bias[j] : 0.f, relu ? PLREF_ACT_RELU : 0, 0.f);
}

/* fc_compute.cc:250-266 + funcs.cc:24-108: product rounded, bias added (second rounding), relu. */
void plref_fc_epilogue_f32_two_roundings(const int32_t* acc, int m, int n, const float* scale,
                                         const float* bias, int relu, float* y) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      volatile float p = (float)acc[(int64_t)i * n + j] * scale[j];
      volatile float v = bias ? p + bias[j] : p;
      float r = v;
      if (relu) r = r > 0.f ? r : 0.f;
      y[(int64_t)i * n + j] = r;
    }
}

void plref_fc_epilogue_i8(const int32_t* acc, int m, int n, const float* scale, const float* bias,
                          int relu, int8_t* y) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j)
      y[(int64_t)i * n + j] = plref_epilogue_i8(acc[(int64_t)i * n + j], scale[j],
                                                bias ? bias[j] : 0.f, relu ? PLREF_ACT_RELU : 0, 0.f);
}

/* type_trans.cc:45 (inv_scale = 1.f / scale), :183-184 (roundf(inv_scale * x), saturate, >= -127). */
void plref_calib_f32_to_i8(const float* x, int8_t* y, float scale, int64_t count) {
  volatile float inv = 1.f / scale;
  const float inv_scale = inv;
  for (int64_t i = 0; i < count; ++i) {
    volatile float p = inv_scale * x[i];
    y[i] = plref_round_sat_i8(p);
  }
}

/* type_trans.cc:268-371: out = in_scale * in. */
void plref_calib_i8_to_f32(const int8_t* x, float* y, float scale, int64_t count) {
  for (int64_t i = 0; i < count; ++i) {
    volatile float p = scale * (float)x[i];
    y[i] = p;
  }
}

/* pooling.cc:1006- (pooling_global_avg): sum over the plane, divide by its size. */
void plref_global_avg_pool_f32(const float* x, int nc, int spatial, float* y) {
  for (int i = 0; i < nc; ++i) {
    double s = 0.0;
    for (int j = 0; j < spatial; ++j) s += x[(int64_t)i * spatial + j];
    y[i] = (float)(s / spatial);
  }
}

/* softmax.cc (softmax_inner1): max-subtracted exp, normalised by the row sum. */
void plref_softmax_f32(const float* x, int rows, int cols, float* y) {
  for (int i = 0; i < rows; ++i) {
    const float* xr = x + (int64_t)i * cols;
    float* yr = y + (int64_t)i * cols;
    float mx = xr[0];
    for (int j = 1; j < cols; ++j) mx = xr[j] > mx ? xr[j] : mx;
    double s = 0.0;
    for (int j = 0; j < cols; ++j) {
      yr[j] = expf(xr[j] - mx);
      s += yr[j];
    }
    for (int j = 0; j < cols; ++j) yr[j] = (float)(yr[j] / s);
  }
}

/* pool_op.cc:44-61 */
int plref_pool_out_size(int in, int k, int pad0, int pad1, int stride, int ceil_mode) {
  if (!ceil_mode) return (in - k + pad0 + pad1) / stride + 1;
  return (in - k + pad0 + pad1 + stride - 1) / stride + 1;
}

/* pooling.cc:38-215 (non-global, non-adaptive branch), loop for loop. */
void plref_pool2d_f32(const float* x, int planes, int h, int w, int oh, int ow, int kh, int kw, int sh_, int sw_,
                      const int pad[4], int is_max, int exclusive, float* y) {
  const int pad_h = pad[0], pad_w = pad[2];
#pragma omp parallel for schedule(static)
  for (int c = 0; c < planes; ++c) {
    const float* xin = x + (int64_t)c * h * w;
    float* yo = y + (int64_t)c * oh * ow;
    for (int ih = 0; ih < oh; ++ih) {
      int sh = ih * sh_, eh = sh + kh;
      sh = (sh - pad_h) < 0 ? 0 : sh - pad_h;
      eh = (eh - pad_h) > h ? h : eh - pad_h;
      for (int iw = 0; iw < ow; ++iw) {
        int sw = iw * sw_, ew = sw + kw;
        sw = (sw - pad_w) < 0 ? 0 : sw - pad_w;
        ew = (ew - pad_w) > w ? w : ew - pad_w;
        float result = 0.f;
        for (int a = sh; a < eh; ++a)
          for (int b = sw; b < ew; ++b) {
            const float v = xin[a * w + b];
            if (a == sh && b == sw) result = v;
            else if (is_max) result = result >= v ? result : v;
            else { volatile float t = result + v; result = t; }
          }
        if (!is_max) {
          if (exclusive) {
            int div = (ew - sw) * (eh - sh);
            div = div > 0 ? div : 1;
            result /= div;
          } else {
            int bh = kh, bw = kw;
            if (ew == w) {
              bw = (sw + kw) >= (w + pad[3]) ? (w + pad[3]) : (sw + kw);
              bw -= sw;
              if ((sw - pad_w) < 0 && (sw + kw) > (w + pad[3])) bw += pad_w;
            }
            if (eh == h) {
              bh = (sh + kh) >= (h + pad[1]) ? (h + pad[1]) : (sh + kh);
              bh -= sh;
              if ((sh - pad_h) < 0 && (sh + kh) > (h + pad[1])) bh += pad_h;
            }
            result /= bh * bw;
          }
        }
        yo[ih * ow + iw] = result;
      }
    }
  }
}

/* elementwise.cc: elementwise_add<float> / elementwise_add_relu<float>. */
void plref_elementwise_add_f32(const float* x, const float* y, float* out, int64_t count, int relu) {
  for (int64_t i = 0; i < count; ++i) {
    volatile float s = x[i] + y[i];
    float r = s;
    if (relu) r = r > 0.f ? r : 0.f;
    out[i] = r;
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * BASELINE config C1: the reference's x86 fp32 path (CxxPredictor plumbing, no GPU), restated for timing beside the
 * GPU number and for the plumbing check.  Parity UNPINNED beyond 1e-5 relative: the reference's GEMM is MKLML's
 * cblas_sgemm (mklml_lnx_2019.0.1.20181227, absent here), whose summation order is not specified.
 *
 * lite/kernels/x86/conv_compute.h:48-150: per image and group, im2col (kCFO: [c][kh][kw][oh][ow],
 * lite/backends/x86/math/im2col.cc:31-57; skipped for 1x1 stride 1 pad 0: IsExpand :31-46) then
 * out[g] (out_step x OH*OW) = filter[g] (out_step x K) . col (K x OH*OW)   (Blas::MatMul -> cblas_sgemm,
 * lite/backends/x86/math/blas_impl.h:38,459-).  The kernel adds NO bias and applies NO activation: batch_norm and relu
 * are separate fp32 instructions of the x86 program.  depthwise_conv2d is registered on the same class
 * (lite/kernels/x86/conv_compute.cc), i.e. a 1 x 9 GEMM per group. */
void plref_im2col_f32(const float* x, int cin_g, int h, int w, int kh, int kw, int pt, int pl, int sh, int sw, int dh,
                      int dw, int oh, int ow, float* col) {
  const int n = oh * ow;
#pragma omp parallel for schedule(static) if ((int64_t)cin_g * kh * kw * n > 65536)
  for (int r = 0; r < cin_g * kh * kw; ++r) {
    const int q = r % kw, p = (r / kw) % kh, c = r / (kw * kh);
    float* dst = col + (int64_t)r * n;
    for (int y = 0; y < oh; ++y) {
      const int ih = y * sh - pt + p * dh;
      for (int xo = 0; xo < ow; ++xo) {
        const int iw = xo * sw - pl + q * dw;
        dst[y * ow + xo] = (ih >= 0 && ih < h && iw >= 0 && iw < w) ? x[((int64_t)c * h + ih) * w + iw] : 0.f;
      }
    }
  }
}

/* C (m x n) = A (m x k) . B (k x n), row-major, alpha 1, beta 0; k ascending per output element */
void plref_sgemm_f32(int m, int n, int k, const float* a, const float* b, float* c) {
#pragma omp parallel for schedule(static) if ((int64_t)m * n * k > 65536 && m > 1)
  for (int i = 0; i < m; ++i) {
    float* ci = c + (int64_t)i * n;
    for (int j = 0; j < n; ++j) ci[j] = 0.f;
    for (int p = 0; p < k; ++p) {
      const float av = a[(int64_t)i * k + p];
      const float* bp = b + (int64_t)p * n;
      for (int j = 0; j < n; ++j) ci[j] += av * bp[j];
    }
  }
}

/* col: caller-provided scratch of (cin/groups)*kh*kw*oh*ow floats (unused for 1x1 stride 1 pad 0) */
void plref_conv2d_f32_x86(const plref_conv_shape* s, const float* x, const float* w, float* y, float* col) {
  int oh, ow;
  plref_conv_out_dims(s, &oh, &ow);
  const int g = s->groups, in_step = s->cin / g, out_step = s->cout / g;
  const int kk = in_step * s->kh * s->kw, n = oh * ow;
  const int expand = !(s->kh == 1 && s->kw == 1 && s->stride[0] == 1 && s->stride[1] == 1 && s->pad[0] == 0 &&
                       s->pad[2] == 0 && s->dil[0] == 1 && s->dil[1] == 1);
  for (int b = 0; b < s->n; ++b) {
    for (int grp = 0; grp < g; ++grp) {
      const float* in_slice = x + ((int64_t)b * s->cin + (int64_t)grp * in_step) * s->h * s->w;
      const float* cm = in_slice;
      if (expand) {
        /* conv_compute.h:113-121 passes {paddings[0], paddings[2], paddings[0], paddings[2]}: top / left pads for both sides */
        plref_im2col_f32(in_slice, in_step, s->h, s->w, s->kh, s->kw, s->pad[0], s->pad[2], s->stride[0], s->stride[1],
                         s->dil[0], s->dil[1], oh, ow, col);
        cm = col;
      }
      plref_sgemm_f32(out_step, n, kk, w + (int64_t)grp * out_step * kk, cm,
                      y + ((int64_t)b * s->cout + (int64_t)grp * out_step) * n);
    }
  }
}

/* lite/kernels/x86/batch_norm_compute.h:43-161 (is_test: y = (x - mean) / sqrt(var + eps) * scale + bias), with the relu
 * that follows it in MobileNetV1 applied in the same pass when relu != 0 (lite/kernels/x86/activation_compute.h) */
void plref_batch_norm_f32(const float* x, float* y, int n, int c, int spatial, const float* scale, const float* bias,
                          const float* mean, const float* var, float eps, int relu) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n * c; ++i) {
    const int ch = i % c;
    const float inv = 1.f / sqrtf(var[ch] + eps);
    const float a = inv * scale[ch], bb = bias[ch] - mean[ch] * inv * scale[ch];
    const float* xp = x + (int64_t)i * spatial;
    float* yp = y + (int64_t)i * spatial;
    for (int j = 0; j < spatial; ++j) {
      float v = xp[j] * a + bb;
      yp[j] = relu ? (v > 0.f ? v : 0.f) : v;
    }
  }
}
