// ref_shim.cc — builds the REFERENCE's own scalar oracle into oracle/_ref/libref_naive.so.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the reference
// header lite/tests/utils/naive_math_impl.h *in place* from /root/reference (never copied into
// this repository) and exports thin extern "C" entry points around its templates, so that
//   * oracle/plref.c (our restatement) can be validated against the reference itself, and
//   * tests/golden/make_golden.py can mint golden vectors from the reference.
// The header is framework-free but forgets a few libc includes, supplied here first.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

#include "lite/tests/utils/naive_math_impl.h"  // -I/root/reference

extern "C" {

// conv_basic<int8_t,int>  lite/tests/utils/naive_math_impl.h:351-453
// (bias is int and added in the accumulator type there; we pass flag_bias=false, act=0 to get
//  the raw accumulator, or act>0 to exercise its integer activation).
void ref_conv_basic_i8(const int8_t* din, int* dout, int num, int chout, int hout, int wout,
                       int chin, int hin, int win, const int8_t* weights, const int* bias,
                       int group, int kernel_w, int kernel_h, int stride_w, int stride_h,
                       int dila_w, int dila_h, int pad_w, int pad_h, int flag_bias, int act_type) {
  conv_basic<int8_t, int>(din, dout, num, chout, hout, wout, chin, hin, win, weights, bias, group,
                          kernel_w, kernel_h, stride_w, stride_h, dila_w, dila_h, pad_w, pad_h,
                          flag_bias != 0, act_type);
}

// conv_basic<float,float> — the float baseline the reference's int8 tests compare against
// (lite/tests/math/conv_int8_compute_test.cc:298-321).
void ref_conv_basic_f32(const float* din, float* dout, int num, int chout, int hout, int wout,
                        int chin, int hin, int win, const float* weights, const float* bias,
                        int group, int kernel_w, int kernel_h, int stride_w, int stride_h,
                        int dila_w, int dila_h, int pad_w, int pad_h, int flag_bias, int act_type,
                        float six, float scale) {
  conv_basic<float, float>(din, dout, num, chout, hout, wout, chin, hin, win, weights, bias, group,
                           kernel_w, kernel_h, stride_w, stride_h, dila_w, dila_h, pad_w, pad_h,
                           flag_bias != 0, act_type, six, scale);
}

// basic_gemm<int8_t,int>  lite/tests/utils/naive_math_impl.h:246-295
void ref_basic_gemm_i8(int trans_a, int trans_b, int m, int n, int k, const int8_t* a, int lda,
                       const int8_t* b, int ldb, int* c, int ldc, const int* bias, int flag_bias,
                       int flag_relu) {
  basic_gemm<int8_t, int>(trans_a != 0, trans_b != 0, m, n, k, 1, a, lda, b, ldb, 0, c, ldc, bias,
                          flag_bias != 0, flag_relu != 0);
}

// basic_gemv<int8_t,int>  lite/tests/utils/naive_math_impl.h:298-343
void ref_basic_gemv_i8(int m, int k, const int8_t* a, const int8_t* b, const int* bias, int* c,
                       int trans_a, int flag_bias, int flag_act) {
  basic_gemv<int8_t, int>(m, k, a, b, bias, c, 1, 0, trans_a != 0, flag_bias != 0, flag_act);
}

}  // extern "C"
