// dw_common.h — device helpers shared by the depthwise kernels (depthwise_i8.hip) and the fused depthwise -> pointwise
// kernel (fused_dwpw_i8.hip): the unaligned row-window fetch and the 4-wide int8 requantisation.
#pragma once
#include "plhip_device.h"

namespace plhip {

// n / d for n < 2^31 with the host-prepared (magic, shift) pair of DwArgs (magic == 0: d is a power of two)
__device__ __forceinline__ uint32_t fastdiv_u31(uint32_t n, uint32_t magic, int sh) {
  return magic ? (__umulhi(n, magic) >> sh) : (n >> sh);
}

// ones: 0x01010101 (the byte-wise +1 of the packed rounding) — a parameter so that a caller can hand over a scalar kernel
// argument instead of a constant the compiler moves into a VGPR again per use
template <int ACT>
__device__ __forceinline__ uint32_t dw_requant4(const int (&a)[4], float s2, float b2, float alpha, float lo2, float hi2,
                                                uint32_t ones = 0x01010101u) {
  if (ACT == ACT_RELU || ACT == ACT_RELU6)  // (lo2 == 0: the conversion saturates negative values to it)
    return pack4_nn_rtz(__fmaf_rn((float)a[0], s2, b2), __fmaf_rn((float)a[1], s2, b2), __fmaf_rn((float)a[2], s2, b2),
                        __fmaf_rn((float)a[3], s2, b2), hi2, ones);
  int q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float y2 = __fmaf_rn((float)a[j], s2, b2);
    if (ACT == ACT_LEAKY) y2 = y2 > 0.f ? y2 : alpha * y2;
    const int t = (int)__builtin_amdgcn_fmed3f(y2, lo2, hi2);
    q[j] = (t + 1 + (t >> 31)) >> 1;
  }
  return pack4_i8(q[0], q[1], q[2], q[3]);
}

// relu / relu6 requantisation of 4 accumulators on doubled values (pack4_nn_rtz, plhip_device.h)
__device__ __forceinline__ uint32_t requant4_nn_rtz(const int (&a)[4], float s2, float b2, float hi2, uint32_t ones = 0x01010101u) {
  return pack4_nn_rtz(__fmaf_rn((float)a[0], s2, b2), __fmaf_rn((float)a[1], s2, b2), __fmaf_rn((float)a[2], s2, b2),
                      __fmaf_rn((float)a[3], s2, b2), hi2, ones);
}

// Byte-validity masks of a row window of ND dwords whose byte 0 is input column `start` (may be negative: left padding):
// byte i is kept iff 0 <= start + i < w.  Built with two 64-bit shifts instead of a compare / select per byte (the
// depthwise kernels are VALU-bound; the per-byte form cost ~30 VALU per dword).  Needs -4 < start and start < w.
template <int ND>
__device__ __forceinline__ void dw_col_masks(int start, int w, uint32_t (&cmask)[ND]) {
  const int lo = start < 0 ? -start : 0;                     // invalid bytes at the low end (<= 3)
  const int hi = w - start < 4 * ND ? w - start : 4 * ND;    // first invalid byte at the high end (>= 1, > lo)
  const int hi8 = hi < 8 ? hi : 8;
  const unsigned long long m01 = (~0ull >> (64 - 8 * (hi8 - lo))) << (8 * lo);
  cmask[0] = (uint32_t)m01;
  if (ND > 1) cmask[1] = (uint32_t)(m01 >> 32);
  if (ND > 2) {
    const int n2 = hi - 8 < 0 ? 0 : hi - 8;                  // valid bytes of dword 2 (0..4)
    cmask[2] = (uint32_t)((1ull << (8 * n2)) - 1ull);
  }
}

// first tap of an accumulator: VOP3P form with the constant 0 as addend (the builtin selects v_dot4c, which needs the
// accumulator zeroed by a separate v_mov first)
__device__ __forceinline__ int sdot4_first(uint32_t a, uint32_t b) {
  int r;
  asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Row fetch, branch free: the row index is clamped (an out-of-image row only zeroes the masks), the column start is
// clamped for the left border (the missing bytes are shifted in as zeros), and — only in the TAIL instantiation, which
// the last workgroup alone runs — the address is pulled back so that the load never crosses the end of the tensor.
template <int ND, bool TAIL>
__device__ __forceinline__ void dw_load_row(const int8_t* __restrict__ xplane, int ih, int h, int w, int lcol, int sh,
                                            long plane_room, const uint32_t (&cmask)[ND], uint32_t (&d)[ND]) {
  const bool rv = ih >= 0 && ih < h;
  const int ihc = ih < 0 ? 0 : (ih >= h ? h - 1 : ih);
  int off = ihc * w + lcol;  // byte offset inside the plane
  int back = 0;
  if (TAIL) {
    const long lim = plane_room - 4 * ND;  // last offset from which 4*ND bytes are still inside the tensor
    if (off > lim) {
      back = off - (int)(lim < 0 ? 0 : lim);
      off -= back;
    }
  }
  __builtin_memcpy(d, xplane + off, 4 * ND);
  if (TAIL && back) {  // bytes were fetched `back` too early: shift them down, zeros come in from the top
    const int s8 = 8 * (back & 3), dw = back >> 2;
    uint32_t t[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const uint32_t lo = (i + dw < ND) ? d[(i + dw < ND) ? i + dw : 0] : 0u;
      const uint32_t hi = (i + dw + 1 < ND) ? d[(i + dw + 1 < ND) ? i + dw + 1 : 0] : 0u;
      t[i] = s8 ? ((lo >> s8) | (hi << (32 - s8))) : lo;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = t[i];
  }
  if (sh) {  // left border lane: make room for the zero padding bytes
    const int s8 = 8 * sh;
    if (ND == 3) d[ND - 1] = (d[ND - 1] << s8) | (d[ND - 2] >> (32 - s8));
    d[1] = (d[1] << s8) | (d[0] >> (32 - s8));
    d[0] = d[0] << s8;
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) d[i] = rv ? (d[i] & cmask[i]) : 0u;
}

}  // namespace plhip
