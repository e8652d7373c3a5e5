// plhip_device.h — device-side helpers shared by the gfx950 INT8 kernels.
//
// Numerics contract (reference: lite/backends/arm/math/conv_block_utils.h:3185-3225
// scalar spec, gemm_prepacked_int8.cc:643-796 vector spec):
//   y  = fma(float(acc), scale[c], bias[c])      one rounding (fmla)
//   act: relu max(y,0) | relu6 min(max(y,0),alpha) | leaky y>0 ? y : alpha*y
//   int8: round half away from zero, clamp to [-127, 127]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plhip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { OUT_I32 = 0, OUT_F32 = 1, OUT_I8 = 2, OUT_GAP = 3 };  // OUT_GAP: fp32, averaged over the plane (fused_dwpw_small.hip only)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_LEAKY = 4 };

__device__ __forceinline__ float epilogue_f32(int acc, float scale, float bias, int act, float alpha) {
  float y = __fmaf_rn((float)acc, scale, bias);
  if (act == ACT_RELU) {
    y = y > 0.f ? y : 0.f;
  } else if (act == ACT_RELU6) {
    y = y > 0.f ? y : 0.f;
    y = y < alpha ? y : alpha;
  } else if (act == ACT_LEAKY) {
    y = y > 0.f ? y : alpha * y;
  }
  return y;
}

// round-half-away-from-zero(clamp(y, -127, 127)) without a transcendental-style sequence:
// 2y is exact in fp32, truncation toward zero of 2y is exact, and
//   round_half_away(y) = (t + 1 + (t >> 31)) >> 1   with t = trunc(2y)
// (t >= 0: (t+1)>>1 ; t < 0: t>>1 = floor(t/2)).  Checked against roundf() in tests/test_host_logic.py.
__device__ __forceinline__ int round_sat_i8(float y) {
  float y2 = __builtin_amdgcn_fmed3f(y + y, -254.f, 254.f);
  int t = (int)y2;  // v_cvt_i32_f32: toward zero; NaN -> 0
  return (t + 1 + (t >> 31)) >> 1;
}

// q = (t + 1) >> 1 on four packed bytes t in 0..254 (the relu / relu6 requantisation on doubled values): v_lerp_u8 computes
// (a + b + (c & 1)) >> 1 per byte without carries between bytes: ONE instruction for add, shift and mask (semantics
// checked on the device, tools/probe_cvt_pk_u8.hip; every VALU instruction costs the SIMD 4 cycles, and the
// requantisation is most of what these kernels issue).  v_cvt_pk_u8_f32 cannot replace the float -> byte conversion:
// it rounds to nearest EVEN (0.5 -> 0, 2.5 -> 2), the reference rounds half away from zero.
__device__ __forceinline__ uint32_t round_half_up4_u8(uint32_t packed_doubled) {
  return __builtin_amdgcn_lerp(packed_doubled, 0u, 0x01010101u);
}

// relu / relu6 requantisation, last step: four DOUBLED values (each <= hi2 <= 254 after the v_min_f32 here; negative ones are
// the relu's zeros) -> four int8 in one dword.  The float -> byte conversion is v_cvt_pk_u8_f32 under round-toward-zero:
// truncation, saturation to 0..255 and the byte insert in ONE instruction per value instead of v_cvt_u32_f32 + shift / or
// (tools/probe_cvt_rtz.hip on the device: == trunc(sat(x, 0, 255)) on 2^20 values incl. every tie and both neighbours of every
// integer; a v_fma_f32 issued behind the restore rounds to nearest again, one issued INSIDE the window does not: the mode
// switch and the four conversions are therefore ONE asm statement that nothing else can enter).  Per value: v_cvt_f32_i32,
// v_fma_f32, v_min_f32, v_cvt_pk_u8_f32 + 1/4 v_lerp_u8 = 4.25 VALU (was 5).  MODE.fp_round (bits 1:0) is 0 = nearest-even
// in every kernel of this library and is restored to it.  Bit-identical to trunc(med3(y, 0, hi2)) -> (t + 1) >> 1.
__device__ __forceinline__ uint32_t pack4_nn_rtz(float y0, float y1, float y2, float y3, float hi2, uint32_t ones = 0x01010101u) {
  y0 = __builtin_fminf(y0, hi2);
  y1 = __builtin_fminf(y1, hi2);
  y2 = __builtin_fminf(y2, hi2);
  y3 = __builtin_fminf(y3, hi2);
  uint32_t p;
  asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %0, %2, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 2, %0\n\t"
      "v_cvt_pk_u8_f32 %0, %4, 3, %0\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(p)
      : "v"(y0), "v"(y1), "v"(y2), "v"(y3));
  return __builtin_amdgcn_lerp(p, 0u, ones);  // (t + 1) >> 1 per byte
}

__device__ __forceinline__ uint32_t pack4_i8(int q0, int q1, int q2, int q3) {
  // low byte of each int32 -> one dword, little endian
  uint32_t lo = __builtin_amdgcn_perm((uint32_t)q1, (uint32_t)q0, 0x0c0c0400u);  // [q0.b0, q1.b0, 0, 0]
  uint32_t hi = __builtin_amdgcn_perm((uint32_t)q3, (uint32_t)q2, 0x04000c0cu);  // [0, 0, q2.b0, q3.b0]
  return lo | hi;
}

// 4x4 byte transpose: in r0..r3 (row j holds bytes for columns 0..3), out o[i] holds column i's
// bytes for rows 0..3 (row 0 in the low byte).  8 v_perm_b32.
__device__ __forceinline__ void transpose4x4_b8(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3,
                                                uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
  // __builtin_amdgcn_perm(hi, lo, sel): selector byte 0-3 picks lo.b0-3, 4-7 picks hi.b0-3
  uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);  // r0.b0 r1.b0 r0.b1 r1.b1
  uint32_t t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);  // r0.b2 r1.b2 r0.b3 r1.b3
  uint32_t t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
  uint32_t t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  o0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);  // r0.b0 r1.b0 r2.b0 r3.b0
  o1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  o2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  o3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}


// Kernel arguments live in memory (the kernarg segment); the compiler loads each field where it is first used, and a
// branch between two uses turns that into SERIAL scalar-load round trips (timeline stamps: three of them, ~2300 cycles
// before the first vector load of the GEMM kernel).  Naming a field here, at the top of a kernel, makes it live in
// the entry block, so all fields arrive with one batch of s_load.
#define PLHIP_PRELOAD(x) asm volatile("" ::"s"(x))

}  // namespace plhip
