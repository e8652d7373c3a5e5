// fused_dwpw_small.hip — depthwise 3x3 [int8_out] + pointwise 1x1 in ONE launch for the 7 x 7 planes at the end of the
// MobileNet programs: 512 -> 1024 (depthwise stride 2 from 14 x 14) and 1024 -> 1024 (stride 1).  As two kernels each of the four
// launches is a few microseconds of work behind a launch's fixed cost (ramp, first fetch, drain: 9-13 us per depthwise launch for
// 6-13 MB); here the int8 tensor between them never leaves the CU and two launches disappear.
// Replaces the instruction pair DepthwiseConv<kInt8,kInt8>::Run (lite/kernels/arm/conv_depthwise.cc:407-446 ->
// conv3x3s{1,2}_depthwise_int8.cc) ; GemmLikeConv<kInt8,*>::Run (lite/kernels/arm/conv_gemmlike.cc:399-462 ->
// gemm_prepacked_int8.cc:2582-2744); results bit-identical to the two kernels.
//
// Structure (the streaming kernel's, fused_dwpw_stream.hip, re-cut for a plane of 49 pixels):
//   * tile = ONE image = 7 rows x 8 slots (7 pixels + 1 junk) = 56 of 64 slots = 2 MFMA n tiles; block = 8 waves; the M output
//     channels are split over MB blocks per image (each produces the image again: the depthwise work of a 7 x 7 plane is small,
//     the matrix work is what a block has to share), grid = images x MB, XCD-contiguous (the MB blocks of an image on one L2);
//   * produce: lane = (channel, output row): its three input rows arrive as ONE 8-byte (stride 1) or 16-byte (stride 2) fetch
//     each, starting one / two bytes in front of the row so that no fetch crosses the end of the tensor; the 7 windows of a row
//     are shifts of those dwords (6 / 8 VALU per row), taps on v_dot4_i32_i8, the reference's requantisation, ONE ds_write_b64
//     per lane into the pixel-linear image[k][96 B] (pitch / 4 = 8 x 3: the transposed read's 8 rows fall into distinct banks);
//     rows outside the image meet a zeroed filter row; the one lane whose fetch would start in front of the tensor fetches from
//     its first byte and shifts;
//   * one barrier; consume: wave w owns M / MB / 8 output channels (2 m tiles) for both n tiles, weight fragments straight from
//     L2, K-outer; epilogue: requantise, two v_permlane32_swap give a lane two 7-pixel rows = 14 contiguous bytes of one channel.
#include <stdlib.h>

#include <type_traits>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"
#include "gemm_tr_common.h"

namespace plhip {

typedef float v2f_s __attribute__((ext_vector_type(2)));

constexpr int F7_PITCH = 96;   // bytes per channel row of the image: 64 slots + pad
constexpr int F7_WAVES = 8;

// diagnostic timeline (plhip_debug_set("fused_stamps", 1)): per wave of the first 1024 blocks: 0 realtime, 1 entry, 2 first operands
// requested, 3 parameters staged (behind the first barrier), 4 produced, 5 behind the barrier, 6 multiplied, 7 realtime end
__device__ unsigned long long g_f7_stamps[1024 * F7_WAVES * 8];
int debug_read_f7_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(g_f7_stamps)) bytes = sizeof(g_f7_stamps);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_f7_stamps), bytes) == hipSuccess ? 0 : -1;
}
#define PLHIP_F7_STAMP(i)                                                                                                 \
  do {                                                                                                                    \
    if (diag && lane == 0) g_f7_stamps[((size_t)vb * F7_WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();          \
  } while (0)

// K, M: channels in / out; S: depthwise stride (input plane 7 S x 7 S); MB: blocks per image along M; PD: iterations in flight
template <int K, int M, int S, int MB, int PD, int OUT, bool DWNN, bool PWNN>
__global__ __launch_bounds__(512, K >= 1024 ? 1 : 2) void fused_dwpw7_kernel(FusedArgs a) {
  constexpr int WI = 7 * S, ND = S == 1 ? 2 : 4;   // input plane width; dwords fetched per input row
  constexpr int NGRP = K * 7;                      // (channel, output row) groups per image
  constexpr int NIT = (NGRP + 511) / 512;          // iterations
  constexpr int KS = K / 32, MTB = M / 32 / MB;    // K-steps; m tiles per block
  constexpr int MW = MTB / F7_WAVES;               // m tiles per wave
  static_assert(K % 32 == 0 && M % (32 * MB * F7_WAVES) == 0 && (S == 1 || S == 2), "geometry");
  const GemmArgs& g = a.pw;
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.dw_w); PLHIP_PRELOAD(a.dw_scale); PLHIP_PRELOAD(a.dw_bias); PLHIP_PRELOAD(a.dw_act);
  PLHIP_PRELOAD(a.dw_alpha); PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.tiles); PLHIP_PRELOAD(a.ones); PLHIP_PRELOAD(g.wp);
  PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias); PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha);
  extern __shared__ __attribute__((aligned(16))) uint8_t f7_lds[];  // image[K][F7_PITCH], then the depthwise parameters [K][32 B]
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned nb = (unsigned)a.tiles, per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;  // block-uniform
  const int b = (int)(vb / (unsigned)MB), mb = (int)(vb - (unsigned)b * MB);
  const int c = lane & 31, h = lane >> 5;
  const bool diag = (g.dbg & 32) != 0 && vb < 1024;
  if (diag && lane == 0) g_f7_stamps[((size_t)vb * F7_WAVES + wave) * 8] = __builtin_amdgcn_s_memrealtime();
  PLHIP_F7_STAMP(1);

  // ------------------------------------------------------------------ produce
  const float dw_hi2 = a.dw_act == ACT_RELU6 ? fminf(a.dw_alpha + a.dw_alpha, 254.f) : 254.f;
  const float dw_leak = a.dw_act == ACT_LEAKY ? a.dw_alpha : 1.f;
  constexpr uint32_t plane_in = (uint32_t)(WI * WI);
  uint8_t* const prm = f7_lds + (size_t)K * F7_PITCH;
  const uint8_t* const xs = reinterpret_cast<const uint8_t*>(a.x);
  uint32_t in[PD][3][ND];
  auto task = [&](int it, int& ch, int& o) {
    int gi = it * 512 + tid;
    if (gi >= NGRP) gi = NGRP - 1;  // surplus lanes of the last iteration recompute the last group (same values, same place)
    ch = gi / 7;
    o = gi - ch * 7;
  };
  using std::integral_constant;
  // row t of a group is input row S o - 1 + t, fetched from FR bytes in front of it (FR = S: 1 / 2): byte i = column i - FR
  auto fetch = [&](auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value, s = it % PD;
    int ch, o;
    task(it, ch, o);
    const int base = (int)((uint32_t)(b * K + ch) * plane_in) - S;
    const int r1 = S * o;                                     // the middle row (always inside)
    const int r0 = o == 0 ? 0 : r1 - 1;                       // above: row -1 -> row 0 against a zeroed filter row
    const int r2 = (S == 1 && o == 6) ? r1 : r1 + 1;          // below: stride 1, row 7 -> row 6 against a zeroed filter row
    int o0 = base + r0 * WI, o1 = base + r1 * WI;
    const int o2 = base + r2 * WI;
    if (it == 0) {  // (only the first channel's row 0 of the first image can start in front of the tensor)
      o0 = o0 < 0 ? 0 : o0;
      o1 = o1 < 0 ? 0 : o1;
    }
    __builtin_memcpy(in[s][0], xs + o0, 4 * ND);
    __builtin_memcpy(in[s][1], xs + o1, 4 * ND);
    __builtin_memcpy(in[s][2], xs + o2, 4 * ND);
  };
  auto compute = [&](auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value, s = it % PD;
    int ch, o;
    task(it, ch, o);
    const v4i pv = *reinterpret_cast<const v4i*>(prm + ch * 32);
    const uint32_t zt = o == 0 ? 0u : 0xffffffffu, zb = (S == 1 && o == 6) ? 0u : 0xffffffffu;
    const uint32_t wr[3] = {(uint32_t)pv[0] & zt, (uint32_t)pv[1], (uint32_t)pv[2] & zb};  // packed filter rows (w0, w1, w2, 0)
    const float dsc = __uint_as_float((uint32_t)pv[3]), dbi = __uint_as_float(*reinterpret_cast<const uint32_t*>(prm + ch * 32 + 16));
    int dacc[8];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      uint32_t d[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) d[i] = in[s][t][i];
      if (it == 0 && t < 2) {  // the lanes that fetched row 0 of the tensor from its first byte: move the bytes up by FR
        const bool fix = b == 0 && (tid == 0 || (S == 1 && tid == 1 && t == 0));  // (stride 1: row 0 is also the row above output row 1)
        if (S == 1) {
          const uint32_t e1 = __builtin_amdgcn_alignbyte(d[1], d[0], 3), e0 = d[0] << 8;
          d[1] = fix ? e1 : d[1];
          d[0] = fix ? e0 : d[0];
        } else {
          const uint32_t e3 = __builtin_amdgcn_alignbyte(d[3], d[2], 2), e2 = __builtin_amdgcn_alignbyte(d[2], d[1], 2);
          const uint32_t e1 = __builtin_amdgcn_alignbyte(d[1], d[0], 2), e0 = d[0] << 16;
          d[3] = fix ? e3 : d[3];
          d[2] = fix ? e2 : d[2];
          d[1] = fix ? e1 : d[1];
          d[0] = fix ? e0 : d[0];
        }
      }
      uint32_t win[7];  // window j: input columns S j - 1 .. S j + 1 in bytes 0..2 (byte 3 meets the filter's zero)
      if (S == 1) {     // byte i = column i - 1
        win[0] = d[0] & 0xffffff00u;  // column -1: the left padding
        win[1] = __builtin_amdgcn_alignbyte(d[1], d[0], 1);
        win[2] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);
        win[3] = __builtin_amdgcn_alignbyte(d[1], d[0], 3);
        win[4] = d[1];
        win[5] = d[1] >> 8;
        win[6] = d[1] >> 16;          // columns 5, 6 and the right padding
      } else {          // byte i = column i - 2
        win[0] = (d[0] >> 8) & 0xffffff00u;
        win[1] = __builtin_amdgcn_alignbyte(d[1], d[0], 3);
        win[2] = d[1] >> 8;
        win[3] = __builtin_amdgcn_alignbyte(d[2], d[1], 3);
        win[4] = d[2] >> 8;
        win[5] = __builtin_amdgcn_alignbyte(d[3], d[2], 3);
        win[6] = d[3] >> 8;
      }
#pragma unroll
      for (int j = 0; j < 7; ++j)
        dacc[j] = t == 0 ? sdot4_first(win[j], wr[0]) : __builtin_amdgcn_sdot4((int)win[j], (int)wr[t], dacc[j], false);
    }
    dacc[7] = 0;  // the junk slot of the row
    uint32_t pk[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int v[4] = {dacc[4 * q], dacc[4 * q + 1], dacc[4 * q + 2], dacc[4 * q + 3]};
      pk[q] = DWNN ? requant4_nn_rtz(v, dsc, dbi, dw_hi2, a.ones) : dw_requant4<ACT_LEAKY>(v, dsc, dbi, dw_leak, -254.f, 254.f);
    }
    const v2i pv2 = {(int)pk[0], (int)pk[1]};
    *reinterpret_cast<v2i*>(f7_lds + (uint32_t)ch * F7_PITCH + (uint32_t)o * 8) = pv2;
  };
  auto prime = [&](auto self, auto it_c) __attribute__((always_inline)) -> void {
    constexpr int it = decltype(it_c)::value;
    if constexpr (it < PD && it < NIT) {
      fetch(it_c);
      self(self, integral_constant<int, it + 1>{});
    }
  };
  prime(prime, integral_constant<int, 0>{});
  PLHIP_F7_STAMP(2);
  // ---- the consumer's first operands, requested here so that they arrive under the depthwise arithmetic
  const int mt0 = mb * MTB + wave * MW;
  const uint32_t trb = (uint32_t)(((h * 2) * 8 + ((lane & 15) >> 1)) * F7_PITCH + ((lane >> 4) & 1) * 16 + (lane & 1) * 8);
  const uint8_t* const wpk = reinterpret_cast<const uint8_t*>(g.wp) + (size_t)mt0 * KS * 1024;  // [mt][ks][64 lanes][16 B]
  const uint32_t wlane = (uint32_t)lane * 16;
  // weight fragments WD - 1 K-steps ahead (a fragment comes from L2: ~1 us; one step ahead every K-step waited for its weights:
  // 1024 -> 1024 took 23.4 us for 4 us of MFMAs)
  constexpr int WD = MW >= 4 ? 2 : 4;
  static_assert(KS % WD == 0, "weight ring");
  v4i Wf[WD][MW];
#pragma unroll
  for (int u = 0; u < WD - 1; ++u)
#pragma unroll
    for (int m = 0; m < MW; ++m) Wf[u][m] = *reinterpret_cast<const v4i*>(wpk + ((size_t)m * KS + u) * 1024 + wlane);
  float psc[MW], pbi[MW];
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    psc[m] = 1.f;
    pbi[m] = 0.f;
    if (OUT != OUT_I32) {
      psc[m] = g.scale[(mt0 + m) * 32 + c];
      pbi[m] = (g.bias ? g.bias : g.scale)[(mt0 + m) * 32 + c];
      if (!g.bias) pbi[m] = 0.f;
    }
  }
  // depthwise parameters of all K channels into LDS: (w0 w1 w2 0 | w3 w4 w5 0 | w6 w7 w8 0 | 2 scale | 2 bias)
  for (int i = tid; i < K; i += 512) {
    const int8_t* wp = a.dw_w + (size_t)i * 9;
    uint32_t pw0, pw1, pw2;
    __builtin_memcpy(&pw0, wp, 4);
    __builtin_memcpy(&pw1, wp + 3, 4);
    __builtin_memcpy(&pw2, wp + 5, 4);
    const float sc = a.dw_scale[i], bi = a.dw_bias ? a.dw_bias[i] : 0.f;
    uint32_t* o = reinterpret_cast<uint32_t*>(prm + i * 32);
    const v4i pv = {(int)(pw0 & 0xffffffu), (int)(pw1 & 0xffffffu), (int)(pw2 >> 8), (int)__float_as_uint(sc + sc)};
    *reinterpret_cast<v4i*>(o) = pv;
    o[4] = __float_as_uint(bi + bi);
  }
  __syncthreads();  // the parameters are in LDS (the row fetches above are in flight meanwhile)
  PLHIP_F7_STAMP(3);
  auto steps = [&](auto self, auto it_c) __attribute__((always_inline)) -> void {
    constexpr int it = decltype(it_c)::value;
    if constexpr (it < NIT) {
      compute(it_c);
      if constexpr (it + PD < NIT) fetch(integral_constant<int, it + PD>{});
      self(self, integral_constant<int, it + 1>{});
    }
  };
  steps(steps, integral_constant<int, 0>{});
  PLHIP_F7_STAMP(4);
  __syncthreads();
  PLHIP_F7_STAMP(5);

  // ------------------------------------------------------------------ consume
  v16i acc[2][MW];
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // one group of WD K-steps; FIRST: its first K-step multiplies into the constant 0 (an inline operand of the MFMA) instead of
  // accumulators zeroed by 16 v_mov each
  auto group = [&](int ks, auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int u = 0; u < WD; ++u) {
      const int kk = ks + u;
      if (kk + WD - 1 < KS) {
#pragma unroll
        for (int m = 0; m < MW; ++m)
          Wf[(u + WD - 1) % WD][m] = *reinterpret_cast<const v4i*>(wpk + ((size_t)m * KS + kk + WD - 1) * 1024 + wlane);
      }
      const uint32_t ka = trb + (uint32_t)kk * (32 * F7_PITCH);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(f7_lds + ka + n * 32));
        const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(f7_lds + ka + n * 32 + 8 * F7_PITCH));
        const v4i av = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
        for (int m = 0; m < MW; ++m)
          acc[n][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, Wf[u][m], (FIRST && u == 0) ? zero16 : acc[n][m], 0, 0, 0);
      }
    }
  };
  group(0, integral_constant<bool, true>{});
#pragma unroll 2
  for (int ks = WD; ks < KS; ks += WD) group(ks, integral_constant<bool, false>{});

  PLHIP_F7_STAMP(6);
  if (OUT != OUT_I8 && OUT != OUT_GAP) __syncthreads();  // the image is dead: its LDS becomes the output staging of the 4-byte forms
  // ------------------------------------------------------------------ epilogue
  // accumulator register r of n tile n: slot 32 n + 8 (r >> 2) + 4 h + (r & 3) = (row slot >> 3, column slot & 7); lane (c, h)
  // owns channel 32 (mt0 + m) + c
  const float hi2 = g.act == ACT_RELU6 ? fminf(g.alpha + g.alpha, 254.f) : 254.f;
  const float leak = g.act == ACT_LEAKY ? g.alpha : 1.f;
  const float fcap = g.act == ACT_RELU6 ? g.alpha : __builtin_huge_valf();
  const float flo = (g.act == ACT_RELU || g.act == ACT_RELU6) ? 0.f : -__builtin_huge_valf();
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int mch = (mt0 + m) * 32 + c;
    const float sc = psc[m], bi = pbi[m];
    const size_t obase = ((size_t)b * M + mch) * 49;
    if (OUT == OUT_GAP) {
      // the plane's average (pooling.cc:1006-, pooling_global_avg): this lane's values in slot order, then its half-wave partner's
      float part = 0.f;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int slot = 32 * n + 8 * gq + 4 * h;
          if ((slot >> 3) >= 7) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (e == 3 && (slot & 7) != 0) continue;  // (the junk slot of the row)
            float y = __fmaf_rn((float)acc[n][m][4 * gq + e], sc, bi);
            if (g.act == ACT_LEAKY) y = y > 0.f ? y : g.alpha * y;
            part += fminf(fmaxf(y, flo), fcap);
          }
        }
      const float other = __shfl_xor(part, 32);
      if (h == 0) reinterpret_cast<float*>(g.y)[(size_t)b * M + mch] = (part + other) / 49.f;
      continue;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      if (OUT == OUT_I8) {
        const float s2 = sc + sc, b2 = bi + bi;
        uint32_t edw[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          int v[4] = {acc[n][m][4 * gq], acc[n][m][4 * gq + 1], acc[n][m][4 * gq + 2], acc[n][m][4 * gq + 3]};
          edw[gq] = PWNN ? requant4_nn_rtz(v, s2, b2, hi2, a.ones) : dw_requant4<ACT_LEAKY>(v, s2, b2, leak, -254.f, 254.f);
        }
        // half exchange: every lane gets 16 consecutive slots = rows 4 n + 2 h, 4 n + 2 h + 1 of its channel
        auto s02 = __builtin_amdgcn_permlane32_swap(edw[0], edw[2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(edw[1], edw[3], false, false);
        // (p0 .. p6 x | q0 .. q6 x) -> 14 contiguous bytes
        const uint32_t A0 = s02[0], A1 = s02[1], A2 = s13[0], A3 = s13[1];
        const uint32_t B1 = __builtin_amdgcn_perm(A2, A1, 0x04020100u);   // p4 p5 p6 q0
        const uint32_t B2 = __builtin_amdgcn_alignbyte(A3, A2, 1);        // q1 q2 q3 q4
        const uint32_t B3 = A3 >> 8;                                      // q5 q6
        const int row = 4 * n + 2 * h;
        int8_t* const yr = reinterpret_cast<int8_t*>(g.y) + obase + row * 7;
        const v2i v01 = {(int)A0, (int)B1};
        if (row < 6) {
          __builtin_memcpy(yr, &v01, 8);
          __builtin_memcpy(yr + 8, &B2, 4);
          const uint16_t v3 = (uint16_t)B3;
          __builtin_memcpy(yr + 12, &v3, 2);
        } else {  // row 6: the plane's last 7 bytes
          __builtin_memcpy(yr, &A0, 4);
          const uint16_t v1 = (uint16_t)A1;
          __builtin_memcpy(yr + 4, &v1, 2);
          const uint8_t v2 = (uint8_t)(A1 >> 16);
          __builtin_memcpy(yr + 6, &v2, 1);
        }
      } else {
        // 4-byte outputs: through LDS (an m tile's 32 channels x 49 values are ONE contiguous 6272-byte piece of the output;
        // written from the accumulators they were 8 + 4 (+ 4)-byte stores to 64 different lines per instruction:
        // 1024 -> 1024 took 28.6 us)
        uint32_t* const stw = reinterpret_cast<uint32_t*>(f7_lds + (size_t)wave * (32 * 196)) + c * 49;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int slot = 32 * n + 8 * gq + 4 * h;
          const int row = slot >> 3, col = slot & 7;  // col 0: 4 values, col 4: 3
          if (row >= 7) continue;
          uint32_t* const sp = stw + row * 7 + col;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t v = (uint32_t)acc[n][m][4 * gq + e];
            if (OUT == OUT_F32) {
              float y = __fmaf_rn((float)acc[n][m][4 * gq + e], sc, bi);
              if (g.act == ACT_LEAKY) y = y > 0.f ? y : g.alpha * y;
              v = __float_as_uint(fminf(fmaxf(y, flo), fcap));
            }
            if (e < 3 || col == 0) sp[e] = v;
          }
        }
      }
    }
    if (OUT != OUT_I8 && OUT != OUT_GAP) {  // copy-out: the m tile's piece as 16-byte stores of consecutive lanes (the wave's own writes: no barrier)
      constexpr int PIECES = 32 * 196 / 16;
      const uint8_t* const stg = f7_lds + (size_t)wave * (32 * 196);
      uint8_t* const yo = reinterpret_cast<uint8_t*>(g.y) + ((size_t)b * M + (mt0 + m) * 32) * 196;
#pragma unroll
      for (int p = 0; p < (PIECES + 63) / 64; ++p) {
        const int piece = p * 64 + lane;
        if (piece < PIECES) {
          const v4i v = *reinterpret_cast<const v4i*>(stg + piece * 16);
          __builtin_memcpy(yo + (size_t)piece * 16, &v, 16);
        }
      }
    }
  }
  if (diag && lane == 0) g_f7_stamps[((size_t)vb * F7_WAVES + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
}

// shapes: 7 x 7 output planes, 3x3, pad 1: (stride 2, 512 -> 1024) and (stride 1, 1024 -> 1024): MobileNetV1's last two pairs
bool fused_small_supported(const FusedArgs& a) {
  if (!(a.oh == 7 && a.ow == 7 && a.h == a.w && a.pt == 1 && a.pl == 1 && (a.stride == 1 || a.stride == 2) && a.h == 7 * a.stride)) return false;
  if (a.n < 1 || (long)a.n * a.C * a.h * a.w >= ((long)1 << 31) - 65536 || (long)a.n * a.pw.M * 49 >= ((long)1 << 31)) return false;
  return (a.stride == 2 && a.C == 512 && a.pw.M == 1024) || (a.stride == 1 && a.C == 1024 && a.pw.M == 1024);
}

template <int K, int M, int S, int MB, int PD, int OUT>
static void launch_small_t(FusedArgs a, hipStream_t s) {
  a.tiles = a.n * MB;
  const unsigned blocks = (unsigned)((a.tiles + 7) / 8 * 8);
  size_t lds = (size_t)K * F7_PITCH + (size_t)K * 32;
  static_assert((size_t)K * F7_PITCH + (size_t)K * 32 >= (size_t)F7_WAVES * 32 * 196, "the image's LDS holds the output staging of the 4-byte forms");
  const bool dwnn = a.dw_act == ACT_RELU || a.dw_act == ACT_RELU6;
  const bool pwnn = OUT == OUT_I8 && (a.pw.act == ACT_RELU || a.pw.act == ACT_RELU6);
#define PLHIP_F7_LAUNCH(DN, PN)                                                                                  \
  do {                                                                                                           \
    auto kfn = fused_dwpw7_kernel<K, M, S, MB, PD, OUT, DN, PN>;                                                 \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);                                                 \
  } while (0)
  if (dwnn && pwnn) PLHIP_F7_LAUNCH(true, true);
  else if (dwnn) PLHIP_F7_LAUNCH(true, false);
  else if (pwnn) PLHIP_F7_LAUNCH(false, true);
  else PLHIP_F7_LAUNCH(false, false);
#undef PLHIP_F7_LAUNCH
}

template <int K, int M, int S, int MB, int PD>
static void launch_small_o(const FusedArgs& a, int out, hipStream_t s) {
  if (out == OUT_I32) launch_small_t<K, M, S, MB, PD, OUT_I32>(a, s);
  else if (out == OUT_GAP) launch_small_t<K, M, S, MB, PD, OUT_GAP>(a, s);
  else if (out == OUT_F32) launch_small_t<K, M, S, MB, PD, OUT_F32>(a, s);
  else launch_small_t<K, M, S, MB, PD, OUT_I8>(a, s);
}

void launch_fused_small(const FusedArgs& a, int out, hipStream_t s) {
  // blocks per image along M: 1 (default) = no duplicated work, half the CUs at batch 128: what several predictors in flight
  // prefer (c3: 379 k img/s against 370 k / 370 k with two blocks / the two kernels; one step in flight 300 k / 308 k / 291 k);
  // 2 (knob FUSED_SMALL = 2) = every CU gets a block, the depthwise stage computed twice: best alone
  const bool one = knob("FUSED_SMALL", 1) != 2;
  if (a.stride == 2) {
    if (one) launch_small_o<512, 1024, 2, 1, 2>(a, out, s);
    else launch_small_o<512, 1024, 2, 2, 2>(a, out, s);
  } else {
    if (one) launch_small_o<1024, 1024, 1, 1, 2>(a, out, s);
    else launch_small_o<1024, 1024, 1, 2, 2>(a, out, s);
  }
}

}  // namespace plhip
