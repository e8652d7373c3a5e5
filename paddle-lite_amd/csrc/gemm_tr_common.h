// gemm_tr_common.h — pieces shared by the transposed-read GEMM kernel (gemm_tr_i8.hip) and the fused depthwise ->
// pointwise kernel (fused_dwpw_i8.hip): the XCD-aware tile map, the LDS-staged int8 epilogue of a 128 (n) x 64 (m) wave
// tile and the partial 16-byte row store.
#pragma once
#include "plhip_device.h"

namespace plhip {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) v2i* lds_v2i_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int N>
__device__ __forceinline__ void tr_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void tr_xcd_tile_map(int b, int mblocks, int NTB, int& mb, int& nb) {
  const int ntx = (NTB + 7) >> 3;  // N blocks per XCD: contiguous ranges, blocks sharing an N tile equal mod 8
  const int x = b & 7, q = b >> 3;
  const int j = q / mblocks;
  mb = q - j * mblocks;
  nb = x * ntx + j;
}

// 16 int8 results of one channel row: bytes [skip, min(16, room)) are real (skip: leading duplicates of an end-aligned
// chunk; room: columns left in the output row — the im2col buffer's rows are padded to a multiple of 4, the output's are
// not).  p may have any alignment.
__device__ __forceinline__ void store_chunk_i8(int8_t* p, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, int skip, int room) {
  if (skip == 0 && room >= 16) {
    const v4i v = {(int)d0, (int)d1, (int)d2, (int)d3};
    __builtin_memcpy(p, &v, 16);  // possibly unaligned: fine for global memory
    return;
  }
  const uint32_t d[4] = {d0, d1, d2, d3};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (4 * q >= skip && 4 * q + 3 < room) {
      __builtin_memcpy(p + 4 * q, &d[q], 4);
    } else if (4 * q + 3 >= skip && 4 * q < room) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e >= skip && 4 * q + e < room) p[4 * q + e] = (int8_t)(d[q] >> (8 * e));
    }
  }
}

// Requantise this wave's 128 (n) x 64 (m) accumulators and lay the int8 tile out in LDS as [64 m][144 B] rows.
// Lane (c, h) owns channel rows 32u + c; register r of n tile t <-> n = 32t + 8(r>>2) + 4h + (r&3).  The requantisation
// works on DOUBLED values (gemm_epilogue.h): y2 = fma(acc, 2s, 2b) = 2y exactly, t = trunc(clamp(y2)),
// q = (t + 1 + (t >> 31)) >> 1; for relu / relu6 four results are packed first and (+1, >>1) finishes them at once.
// Two v_permlane32_swap per tile then give every lane 16 consecutive columns of its row (h = 0: 32t + 0..15, h = 1:
// 32t + 16..31): one ds_write_b128 per tile and row.  Pitch 144: the 8 lanes of a write group hit 8 distinct 16-byte
// bank slots.
// one 32 (n) x 32 (m) accumulator tile -> this lane's 16 consecutive int8 columns of its channel row (chunk 2t + h)
template <int ACT>
__device__ __forceinline__ v4i tr_requant_chunk(const v16i& acc, float s2, float b2, float alpha, float lo2, float hi2) {
  uint32_t dw[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    if (ACT == ACT_RELU || ACT == ACT_RELU6) {
      dw[gq] = pack4_nn_rtz(__fmaf_rn((float)acc[4 * gq], s2, b2), __fmaf_rn((float)acc[4 * gq + 1], s2, b2),
                            __fmaf_rn((float)acc[4 * gq + 2], s2, b2), __fmaf_rn((float)acc[4 * gq + 3], s2, b2), hi2);
    } else {
      int qv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y2 = __fmaf_rn((float)acc[4 * gq + e], s2, b2);
        if (ACT == ACT_LEAKY) y2 = y2 > 0.f ? y2 : alpha * y2;
        y2 = __builtin_amdgcn_fmed3f(y2, lo2, hi2);
        const int tq = (int)y2;
        qv[e] = (tq + 1 + (tq >> 31)) >> 1;
      }
      dw[gq] = pack4_i8(qv[0], qv[1], qv[2], qv[3]);
    }
  }
  // half exchange (lanes 32-63 of the first <-> lanes 0-31 of the second)
  auto s02 = __builtin_amdgcn_permlane32_swap(dw[0], dw[2], false, false);
  auto s13 = __builtin_amdgcn_permlane32_swap(dw[1], dw[3], false, false);
  const v4i v = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
  return v;
}

// calib-only tail of an fp32-output conv (conv2d[fp32_out] -> calib with the fp32 value unused): the int8 value is
// round_sat(inv2 * act(fma(acc, s, b))): the fp32 result is rounded first, then scaled and rounded again, exactly as the two
// instructions do (type_trans.cc:45,183-184).  Same lane / register layout as tr_stage_i8.
__device__ __forceinline__ void tr_stage_i8_calib(const v16i (&acc)[4][2], const float (&sc)[2], const float (&bi)[2], int act,
                                                  float alpha, float inv2, uint8_t* stg, int c, int h) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t dw[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        int qv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[e] = round_sat_i8(inv2 * epilogue_f32(acc[t][u][4 * gq + e], sc[u], bi[u], act, alpha));
        dw[gq] = pack4_i8(qv[0], qv[1], qv[2], qv[3]);
      }
      auto s02 = __builtin_amdgcn_permlane32_swap(dw[0], dw[2], false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(dw[1], dw[3], false, false);
      const v4i v = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      *reinterpret_cast<v4i*>(stg + (32 * u + c) * 144 + (2 * t + h) * 16) = v;
    }
  }
}

template <int ACT>
__device__ __forceinline__ void tr_stage_i8(const v16i (&acc)[4][2], const float (&sc)[2], const float (&bi)[2], float alpha,
                                            uint8_t* stg, int c, int h) {
  const float hi2 = ACT == ACT_RELU6 ? fminf(alpha + alpha, 254.f) : 254.f;
  const float lo2 = (ACT == ACT_RELU || ACT == ACT_RELU6) ? 0.f : -254.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float s2 = sc[u] + sc[u], b2 = bi[u] + bi[u];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<v4i*>(stg + (32 * u + c) * 144 + (2 * t + h) * 16) = tr_requant_chunk<ACT>(acc[t][u], s2, b2, alpha, lo2, hi2);
  }
}

}  // namespace plhip
