// gemm_epilogue.h — the fused dequant / requant + bias + activation epilogue of the 32x32x32 int8 MFMA tile, shared by
// the GEMM kernels (gemm_i8.hip) and the MFMA stem convolution (conv_direct_i8.hip).
#pragma once
#include "plhip_device.h"
#include "plhip_kernels.h"

namespace plhip {

// A fragment of tile row-block mt (32-row tiles mt*MA + a), K-step ks: one coalesced 1-KiB load per fragment.
template <int MA>
__device__ __forceinline__ void load_a(const int8_t* __restrict__ wp, int mt, int KS, int ks, int lane, v4i (&af)[MA]) {
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    const size_t off = ((size_t)((size_t)(mt * MA + a) * KS + ks) * 64 + lane) * 16;
    af[a] = *reinterpret_cast<const v4i*>(wp + off);
  }
}

// ---- epilogue ----------------------------------------------------------------------------------------------
// C/D layout of the 32x32 MFMA: col = lane&31 (-> n = 4c+i), row = (r&3) + 8*(r>>2) + 4*(lane>>5).  For register
// group gq = r>>2 a lane therefore owns 4 CONSECUTIVE rows 8gq + 4h + (0..3): their scales / biases are one 16-byte
// load each (same address for the 32 lanes of a half-wave).  The activation is a template parameter so that the
// 128 outputs of a lane are processed by straight-line code (no per-element branches).
//
// int8 requantisation works on DOUBLED values: y2 = fma(acc, 2s, 2b) = 2y exactly (power-of-two scaling commutes
// with rounding), t = trunc(clamp(y2)), q = round_half_away(y) = (t + 1 + (t>>31)) >> 1.  For relu / relu6 the values
// are non-negative, so the four results are packed first and (+1, >>1) is applied to the 4 bytes at once.
template <int ACT>
__device__ __forceinline__ float act2(float y2, float alpha) {  // activation on the doubled value
  if (ACT == ACT_LEAKY) return y2 > 0.f ? y2 : alpha * y2;      // alpha*(2y) == 2*(alpha*y)
  return y2;                                                    // relu / relu6 are folded into the clamp
}

template <int MA, int OUT, bool VEC_STORE, bool MFULL, int ACT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const v16i (&acc)[MA][4], int mt, int h, int b, int hw,
                                              const float* lsb, int hwy_room, int skip = 0) {
  // hwy_room: columns hw+i with i < hwy_room are real outputs (GEMM: HWY - hw, the im2col pitch pad; fused dw+pw: the
  // columns left in the output row); skip: the lane's first `skip` columns are duplicates and are not stored (the
  // end-aligned last 16-byte piece of an image in the LDS-DMA kernel when HW % 4 != 0)
  // address = per-lane part (image, column, the half's 4-row shift) + wave-uniform row offset: one 64-bit add per row
  // instead of a per-lane integer multiply (quarter rate) per row
  const size_t ylane = (size_t)b * g.y_bstride + hw + (size_t)(4 * h) * (uint32_t)g.HWY;
  const float hi2 = ACT == ACT_RELU6 ? fminf(g.alpha + g.alpha, 254.f) : 254.f;
  const float lo2 = (ACT == ACT_RELU || ACT == ACT_RELU6) ? 0.f : -254.f;
  // Fused residual operand: the 4 rows of group (a, gq) are fetched RD GROUPS AHEAD of their use.  Fetched where they are
  // used, every group's loads sit behind the previous group's stores in the in-order memory queue (the compiler may not
  // move a load above a store it cannot prove distinct) and the epilogue becomes one memory round trip per group:
  // ResNet50's residual convs ran at ~4 TB/s.
  constexpr int RD = 2;  // groups in flight ahead of the one being stored
  float rbuf[RD + 1][4][4];
  auto fetch_res = [&](int a, int gq, float (&r)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[e][i] = 0.f;
      const int m = (mt * MA + a) * 32 + 8 * gq + 4 * h + e;
      if (!MFULL && m >= g.M) continue;
      const float* rp = g.res + ylane + (size_t)((uint32_t)((mt * MA + a) * 32 + 8 * gq + e) * (uint32_t)g.HWY);
      if (VEC_STORE) {
        const v4f rv = *reinterpret_cast<const v4f*>(rp);
        r[e][0] = rv[0]; r[e][1] = rv[1]; r[e][2] = rv[2]; r[e][3] = rv[3];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i >= skip && i < hwy_room) r[e][i] = rp[i];
      }
    }
  };
  const bool has_res = OUT == OUT_F32 && g.res != nullptr;  // kernel-uniform
  if (has_res) {
#pragma unroll
    for (int p = 0; p < RD && p < MA * 4; ++p) fetch_res(p >> 2, p & 3, rbuf[p]);
  }
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    const int mbase = (mt * MA + a) * 32;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int gi = a * 4 + gq;
      if (has_res && gi + RD < MA * 4) fetch_res((gi + RD) >> 2, (gi + RD) & 3, rbuf[(gi + RD) % (RD + 1)]);
      const int m0 = mbase + 8 * gq + 4 * h;
      if (!MFULL && m0 >= g.M) continue;
      const int mu0 = mbase + 8 * gq;  // uniform part of the row index
      // scale / bias come from LDS (staged at kernel start): a global load here would sit behind the previous rows'
      // stores in the in-order vmcnt queue and serialise the epilogue into one memory round trip per row group
      v4f sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (OUT != OUT_I32) {
        sc = *reinterpret_cast<const v4f*>(lsb + a * 32 + 8 * gq + 4 * h);
        bi = *reinterpret_cast<const v4f*>(lsb + MA * 32 + a * 32 + 8 * gq + 4 * h);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        const int m = m0 + e;
        if (!MFULL && m >= g.M) continue;
        const int v0 = acc[a][0][r], v1 = acc[a][1][r], v2 = acc[a][2][r], v3 = acc[a][3][r];
        const size_t yoff = ylane + (size_t)((uint32_t)(mu0 + e) * (uint32_t)g.HWY);  // one image's output is < 2^31 elements (checked on the host)
        if (OUT == OUT_I32) {
          int* yp = reinterpret_cast<int*>(g.y) + yoff;
          if (VEC_STORE) {
            v4i v = {v0, v1, v2, v3};
            *reinterpret_cast<v4i*>(yp) = v;
          } else if (hwy_room >= 4 && skip == 0) {
            v4i v = {v0, v1, v2, v3};
            __builtin_memcpy(yp, &v, 16);  // possibly unaligned: fine for global memory
          } else {
            if (0 >= skip && 0 < hwy_room) yp[0] = v0;
            if (1 >= skip && 1 < hwy_room) yp[1] = v1;
            if (2 >= skip && 2 < hwy_room) yp[2] = v2;
            if (3 >= skip && 3 < hwy_room) yp[3] = v3;
          }
        } else if (OUT == OUT_F32) {
          const float s = sc[e], bb = bi[e];
          float f[4] = {__fmaf_rn((float)v0, s, bb), __fmaf_rn((float)v1, s, bb), __fmaf_rn((float)v2, s, bb),
                        __fmaf_rn((float)v3, s, bb)};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (ACT == ACT_RELU) f[i] = fmaxf(f[i], 0.f);
            if (ACT == ACT_RELU6) f[i] = fminf(fmaxf(f[i], 0.f), g.alpha);
            if (ACT == ACT_LEAKY) f[i] = f[i] > 0.f ? f[i] : g.alpha * f[i];
          }
          if (g.res) {  // fused residual add (+ relu): kernel-uniform; operand prefetched above
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              f[i] = f[i] + rbuf[gi % (RD + 1)][e][i];
              if (g.res_relu) f[i] = f[i] > 0.f ? f[i] : 0.f;
            }
          }
          if (g.y) {
            float* yp = reinterpret_cast<float*>(g.y) + yoff;
            if (VEC_STORE) {
              v4f v = {f[0], f[1], f[2], f[3]};
              *reinterpret_cast<v4f*>(yp) = v;
            } else if (hwy_room >= 4 && skip == 0) {
              v4f v = {f[0], f[1], f[2], f[3]};
              __builtin_memcpy(yp, &v, 16);
            } else {
              if (0 >= skip && 0 < hwy_room) yp[0] = f[0];
              if (1 >= skip && 1 < hwy_room) yp[1] = f[1];
              if (2 >= skip && 2 < hwy_room) yp[2] = f[2];
              if (3 >= skip && 3 < hwy_room) yp[3] = f[3];
            }
          }
          if (g.y2) {  // fused calib fp32 -> int8 of the value just produced
            uint32_t packed;
            // kernel-uniform: the value is >= 0 (its own relu / relu6 with no residual behind it, or the relu behind the
            // residual add): round_sat_i8 on doubled values as the int8-output path does it — 2 inv f = fl(inv f) doubled
            // exactly, t = trunc(min(., 254)), the four (t + 1) >> 1 in one v_lerp_u8: 4 VALU per output instead of ~7.75
            const bool nonneg = g.res ? g.res_relu != 0 : (ACT == ACT_RELU || ACT == ACT_RELU6);
            if (nonneg) {
              const float i2 = g.inv_scale2 + g.inv_scale2;
              packed = pack4_nn_rtz(i2 * f[0], i2 * f[1], i2 * f[2], i2 * f[3], 254.f);
            } else {
              packed = pack4_i8(round_sat_i8(g.inv_scale2 * f[0]), round_sat_i8(g.inv_scale2 * f[1]),
                                round_sat_i8(g.inv_scale2 * f[2]), round_sat_i8(g.inv_scale2 * f[3]));
            }
            int8_t* qp = g.y2 + yoff;
            if (VEC_STORE) {
              *reinterpret_cast<uint32_t*>(qp) = packed;
            } else if (hwy_room >= 4 && skip == 0) {
              __builtin_memcpy(qp, &packed, 4);
            } else {
              if (0 >= skip && 0 < hwy_room) qp[0] = (int8_t)(packed & 0xff);
              if (1 >= skip && 1 < hwy_room) qp[1] = (int8_t)((packed >> 8) & 0xff);
              if (2 >= skip && 2 < hwy_room) qp[2] = (int8_t)((packed >> 16) & 0xff);
              if (3 >= skip && 3 < hwy_room) qp[3] = (int8_t)(packed >> 24);
            }
          }
        } else {
          const float s2 = sc[e], b2 = bi[e];  // staged already doubled for int8 output (store_scale_bias)
          const int vv[4] = {v0, v1, v2, v3};
          uint32_t packed;
          if (ACT == ACT_RELU || ACT == ACT_RELU6) {
            packed = pack4_nn_rtz(__fmaf_rn((float)vv[0], s2, b2), __fmaf_rn((float)vv[1], s2, b2), __fmaf_rn((float)vv[2], s2, b2),
                                  __fmaf_rn((float)vv[3], s2, b2), hi2);
          } else {
            int q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float y2 = __builtin_amdgcn_fmed3f(act2<ACT>(__fmaf_rn((float)vv[i], s2, b2), g.alpha), lo2, hi2);
              const int t = (int)y2;
              q[i] = (t + 1 + (t >> 31)) >> 1;
            }
            packed = pack4_i8(q[0], q[1], q[2], q[3]);
          }
          int8_t* yp = reinterpret_cast<int8_t*>(g.y) + yoff;
          if (VEC_STORE) {
            *reinterpret_cast<uint32_t*>(yp) = packed;
          } else if (hwy_room >= 4 && skip == 0) {
            __builtin_memcpy(yp, &packed, 4);  // possibly unaligned: fine for global memory
          } else {
            if (0 >= skip && 0 < hwy_room) yp[0] = (int8_t)(packed & 0xff);
            if (1 >= skip && 1 < hwy_room) yp[1] = (int8_t)((packed >> 8) & 0xff);
            if (2 >= skip && 2 < hwy_room) yp[2] = (int8_t)((packed >> 16) & 0xff);
            if (3 >= skip && 3 < hwy_room) yp[3] = (int8_t)(packed >> 24);
          }
        }
      }
    }
  }
}

// Stage this wave's MA*32 folded scales and biases into LDS: lsb[0 .. MA*32) scales, lsb[MA*32 .. 2*MA*32) biases.
// Two halves on purpose: the global loads are issued early (right behind the first operand loads, so that no wait
// sits in front of the main loop) and the values reach LDS only after the K loop.
template <int MA>
__device__ __forceinline__ void load_scale_bias(const GemmArgs& g, int mt, int lane, float& s, float& b) {
  s = 1.f;
  b = 0.f;
  if (lane < MA * 32) {
    const int m = mt * MA * 32 + lane;
    if (g.scale && m < g.M) s = g.scale[m];
    if (g.bias && m < g.M) b = g.bias[m];
  }
}
template <int MA, int OUT>
__device__ __forceinline__ void store_scale_bias(float* lsb, int lane, float s, float b) {
  if (OUT == OUT_I8) {  // the int8 requantisation works on doubled values (exact: power-of-two scaling)
    s += s;
    b += b;
  }
  if (lane < MA * 32) {
    lsb[lane] = s;
    lsb[MA * 32 + lane] = b;
  }
}
template <int MA, int OUT>
__device__ __forceinline__ void stage_scale_bias(const GemmArgs& g, int mt, int lane, float* lsb) {
  float s, b;
  load_scale_bias<MA>(g, mt, lane, s, b);
  store_scale_bias<MA, OUT>(lsb, lane, s, b);
}

}  // namespace plhip
