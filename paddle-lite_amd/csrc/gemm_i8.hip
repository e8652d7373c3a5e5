// gemm_i8.hip — int8 x int8 -> int32 GEMM on v_mfma_i32_32x32x32_i8 with NCHW-native operands and a fused
// per-channel dequant/requant + bias + activation epilogue, plus the weight pre-pack and im2col kernels.
//
// Replaces (reference, ARM): gemm_prepack_int8 (lite/backends/arm/math/gemm_prepacked_int8.cc:5263-5457,
// hot loop :2582-2744, epilogue :643-796), prepackA_int8 (:109-224), packb_int8 (:3285),
// im2col<int8_t> (lite/backends/arm/math/conv_impl.cc:103-153) and the batch/group driver loops of
// conv1x1s1_gemm_int8 / conv_im2col_gemm_int8 (conv_impl.cc:260-331, 490-598).
//
// Shape mapping (conv_impl.cc:275-299): per group  Y[b] (M x N) = W (M x K) * X[b] (K x N),
// M = cout/g, K = cin/g*kh*kw, N = oh*ow.  Unlike the reference, the batch is folded into N
// (n = b*HWX + hw) so that late layers (N = 49) still fill whole MFMA tiles.
//
// MI355X design
//   * one wave owns a (32*MA) x 128 output tile: MA A-fragments x 4 B-fragments of the 32x32x32 MFMA,
//     accumulators live in 64*MA VGPR/AGPRs;
//   * A (weights) is pre-packed once into MFMA fragment order, so a wave's A fragment is one fully
//     coalesced 1 KiB dwordx4 load served by L2;
//   * B (activations, K x N with N contiguous = NCHW slab) is loaded with coalesced dword loads
//     (32 lanes x 4 B = one 128-B line per k row) and transposed IN REGISTERS with v_perm_b32 into the
//     K-contiguous 16-byte-per-lane operand the MFMA wants: lane (c = lane&31, h = lane>>5) ends up
//     with, for i = 0..3, column n = 4c+i, k = 16h..16h+15.  MFMA i therefore computes columns
//     {4c+i}, so each lane finishes with 4 CONSECUTIVE n for every output row and the int8 result is
//     stored as one dword per row (32 lanes x 4 B = 128 contiguous bytes of an NCHW row);
//   * no LDS and no barrier in this first version: tiles are private to the wave.
// The MFMA's k-slot <-> (lane>>5, byte) map never matters: A and B use the same one.
#include "plhip_device.h"
#include "plhip_kernels.h"

namespace plhip {

template <int MA>
__device__ __forceinline__ void load_a(const int8_t* __restrict__ wp, int mt, int KS, int ks, int lane, v4i (&af)[MA]) {
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    const size_t off = ((size_t)((size_t)(mt * MA + a) * KS + ks) * 64 + lane) * 16;
    af[a] = *reinterpret_cast<const v4i*>(wp + off);
  }
}

// ALIGNED: every row dword is 4-byte aligned and inside the tensor.  Otherwise (dense slabs with HW % 4 != 0, e.g. the
// 7x7 layers, or a misaligned base) the dwords are read unaligned — legal for global memory on gfx950 — and the one
// dword that would cross the end of the tensor is assembled bytewise.  Columns >= HW of a 4-column group then hold
// bytes of the next row: harmless, a GEMM column only ever feeds its own (discarded) output column.
template <bool ALIGNED>
__device__ __forceinline__ void load_b(const int8_t* __restrict__ xb, int ks, int h, int K, int XP, long room, uint32_t (&raw)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    int k = ks * 32 + 16 * h + j;
    k = k < K ? k : K - 1;  // rows >= K meet zero-padded weights; only the address must stay legal
    const long off = (long)k * XP;
    if (ALIGNED) {
      raw[j] = *reinterpret_cast<const uint32_t*>(xb + off);
    } else if (off + 4 <= room) {
      uint32_t v;
      __builtin_memcpy(&v, xb + off, 4);
      raw[j] = v;
    } else {
      uint32_t v = 0;
      for (int i = 0; i < 4; ++i)
        if (off + i < room) v |= (uint32_t)(uint8_t)xb[off + i] << (8 * i);
      raw[j] = v;
    }
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
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const v16i (&acc)[MA][4], int mt, int h, int b, int hw) {
  const int hwy_room = g.HWY - hw;  // columns hw+i with i < hwy_room are real outputs (im2col pitch pad)
  const size_t ybase = (size_t)b * g.y_bstride + hw;
  const float hi2 = ACT == ACT_RELU6 ? fminf(g.alpha + g.alpha, 254.f) : 254.f;
  const float lo2 = (ACT == ACT_RELU || ACT == ACT_RELU6) ? 0.f : -254.f;
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    const int mbase = (mt * MA + a) * 32;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int m0 = mbase + 8 * gq + 4 * h;
      if (!MFULL && m0 >= g.M) continue;
      v4f sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (OUT != OUT_I32) {
        if (MFULL || m0 + 4 <= g.M) {
          __builtin_memcpy(&sc, g.scale + m0, 16);
          if (g.bias) __builtin_memcpy(&bi, g.bias + m0, 16);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (m0 + e < g.M) {
              sc[e] = g.scale[m0 + e];
              if (g.bias) bi[e] = g.bias[m0 + e];
            }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        const int m = m0 + e;
        if (!MFULL && m >= g.M) continue;
        const int v0 = acc[a][0][r], v1 = acc[a][1][r], v2 = acc[a][2][r], v3 = acc[a][3][r];
        const size_t yoff = ybase + (uint32_t)(m * g.HWY);  // one image's output is < 2^31 elements (checked on the host)
        if (OUT == OUT_I32) {
          int* yp = reinterpret_cast<int*>(g.y) + yoff;
          if (VEC_STORE) {
            v4i v = {v0, v1, v2, v3};
            *reinterpret_cast<v4i*>(yp) = v;
          } else {
            if (0 < hwy_room) yp[0] = v0;
            if (1 < hwy_room) yp[1] = v1;
            if (2 < hwy_room) yp[2] = v2;
            if (3 < hwy_room) yp[3] = v3;
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
          float* yp = reinterpret_cast<float*>(g.y) + yoff;
          if (VEC_STORE) {
            v4f v = {f[0], f[1], f[2], f[3]};
            *reinterpret_cast<v4f*>(yp) = v;
          } else {
            if (0 < hwy_room) yp[0] = f[0];
            if (1 < hwy_room) yp[1] = f[1];
            if (2 < hwy_room) yp[2] = f[2];
            if (3 < hwy_room) yp[3] = f[3];
          }
        } else {
          const float s2 = sc[e] + sc[e], b2 = bi[e] + bi[e];
          const int vv[4] = {v0, v1, v2, v3};
          uint32_t packed;
          if (ACT == ACT_RELU || ACT == ACT_RELU6) {
            uint32_t t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              t[i] = (uint32_t)__builtin_amdgcn_fmed3f(__fmaf_rn((float)vv[i], s2, b2), lo2, hi2);  // trunc, 0..254
            const uint32_t p = (t[0] | (t[1] << 8)) | ((t[2] | (t[3] << 8)) << 16);
            packed = ((p + 0x01010101u) >> 1) & 0x7f7f7f7fu;
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
          } else {
            if (0 < hwy_room) yp[0] = (int8_t)(packed & 0xff);
            if (1 < hwy_room) yp[1] = (int8_t)((packed >> 8) & 0xff);
            if (2 < hwy_room) yp[2] = (int8_t)((packed >> 16) & 0xff);
            if (3 < hwy_room) yp[3] = (int8_t)(packed >> 24);
          }
        }
      }
    }
  }
}

template <int MA, int OUT, bool VEC_STORE, bool MFULL, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_i8_nchw_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler too
  const long wid = (long)blockIdx.x * 4 + wave;
  if (wid >= (long)g.MT * g.NT) return;  // wave-uniform; the kernel uses no barrier
  const int mt = (int)(wid % g.MT);
  const int nt = (int)(wid / g.MT);
  const int c = lane & 31, h = lane >> 5;
  const int ntot = g.NB * g.HWX;  // multiple of 4 by construction

  int n4 = nt * 128 + 4 * c;
  const bool nvalid = n4 < ntot;
  if (!nvalid) n4 = 0;
  const int b = n4 / g.HWX;
  const int hw = n4 - b * g.HWX;
  const int8_t* xb = g.x + (size_t)b * g.x_bstride + hw;
  const long room = g.x_bytes - ((long)b * (long)g.x_bstride + hw);  // bytes from xb to the end of the tensor

  v16i acc[MA][4];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0;

  uint32_t raw[16];
  v4i af[MA];
  load_b<ALIGNED>(xb, 0, h, g.K, g.XP, room, raw);
  load_a<MA>(g.wp, mt, g.KS, 0, lane, af);

  for (int ks = 0; ks < g.KS; ++ks) {
    v4i bf[4];
#pragma unroll
    for (int jg = 0; jg < 4; ++jg) {
      uint32_t o0, o1, o2, o3;
      transpose4x4_b8(raw[4 * jg], raw[4 * jg + 1], raw[4 * jg + 2], raw[4 * jg + 3], o0, o1, o2, o3);
      bf[0][jg] = (int)o0;
      bf[1][jg] = (int)o1;
      bf[2][jg] = (int)o2;
      bf[3][jg] = (int)o3;
    }
    v4i ac[MA];
#pragma unroll
    for (int a = 0; a < MA; ++a) ac[a] = af[a];
    if (ks + 1 < g.KS) {  // prefetch the next K-step under this step's MFMAs
      load_b<ALIGNED>(xb, ks + 1, h, g.K, g.XP, room, raw);
      load_a<MA>(g.wp, mt, g.KS, ks + 1, lane, af);
    }
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[a][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ac[a], bf[i], acc[a][i], 0, 0, 0);
  }

  if (!nvalid) return;
  if (OUT == OUT_I32) {
    gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw);
    return;
  }
  switch (g.act) {  // wave-uniform: one straight-line epilogue per activation
    case ACT_RELU: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU>(g, acc, mt, h, b, hw); break;
    case ACT_RELU6: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw); break;
    case ACT_LEAKY: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw); break;
    default: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw); break;
  }
}

// ---- weight pre-pack: [G][Mg][Kg] row-major (OIHW flattened) -> [G][MT32][KS][64 lanes][16 B] ----
// lane (r = lane&31, h = lane>>5), byte j  <-  W[g][mt32*32 + r][ks*32 + 16h + j]   (0 outside).
__global__ void pack_weights_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, int G, int Mg, int Kg,
                                    int MT32, int KS) {
  const size_t total = (size_t)G * MT32 * KS * 1024;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 15;
    const int lane = (idx >> 4) & 63;
    size_t t = idx >> 10;
    const int ks = t % KS;
    t /= KS;
    const int mt32 = t % MT32;
    const int grp = (int)(t / MT32);
    const int m = mt32 * 32 + (lane & 31);
    const int k = ks * 32 + 16 * (lane >> 5) + j;
    int8_t v = 0;
    if (m < Mg && k < Kg) v = w[((size_t)grp * Mg + m) * Kg + k];
    wp[idx] = v;
  }
}

// ---- im2col: x NCHW -> col[b][g][Kg][Np], Np = roundup(oh*ow, 4), pad columns and OOB taps = 0 ----
// Row index k = c*kh*kw + r*kw + q (conv_impl.cc:103-153).  One thread writes one dword (4 columns).
__global__ void im2col_i8_kernel(Im2colArgs a) {
  const int np4 = a.Np >> 2;
  const size_t total = (size_t)a.rows * np4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q4 = (int)(idx % np4);
    const size_t row = idx / np4;  // (b*G + g)*Kg + k
    const int k = (int)(row % a.Kg);
    const size_t bg = row / a.Kg;
    const int grp = (int)(bg % a.G);
    const int b = (int)(bg / a.G);
    const int kq = k % a.kw;
    const int kr = (k / a.kw) % a.kh;
    const int ci = k / (a.kw * a.kh);
    const int8_t* xp = a.x + ((size_t)b * a.cin + (size_t)grp * a.cin_g + ci) * a.h * a.w;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = q4 * 4 + i;
      int v = 0;
      if (n < a.N) {
        const int oy = n / a.ow, ox = n - oy * a.ow;
        const int ih = oy * a.sh - a.pt + kr * a.dh;
        const int iw = ox * a.sw - a.pl + kq * a.dw;
        if (ih >= 0 && ih < a.h && iw >= 0 && iw < a.w) v = (uint8_t)xp[(size_t)ih * a.w + iw];
      }
      out |= (uint32_t)v << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(a.col + row * a.Np + (size_t)q4 * 4) = out;
  }
}

// ---- host-side launchers (called from plhip_capi.hip) ----
template <int MA, int OUT>
static void launch_gemm_t(const GemmArgs& g, bool vec_store, bool aligned, hipStream_t s) {
  const long waves = (long)g.MT * g.NT;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  const bool mfull = g.M % (32 * MA) == 0;
  if (!aligned)
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, false, false, false>), dim3(blocks), dim3(256), 0, s, g);
  else if (vec_store && mfull)
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, true, true, true>), dim3(blocks), dim3(256), 0, s, g);
  else if (vec_store)
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, true, false, true>), dim3(blocks), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, false, false, true>), dim3(blocks), dim3(256), 0, s, g);
}

void launch_gemm_i8(const GemmArgs& g, int ma, int out, bool vec_store, bool aligned_loads, hipStream_t s) {
  if (!aligned_loads) vec_store = false;
  if (ma == 1) {
    if (out == OUT_I32) launch_gemm_t<1, OUT_I32>(g, vec_store, aligned_loads, s);
    else if (out == OUT_F32) launch_gemm_t<1, OUT_F32>(g, vec_store, aligned_loads, s);
    else launch_gemm_t<1, OUT_I8>(g, vec_store, aligned_loads, s);
  } else {
    if (out == OUT_I32) launch_gemm_t<2, OUT_I32>(g, vec_store, aligned_loads, s);
    else if (out == OUT_F32) launch_gemm_t<2, OUT_F32>(g, vec_store, aligned_loads, s);
    else launch_gemm_t<2, OUT_I8>(g, vec_store, aligned_loads, s);
  }
}

void launch_pack_weights(const int8_t* w, int8_t* wp, int G, int Mg, int Kg, int MT32, int KS, hipStream_t s) {
  const size_t total = (size_t)G * MT32 * KS * 1024;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wp, G, Mg, Kg, MT32, KS);
}

void launch_im2col(const Im2colArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.rows * (a.Np >> 2);
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(im2col_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

}  // namespace plhip
