// conv_direct_i8.hip — direct int8 3x3 stride-2 convolution for small Cin (network stems: MobileNet 3 -> 32).
//
// Replaces (reference, ARM): DirectConv<kInt8,*> (lite/kernels/arm/conv_direct.{h,cc}) -> conv_3x3s2_direct_int8
// (lite/backends/arm/math/conv3x3s2_direct_int8.cc:32-) and its epilogue write_int32_nchwc8_to_nchw
// (conv_block_utils.h:3875-).  The GEMM formulation would need K = 27 -> 32 and an im2col buffer 9x the input; with
// Cin <= 4 the op is HBM-bound (0.55 MB in+out per image), so it is computed directly:
//   * one lane = 4 consecutive outputs of one output row, for a block of COB output channels at a time;
//   * per (ci, filter row) ONE unaligned 12-byte global load holds all taps of the 4 outputs (zero padding by byte
//     masks); the 4 three-byte windows are cut with v_alignbyte_b32;
//   * the packed filter rows (w0,w1,w2,0) of the whole layer sit in LDS ([ci*3+r][cout] dwords, broadcast reads);
//     MACs are v_dot4_i32_i8; outputs leave as one dword per (lane, channel, row): 32 lanes x 4 B coalesced.
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "gemm_epilogue.h"
#include "dw_common.h"

namespace plhip {

#define DS2_COB 16   // output channels per accumulation pass (64 accumulators)
#define DS2_MAXCIN 4

bool conv3x3s2_direct_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int pl) {
  return groups == 1 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && dh == 1 && dw == 1 && cin <= DS2_MAXCIN && cout <= 128 &&
         pl <= 3;
}

// packed block = [dot4 layout: cin*3 x coutp dwords][MFMA A fragments: ceil(cout/32) x 1 KiB] — both are always written,
// the kernel is picked per launch (the MFMA form needs Cin*9 <= 32 and OW % 4 == 0)
static size_t ds2_dot4_bytes(int cin, int cout) { return (size_t)cin * 3 * ((cout + 3) / 4 * 4) * 4; }
size_t conv3x3s2_dot4_bytes(int cin, int cout) { return ds2_dot4_bytes(cin, cout); }
size_t conv3x3s2_direct_packed_bytes(int cin, int cout) { return ds2_dot4_bytes(cin, cout) + (size_t)((cout + 31) / 32) * 1024; }

__global__ void pack_conv3x3s2_direct_kernel(const int8_t* __restrict__ w, uint32_t* __restrict__ wp, int cin, int cout, int coutp) {
  const int total = cin * 3 * coutp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int co = idx % coutp;
    const int cr = idx / coutp;  // ci*3 + r
    uint32_t v = 0;
    if (co < cout) {
      const int8_t* p = w + ((size_t)co * cin * 3 + cr) * 3;  // OIHW: ((co*cin + ci)*3 + r)*3 + q
      v = (uint32_t)(uint8_t)p[0] | ((uint32_t)(uint8_t)p[1] << 8) | ((uint32_t)(uint8_t)p[2] << 16);
    }
    wp[idx] = v;
  }
}

template <int OUT, int ACT>
__device__ __forceinline__ void ds2_store(const DirectS2Args& a, size_t off, int room, const int (&acc)[4], float s, float bi) {
  if (OUT == OUT_I32) {
    int* yp = reinterpret_cast<int*>(a.y) + off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < room) yp[j] = acc[j];
  } else if (OUT == OUT_F32) {
    float* yp = reinterpret_cast<float*>(a.y) + off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = __fmaf_rn((float)acc[j], s, bi);
      if (ACT == ACT_RELU) f = fmaxf(f, 0.f);
      if (ACT == ACT_RELU6) f = fminf(fmaxf(f, 0.f), a.alpha);
      if (ACT == ACT_LEAKY) f = f > 0.f ? f : a.alpha * f;
      if (j < room) yp[j] = f;
    }
  } else {
    const float hi2 = ACT == ACT_RELU6 ? fminf(a.alpha + a.alpha, 254.f) : 254.f;
    const float lo2 = (ACT == ACT_RELU || ACT == ACT_RELU6) ? 0.f : -254.f;
    const float s2 = s + s, b2 = bi + bi;
    uint32_t pk;
    if (ACT == ACT_RELU || ACT == ACT_RELU6) {
      pk = pack4_nn_rtz(__fmaf_rn((float)acc[0], s2, b2), __fmaf_rn((float)acc[1], s2, b2), __fmaf_rn((float)acc[2], s2, b2),
                        __fmaf_rn((float)acc[3], s2, b2), hi2);
    } else {
      int q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float y2 = __fmaf_rn((float)acc[j], s2, b2);
        if (ACT == ACT_LEAKY) y2 = y2 > 0.f ? y2 : a.alpha * y2;
        const int t = (int)__builtin_amdgcn_fmed3f(y2, lo2, hi2);
        q[j] = (t + 1 + (t >> 31)) >> 1;
      }
      pk = pack4_i8(q[0], q[1], q[2], q[3]);
    }
    int8_t* yp = reinterpret_cast<int8_t*>(a.y) + off;
    if (room >= 4) {
      __builtin_memcpy(yp, &pk, 4);
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < room) yp[j] = (int8_t)((pk >> (8 * j)) & 0xff);
    }
  }
}

template <int OUT, int ACT>
__device__ __forceinline__ void ds2_body(const DirectS2Args& a, const uint32_t* lw, const float* lsb,
                                         const uint32_t (&win)[DS2_MAXCIN * 3][4], int b, int oy, int xq) {
  const int room = a.ow - 4 * xq;
  const size_t plane = (size_t)a.oh * a.ow;
  for (int cb = 0; cb < a.cout; cb += DS2_COB) {  // uniform loop
    int acc[DS2_COB][4];
#pragma unroll
    for (int c = 0; c < DS2_COB; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[c][j] = 0;
#pragma unroll
    for (int cr = 0; cr < DS2_MAXCIN * 3; ++cr) {
      if (cr >= a.cin * 3) break;  // uniform
      const v4i* wrow = reinterpret_cast<const v4i*>(lw + cr * a.coutp + cb);  // 16-byte aligned: coutp % 4 == 0, cb % 16 == 0
#pragma unroll
      for (int c4 = 0; c4 < DS2_COB / 4; ++c4) {
        const v4i w4 = wrow[c4];  // LDS broadcast read (same address in every lane); rows beyond cout are zero / unused
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[c4 * 4 + e][j] = __builtin_amdgcn_sdot4((int)win[cr][j], w4[e], acc[c4 * 4 + e][j], false);
      }
    }
#pragma unroll
    for (int c = 0; c < DS2_COB; ++c) {
      const int co = cb + c;
      if (co >= a.cout) break;  // uniform
      // scale / bias from LDS: a global load here would queue behind the previous channels' stores (in-order vmcnt)
      const float s = lsb[co];
      const float bi = lsb[a.coutp + co];
      const size_t off = ((size_t)b * a.cout + co) * plane + (size_t)oy * a.ow + 4 * xq;
      ds2_store<OUT, ACT>(a, off, room, acc[c], s, bi);
    }
  }
}

template <int OUT>
__global__ __launch_bounds__(256) void conv3x3s2_direct_kernel(DirectS2Args a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lw[];  // [cin*3][coutp (+ room so that a COB block never runs off)]
  const int wtotal = a.cin * 3 * a.coutp;
  for (int i = threadIdx.x; i < wtotal + DS2_COB; i += 256) lw[i] = i < wtotal ? a.wp[i] : 0u;
  float* lsb = reinterpret_cast<float*>(lw + wtotal + DS2_COB);  // [coutp] scales, [coutp] biases
  for (int i = threadIdx.x; i < a.coutp; i += 256) {
    lsb[i] = (OUT != OUT_I32 && a.scale && i < a.cout) ? a.scale[i] : 1.f;
    lsb[a.coutp + i] = (OUT != OUT_I32 && a.bias && i < a.cout) ? a.bias[i] : 0.f;
  }
  __syncthreads();

  const int owq = (a.ow + 3) >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.n * a.oh * owq;
  if (gid >= total) return;
  const int xq = (int)(gid % owq);
  const long t = gid / owq;
  const int oy = (int)(t % a.oh);
  const int b = (int)(t / a.oh);
  const int start = 8 * xq - a.pl;  // input column of byte 0 of the 12-byte row window
  const int sh = start < 0 ? -start : 0;
  const int lcol = start + sh;
  const long tensor_bytes = (long)a.n * a.cin * a.h * a.w;

  uint32_t cmask[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = start + 4 * d + i;
      if (col >= 0 && col < a.w) m |= 0xffu << (8 * i);
    }
    cmask[d] = m;
  }

  uint32_t win[DS2_MAXCIN * 3][4];
#pragma unroll
  for (int cr = 0; cr < DS2_MAXCIN * 3; ++cr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) win[cr][j] = 0;
    if (cr >= a.cin * 3) continue;  // uniform
    const int ci = cr / 3, r = cr - 3 * (cr / 3);
    const int ih = 2 * oy - a.pt + r;
    uint32_t d[3] = {0, 0, 0};
    if (ih >= 0 && ih < a.h && lcol < a.w) {
      const long gofs = (((long)b * a.cin + ci) * a.h + ih) * a.w + lcol;
      const int8_t* src = a.x + gofs;
      if (gofs + 12 <= tensor_bytes) {
        __builtin_memcpy(d, src, 12);
      } else {
        for (int i = 0; i < 12; ++i)
          if (gofs + i < tensor_bytes) d[i >> 2] |= (uint32_t)(uint8_t)src[i] << (8 * (i & 3));
      }
    }
    if (sh) {
      const int s8 = 8 * sh;
      d[2] = (d[2] << s8) | (d[1] >> (32 - s8));
      d[1] = (d[1] << s8) | (d[0] >> (32 - s8));
      d[0] = d[0] << s8;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] &= cmask[i];
    // output j uses bytes 2j .. 2j+2 of the window
    win[cr][0] = d[0];
    win[cr][1] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);
    win[cr][2] = d[1];
    win[cr][3] = __builtin_amdgcn_alignbyte(d[2], d[1], 2);
  }

  if (OUT == OUT_I32) {
    ds2_body<OUT, ACT_NONE>(a, lw, lsb, win, b, oy, xq);
    return;
  }
  switch (a.act) {
    case ACT_RELU: ds2_body<OUT, ACT_RELU>(a, lw, lsb, win, b, oy, xq); break;
    case ACT_RELU6: ds2_body<OUT, ACT_RELU6>(a, lw, lsb, win, b, oy, xq); break;
    case ACT_LEAKY: ds2_body<OUT, ACT_LEAKY>(a, lw, lsb, win, b, oy, xq); break;
    default: ds2_body<OUT, ACT_NONE>(a, lw, lsb, win, b, oy, xq); break;
  }
}

// =====================================================================================================================
// MFMA form of the stem (Cin*9 <= 27, OW % 4 == 0).  The 27 taps are ONE K-step of v_mfma_i32_32x32x32_i8.
// Lane (c, h): c = quad (4 consecutive ox, one per MFMA i), h = k half.  The k order is free as long as the packed A
// fragments agree, so it is chosen to give both halves the SAME structure: half h owns the row windows
// cr = 5h .. 5h+4  (cr = ci*3 + filter row; windows >= 3*Cin do not exist and read as zero), 3 taps each:
//     operand bytes = (L0.0 L0.1 L0.2 L1.0 | L1.1 L1.2 L2.0 L2.1 | L2.2 L3.0 L3.1 L3.2 | L4.0 L4.1 L4.2 0).
// A lane therefore fetches 5 rows (not 9), and builds each operand with 3 v_perm + 1 v_and using constant selectors.
// PMC on the previous form (both halves fetched all 9 windows through the guarded general fetch, 64-bit index math):
// 1057 VALU per wave against ~300 for the requantisation of its 4096 outputs -- the stem was VALU-bound at 2.5x its
// HBM time.  The 32 couts of a tile are the MFMA rows; the epilogue is the GEMM one (4 consecutive ox per lane -> one
// dword store).  The first and the last workgroup run the guarded fetch (a window may start one byte before / end a
// few bytes after the tensor); all others read the window from its true start column and mask.
template <int OUT, bool MFULL, bool GUARD>
__device__ __forceinline__ void stem_mfma_body(const DirectS2Args& a, const int8_t* __restrict__ afrag, float* lsb, int lane,
                                               int wave, int bx, int by, int bz) {
  // block = (quad tile of a row, group of 4 output rows, image), decoded by the kernel from a 1-D XCD-contiguous id: the
  // image, the output row and with them every row offset / validity are wave-uniform; only the quad index is per lane.
  const int c = lane & 31, h = lane >> 5;
  const int owq = a.ow >> 2;  // OW % 4 == 0 here
  const int b = bz;
  const int oy = by * 4 + wave;
  if (oy >= a.oh) return;  // wave-uniform; no barrier in this kernel
  int xq = bx * 32 + c;
  const bool qvalid = xq < owq;
  if (!qvalid) xq = owq - 1;
  const int start = 8 * xq - a.pl;
  const int ncr = a.cin * 3;

  GemmArgs g;
  g.y = a.y;
  g.scale = a.scale;
  g.bias = a.bias;
  g.M = a.cout;
  g.HWY = a.oh * a.ow;
  g.y_bstride = (size_t)a.cout * a.oh * a.ow;
  g.act = a.act;
  g.alpha = a.alpha;
  g.res = nullptr; g.res_relu = 0; g.y2 = nullptr; g.inv_scale2 = 0.f;  // no fused graph tail on the stem
  const v4i af0 = *reinterpret_cast<const v4i*>(afrag + (size_t)lane * 16);
  if (OUT != OUT_I32) stage_scale_bias<1, OUT>(g, 0, lane, lsb);

  uint32_t cmask[3];
  dw_col_masks<3>(start, a.w, cmask);  // -4 < start < w (host check)
  const int sh = start < 0 ? -start : 0;  // GUARD only
  int lcol = start + sh;
  if (lcol > a.w - 1) lcol = a.w - 1;
  const int img_base = b * a.cin * a.h * a.w;  // tensor < 2^31 bytes (host check)
  const long tensor_bytes = (long)a.n * a.cin * a.h * a.w;

  uint32_t win[4][5];  // [j][L]
#pragma unroll
  for (int L = 0; L < 5; ++L) {
    // window cr = 5h + L -> (ci, filter row); both candidates are wave-uniform, the half picks one
    int ro[2], rm[2], ihcs[2], cics[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int cr = 5 * hh + L;
      const int ci = cr / 3, r3 = cr % 3;
      const int ih = 2 * oy - a.pt + r3;
      const bool rv = ih >= 0 && ih < a.h && cr < ncr;
      ihcs[hh] = ih < 0 ? 0 : (ih >= a.h ? a.h - 1 : ih);
      cics[hh] = ci < a.cin ? ci : a.cin - 1;
      ro[hh] = img_base + (cics[hh] * a.h + ihcs[hh]) * a.w;
      rm[hh] = rv ? -1 : 0;
    }
    const uint32_t rmask = (uint32_t)(h ? rm[1] : rm[0]);
    uint32_t d[3];
    if (GUARD) {
      const uint32_t nomask[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
      const int cic = h ? cics[1] : cics[0], ihc = h ? ihcs[1] : ihcs[0];
      dw_load_row<3, true>(a.x + img_base + (cic * a.h) * a.w, ihc, a.h, a.w, lcol, sh,
                           tensor_bytes - img_base - (long)(cic * a.h) * a.w, nomask, d);
    } else {
      const uint32_t off = (uint32_t)((h ? ro[1] : ro[0]) + start);
      __builtin_memcpy(d, a.x + off, 12);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] &= cmask[i] & rmask;
    win[0][L] = d[0];
    win[1][L] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);
    win[2][L] = d[1];
    win[3][L] = __builtin_amdgcn_alignbyte(d[2], d[1], 2);
  }
  v4i bf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf[j][0] = (int)__builtin_amdgcn_perm(win[j][1], win[j][0], 0x04020100u);
    bf[j][1] = (int)__builtin_amdgcn_perm(win[j][2], win[j][1], 0x05040201u);
    bf[j][2] = (int)__builtin_amdgcn_perm(win[j][3], win[j][2], 0x06050402u);
    bf[j][3] = (int)(win[j][4] & 0x00ffffffu);
  }

  const int hw = oy * a.ow + 4 * xq;
  const int MT = (a.cout + 31) >> 5;
  for (int mt = 0; mt < MT; ++mt) {  // uniform
    v4i af = af0;
    if (mt > 0) {
      af = *reinterpret_cast<const v4i*>(afrag + ((size_t)mt * 64 + lane) * 16);
      if (OUT != OUT_I32) stage_scale_bias<1, OUT>(g, mt, lane, lsb);
    }
    v16i acc[1][4];
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (an inline operand of the MFMA: no accumulator zeroing)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[j], zero16, 0, 0, 0);
    if (qvalid) {
      if (OUT == OUT_I32) {
        gemm_epilogue<1, OUT, true, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw);
      } else {
        switch (a.act) {
          case ACT_RELU: gemm_epilogue<1, OUT, true, MFULL, ACT_RELU>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
          case ACT_RELU6: gemm_epilogue<1, OUT, true, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
          case ACT_LEAKY: gemm_epilogue<1, OUT, true, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
          default: gemm_epilogue<1, OUT, true, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
        }
      }
    }
  }
}

template <int OUT, bool MFULL>
__global__ __launch_bounds__(256) void conv3x3s2_mfma_kernel(DirectS2Args a, const int8_t* __restrict__ afrag) {
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.y); PLHIP_PRELOAD(a.scale); PLHIP_PRELOAD(a.bias); PLHIP_PRELOAD(afrag);
  PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.cin); PLHIP_PRELOAD(a.h); PLHIP_PRELOAD(a.w); PLHIP_PRELOAD(a.cout); PLHIP_PRELOAD(a.oh);
  PLHIP_PRELOAD(a.ow); PLHIP_PRELOAD(a.pt); PLHIP_PRELOAD(a.pl); PLHIP_PRELOAD(a.act); PLHIP_PRELOAD(a.alpha);
  __shared__ __attribute__((aligned(16))) float lsb_all[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* lsb = lsb_all[wave];
  // 1-D grid of 8 * per blocks; XCD x (= blockIdx % 8, round-robin dispatch) gets the x-th eighth of the (image, row
  // group, column tile) space, so that row groups sharing an input row sit on one L2.  All of it is wave-uniform.
  const int nx = ((a.ow >> 2) + 31) >> 5, ny = (a.oh + 3) >> 2;
  const unsigned nb = (unsigned)(nx * ny * a.n), per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;
  const int bx = (int)(vb % (unsigned)nx);
  const unsigned t = vb / (unsigned)nx;
  const int by = (int)(t % (unsigned)ny), bz = (int)(t / (unsigned)ny);
  // only the first rows of the first image / the last rows of the last image can touch bytes outside the tensor
  const bool guard = (bz == 0 && by == 0) || (bz + 1 == a.n && by + 1 == ny);
  if (guard) stem_mfma_body<OUT, MFULL, true>(a, afrag, lsb, lane, wave, bx, by, bz);
  else stem_mfma_body<OUT, MFULL, false>(a, afrag, lsb, lane, wave, bx, by, bz);
}

// A fragments: tile mt, lane (r = lane&31, h = lane>>5), byte j  <-  W[mt*32 + r][window cr = 5h + j/3][tap j%3]
// (h = 0: j < 15, h = 1: j < 12 and cr < 3*Cin; zero elsewhere) -- the k order of stem_mfma_body.
__global__ void pack_conv3x3s2_mfma_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ afrag, int cin, int cout) {
  const int total = ((cout + 31) / 32) * 1024;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 15, lane = (idx >> 4) & 63, mt = idx >> 10;
    const int co = mt * 32 + (lane & 31), hh = lane >> 5;
    const int cr = 5 * hh + j / 3, q = j % 3;
    const bool used = j < 15 && cr < cin * 3 && co < cout;
    afrag[idx] = used ? w[((size_t)co * cin * 3 + cr) * 3 + q] : (int8_t)0;
  }
}

void launch_pack_conv3x3s2_direct(const int8_t* w_oihw, uint32_t* wp, int cin, int cout, hipStream_t s) {
  const int coutp = (cout + 3) / 4 * 4;
  const int total = cin * 3 * coutp;
  hipLaunchKernelGGL(pack_conv3x3s2_direct_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_oihw, wp, cin, cout, coutp);
  int8_t* afrag = reinterpret_cast<int8_t*>(wp) + ds2_dot4_bytes(cin, cout);
  const int ftotal = ((cout + 31) / 32) * 1024;
  hipLaunchKernelGGL(pack_conv3x3s2_mfma_kernel, dim3((ftotal + 255) / 256), dim3(256), 0, s, w_oihw, afrag, cin, cout);
}

void launch_conv3x3s2_direct(const DirectS2Args& a, int out, hipStream_t s) {
  const int mfma_env = knob("STEM_MFMA", 1);
  const size_t esz = out == OUT_I8 ? 1 : 4;
  const long tensor = (long)a.n * a.cin * a.h * a.w;
  const int owq = a.ow >> 2;
  if (mfma_env && a.cin * 3 <= 9 && (a.ow & 3) == 0 && ((uintptr_t)a.y & (4 * esz - 1)) == 0 && tensor < (1L << 31) &&
      8 * (owq - 1) - a.pl < a.w && (long)((owq + 31) / 32) * ((a.oh + 3) / 4) * a.n < (1L << 31) - 8) {
    const long nblk = (long)((owq + 31) / 32) * ((a.oh + 3) / 4) * a.n;
    const dim3 blocks((unsigned)((nblk + 7) / 8 * 8));
    const int8_t* afrag = reinterpret_cast<const int8_t*>(a.wp) + ds2_dot4_bytes(a.cin, a.cout);
    const bool mfull = a.cout % 32 == 0;
#define PLHIP_STEM(O)                                                                                       \
  do {                                                                                                      \
    if (mfull) hipLaunchKernelGGL((conv3x3s2_mfma_kernel<O, true>), blocks, dim3(256), 0, s, a, afrag);  \
    else hipLaunchKernelGGL((conv3x3s2_mfma_kernel<O, false>), blocks, dim3(256), 0, s, a, afrag);       \
  } while (0)
    if (out == OUT_I32) PLHIP_STEM(OUT_I32);
    else if (out == OUT_F32) PLHIP_STEM(OUT_F32);
    else PLHIP_STEM(OUT_I8);
#undef PLHIP_STEM
    return;
  }
  const long total = (long)a.n * a.oh * ((a.ow + 3) >> 2);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  const size_t lds = ((size_t)a.cin * 3 * a.coutp + DS2_COB + 2 * a.coutp) * 4;
  if (out == OUT_I32) hipLaunchKernelGGL((conv3x3s2_direct_kernel<OUT_I32>), dim3(blocks), dim3(256), lds, s, a);
  else if (out == OUT_F32) hipLaunchKernelGGL((conv3x3s2_direct_kernel<OUT_F32>), dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv3x3s2_direct_kernel<OUT_I8>), dim3(blocks), dim3(256), lds, s, a);
}

}  // namespace plhip
