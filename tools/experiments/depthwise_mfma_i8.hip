// depthwise_mfma_i8.hip — depthwise 3x3 (stride 1 / 2) with the nine taps on the MATRIX pipe.
//
// Replaces conv_depthwise_3x3s1_int8 / conv_depthwise_3x3s2_int8 (lite/backends/arm/math/conv3x3s1_depthwise_int8.cc:33-447,
// conv3x3s2_depthwise_int8.cc:32-; dispatch conv_impl.cc:798-1018) like depthwise3x3_direct_kernel (depthwise_i8.hip), which
// stays as the fallback for 32-bit-per-lane corner shapes.
//
// Why (round-2 PMC, profiles/r02_final_pmc_sq_c3.csv): the direct kernel issues ~13 VALU instructions per output
// (v_dot4 taps, v_alignbyte windows, masks, addressing, requantisation) and a VALU instruction costs a SIMD 4 cycles: the
// depthwise family ran at 0.36 of HBM with 50-77 % of its wave cycles queueing for the VALU, while the matrix pipe idled.
// A 3x3 depthwise plane IS a small banded matrix product per channel:
//     out[y0 + j][x0 + i] = sum_{r, c} A[i][(r, c)] * B[(r, c)][j],   A[i][(r, c)] = w[r][c - S i]  (0 <= c - S i <= 2, else 0),
//                                                                     B[(r, c)][j] = in[S (y0 + j) - pt + r][xs + c]
// i.e. ONE v_mfma_i32_16x16x64_i8 (K = 4 k-blocks of 16: rows r = 0..2 of a 16-column input window, the 4th block zero)
// produces a tile of 16 output rows x 14 (stride 1) / 7 (stride 2) output columns: 224 / 112 outputs for 16 cycles of a
// pipe that had nothing to do, and
//   * the B operand of lane (j = lane & 15, kb = lane >> 4) is 16 CONSECUTIVE input bytes of row S (y0 + j) - pt + kb:
//     one unaligned global_load_dwordx4, no window assembly at all;
//   * the A operand (the Toeplitz band of this channel's 9 weights) is built once per work item (~30 VALU); the first /
//     last tile of a row use the same band shifted by whole bytes: their 16-byte window is clamped into the row
//     (xs in [0, W - 16]), so the zero padding columns never exist as data and no load leaves the row;
//   * rows outside the image are the only thing masked (4 v_cndmask per tile);
//   * what is left on the VALU is the reference's requantisation (gemm_epilogue.h: doubled values, 5.5 instructions per
//     output) + one 4-byte store per lane: ~8 lane-operations per output instead of ~13.
// MobileNet's plane widths are multiples of 14 (112, 56, 28, 14; 7 = half a tile), so tiles waste nothing but 2 of 16 A rows.
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"

namespace plhip {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct DwMfmaArgs {
  const int8_t* x;
  const int8_t* wt;  // [C][9]
  void* y;
  const float* scale;
  const float* bias;
  int C, B, h, w, oh, ow, pt, pl;
  int TX, TY;        // tiles per plane along x (14 / 7 output columns each) and y (16 output rows each)
  int IB;            // images per work item (small planes: several, so that an item carries >= 8 MFMAs)
  int items;         // C * ceil(B / IB)
  long x_bytes;      // readable bytes of x (only a window of the LAST row of the tensor can cross it, when W < 16)
  int act;
  float alpha;
};

// logical shifts of a 16-byte vector by a WAVE-UNIFORM number of bytes (0..15); byte 0 is the low byte of v[0]
__device__ __forceinline__ v4u shr_bytes(v4u v, int n) {  // out byte k = in byte k + n
  uint64_t lo = v[0] | ((uint64_t)v[1] << 32), hi = v[2] | ((uint64_t)v[3] << 32);
  if (n >= 8) {
    lo = hi >> (8 * (n - 8));
    hi = 0;
  } else if (n > 0) {
    lo = (lo >> (8 * n)) | (hi << (64 - 8 * n));
    hi >>= 8 * n;
  }
  return v4u{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}
__device__ __forceinline__ v4u shl_bytes(v4u v, int n) {  // out byte k = in byte k - n
  uint64_t lo = v[0] | ((uint64_t)v[1] << 32), hi = v[2] | ((uint64_t)v[3] << 32);
  if (n >= 8) {
    hi = lo << (8 * (n - 8));
    lo = 0;
  } else if (n > 0) {
    hi = (hi << (8 * n)) | (lo >> (64 - 8 * n));
    lo <<= 8 * n;
  }
  return v4u{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}

template <int S, int OUT, bool NONNEG>
__global__ __launch_bounds__(256) void depthwise3x3_mfma_kernel(DwMfmaArgs a) {
  constexpr int OWT = S == 1 ? 14 : 7;  // output columns per tile; the input window advances 14 columns per tile either way
  const int lane = threadIdx.x & 63;
  const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (item >= a.items) return;
  const int c = item % a.C;              // wave-uniform
  const int b0 = (item / a.C) * a.IB;
  const int nb = a.B - b0 < a.IB ? a.B - b0 : a.IB;
  const int li = lane & 15, kb = lane >> 4;  // A: output column i = li, k-block kb; B / D: output row j = li

  // ---- the Toeplitz band of this channel, middle form: window starts at column S*14*tx - pl, so tap s of output column i
  // sits in window byte S*i + s
  v4u am = {0u, 0u, 0u, 0u};
  if (kb < 3 && li < OWT) {
    const int8_t* wr = a.wt + (size_t)c * 9 + kb * 3;
    const unsigned w24 = (unsigned)(uint8_t)wr[0] | ((unsigned)(uint8_t)wr[1] << 8) | ((unsigned)(uint8_t)wr[2] << 16);
    const int sh = S * li;  // byte position of tap 0: 0..13 (stride 1), 0..12 (stride 2)
    // place 3 bytes at byte offset sh of a 16-byte vector
    const int d = sh >> 2, bb = 8 * (sh & 3);
    const unsigned lo = w24 << bb, hi = bb > 8 ? w24 >> (32 - bb) : 0u;  // bb in {0, 8, 16, 24}: spills into the next dword for 16, 24
    am[0] = d == 0 ? lo : 0u;
    am[1] = d == 1 ? lo : (d == 0 ? hi : 0u);
    am[2] = d == 2 ? lo : (d == 1 ? hi : 0u);
    am[3] = d == 3 ? lo : (d == 2 ? hi : 0u);
  }
  // first tile: the window starts at column 0 instead of -pl: the band moves pl bytes down (the tap that would read
  // column -1 falls off: it multiplies a padding zero).  last tile: the window starts at W - 16 instead of
  // S*14*(TX-1) - pl: the band moves up by the difference.  W < 16: one tile, window = the row from column 0; bytes past
  // the row (the next row's first bytes) get no weight.
  const int wclamp = a.w >= 16 ? a.w - 16 : 0;
  const int xs_last_want = 14 * (a.TX - 1) - a.pl;
  const int xs_last = xs_last_want < 0 ? 0 : (xs_last_want > wclamp ? wclamp : xs_last_want);
  v4u a0 = shr_bytes(am, a.pl);  // tx == 0 (xs = 0)
  v4u al = a.TX > 1 ? (xs_last_want >= xs_last ? shl_bytes(am, xs_last_want - xs_last) : shr_bytes(am, xs_last - xs_last_want)) : a0;
  if (a.w < 16) {  // window byte c is input column c: columns >= W belong to the next row
    const int keep = a.w;  // bytes 0 .. W-1
    v4u m;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = keep >= 4 * k + 4 ? 0xffffffffu : (keep > 4 * k ? (1u << (8 * (keep - 4 * k))) - 1u : 0u);
    a0 = a0 & m;
    al = al & m;
  }

  // ---- requantisation constants of this channel (wave-uniform)
  float sc = 1.f, bi = 0.f;
  if (OUT != OUT_I32) {
    sc = a.scale[c];
    if (a.bias) bi = a.bias[c];
  }
  const float s2 = sc + sc, b2 = bi + bi;
  const float hi2 = a.act == ACT_RELU6 ? fminf(a.alpha + a.alpha, 254.f) : 254.f;
  const float lo2 = NONNEG ? 0.f : -254.f;
  const float leak = a.act == ACT_LEAKY ? a.alpha : 1.f;
  const float fcap = a.act == ACT_RELU6 ? a.alpha : __builtin_huge_valf();
  const float flo = (a.act == ACT_RELU || a.act == ACT_RELU6) ? 0.f : -__builtin_huge_valf();

  const size_t plane_in = (size_t)a.h * a.w, plane_out = (size_t)a.oh * a.ow;
  const int total = nb * a.TX * a.TY;

  // B operand of tile (image bimg, tile row ty, tile column tx): 16 bytes of input row S*(16 ty + j) - pt + kb from column xs
  auto load_b = [&](int bimg, int ty, int tx) -> v4u {
    const int want = 14 * tx - a.pl;
    const int xs = want < 0 ? 0 : (want > wclamp ? wclamp : want);
    const int yin = S * (16 * ty + li) - a.pt + kb;
    v4u v = {0u, 0u, 0u, 0u};
    if (kb < 3 && yin >= 0 && yin < a.h) {
      const size_t off = ((size_t)(b0 + bimg) * a.C + c) * plane_in + (size_t)yin * a.w + xs;
      if (a.w >= 16 || (long)off + 16 <= a.x_bytes) {
        __builtin_memcpy(&v, a.x + off, 16);  // unaligned: fine for global memory
      } else {  // W < 16, last row(s) of the tensor: the 16-byte window would cross its end
        for (int k = 0; k < 16; ++k)
          if ((long)off + k < a.x_bytes) v[k >> 2] |= (unsigned)(uint8_t)a.x[off + k] << (8 * (k & 3));
      }
    }
    return v;
  };
  // the tile walk (wave-uniform counters, no division): tx fastest, then ty, then the image
  auto advance = [&](int& bimg, int& ty, int& tx) {
    if (++tx == a.TX) {
      tx = 0;
      if (++ty == a.TY) {
        ty = 0;
        ++bimg;
      }
    }
  };
  int bimg = 0, ty = 0, tx = 0;     // the tile being multiplied
  int nbimg = 0, nty = 0, ntx = 0;  // the tile whose window is being fetched
  v4u bn = load_b(0, 0, 0);
  for (int it = 0; it < total; ++it) {
    const v4u bc = bn;
    advance(nbimg, nty, ntx);
    if (it + 1 < total) bn = load_b(nbimg, nty, ntx);  // the next tile's window travels while this one is multiplied and stored
    const int cb = bimg, cty = ty, ctx = tx;
    advance(bimg, ty, tx);
    const v4u av = ctx == 0 ? a0 : (ctx == a.TX - 1 ? al : am);  // wave-uniform select
    const v4i zero = {0, 0, 0, 0};
    const v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i)av, (v4i)bc, zero, 0, 0, 0);
    // D: lane (j = li, q = kb) holds out[16 ty + j][OWT tx + 4 q + 0..3]
    const int y = 16 * cty + li, x0 = OWT * ctx + 4 * kb;
    const int nvalid = (y < a.oh) ? ((OWT - 4 * kb < a.ow - x0 ? OWT - 4 * kb : a.ow - x0)) : 0;  // columns this lane owns
    if (nvalid <= 0) continue;
    const size_t yoff = ((size_t)(b0 + cb) * a.C + c) * plane_out + (size_t)y * a.ow + x0;
    if (OUT == OUT_I8) {
      uint32_t packed;
      if (NONNEG) {
        uint32_t t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (uint32_t)__builtin_amdgcn_fmed3f(__fmaf_rn((float)acc[e], s2, b2), lo2, hi2);
        const uint32_t p = (t[0] | (t[1] << 8)) | ((t[2] | (t[3] << 8)) << 16);
        packed = ((p + 0x01010101u) >> 1) & 0x7f7f7f7fu;
      } else {
        int qv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y2 = __fmaf_rn((float)acc[e], s2, b2);
          y2 = y2 > 0.f ? y2 : leak * y2;
          const int tq = (int)__builtin_amdgcn_fmed3f(y2, lo2, hi2);
          qv[e] = (tq + 1 + (tq >> 31)) >> 1;
        }
        packed = pack4_i8(qv[0], qv[1], qv[2], qv[3]);
      }
      int8_t* yp = reinterpret_cast<int8_t*>(a.y) + yoff;
      if (nvalid >= 4) {
        __builtin_memcpy(yp, &packed, 4);
      } else {
        if (nvalid >= 2) {
          const uint16_t h2 = (uint16_t)packed;
          __builtin_memcpy(yp, &h2, 2);
        }
        if (nvalid & 1) yp[nvalid - 1] = (int8_t)(packed >> (8 * (nvalid - 1)));
      }
    } else if (OUT == OUT_F32) {
      float f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = __fmaf_rn((float)acc[e], sc, bi);
        if (a.act == ACT_LEAKY) v = v > 0.f ? v : a.alpha * v;
        f[e] = fminf(fmaxf(v, flo), fcap);
      }
      float* yp = reinterpret_cast<float*>(a.y) + yoff;
      if (nvalid >= 4) {
        const v4f v = {f[0], f[1], f[2], f[3]};
        __builtin_memcpy(yp, &v, 16);
      } else {
        for (int e = 0; e < nvalid; ++e) yp[e] = f[e];
      }
    } else {
      int* yp = reinterpret_cast<int*>(a.y) + yoff;
      if (nvalid >= 4) {
        __builtin_memcpy(yp, &acc, 16);
      } else {
        for (int e = 0; e < nvalid; ++e) yp[e] = acc[e];
      }
    }
  }
}

static int dw_mfma_env() {  // PLHIP_DW_MFMA=0: the direct VALU kernel only (A/B runs)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PLHIP_DW_MFMA");
    v = e ? atoi(e) : 1;
  }
  return v;
}

template <int S, int OUT>
static void launch_dw_mfma_t(const DwMfmaArgs& m, hipStream_t s) {
  const unsigned blocks = (unsigned)((m.items + 3) / 4);
  if (OUT == OUT_I8 && !(m.act == ACT_RELU || m.act == ACT_RELU6))
    hipLaunchKernelGGL((depthwise3x3_mfma_kernel<S, OUT, false>), dim3(blocks), dim3(256), 0, s, m);
  else
    hipLaunchKernelGGL((depthwise3x3_mfma_kernel<S, OUT, true>), dim3(blocks), dim3(256), 0, s, m);
}

// true = launched.  3x3, stride 1 / 2 (both axes equal), dilation 1, left pad 0 / 1 (the band can move one byte down), any
// top pad, W >= 3; C * B planes.
bool launch_dw_mfma(const DwArgs& a, int out, hipStream_t s) {
  if (!dw_mfma_env()) return false;
  if (a.kh != 3 || a.kw != 3 || a.sh != a.sw || (a.sw != 1 && a.sw != 2) || a.dh != 1 || a.dw != 1) return false;
  if (a.pl < 0 || a.pl > 1 || a.pt < 0 || a.w < 3 || a.h < 1 || a.C < 1 || a.planes % a.C) return false;
  const int owt = a.sw == 1 ? 14 : 7;
  // every output column of a tile must find its 3 taps inside the tile's 16-byte window: the last column needs window
  // byte S*(owt-1) + 2 <= 15 (true for 14 / 7), and the clamped last window must still cover the last outputs: it does
  // when the row has at least ow*S - pl + ... columns, i.e. for every legal conv geometry
  if ((long)a.planes * a.h * a.w >= ((long)1 << 31) || (long)a.planes * a.oh * a.ow >= ((long)1 << 31)) return false;
  DwMfmaArgs m;
  m.x = a.x; m.wt = a.wt; m.y = a.y; m.scale = a.scale; m.bias = a.bias;
  m.C = a.C; m.B = a.planes / a.C; m.h = a.h; m.w = a.w; m.oh = a.oh; m.ow = a.ow; m.pt = a.pt; m.pl = a.pl;
  m.TX = (a.ow + owt - 1) / owt;
  m.TY = (a.oh + 15) / 16;
  // middle tiles use the unshifted band: their window (column 14 tx - pl) must lie inside the row as it is
  if (m.TX > 2 && 14 * (m.TX - 2) - a.pl > (a.w >= 16 ? a.w - 16 : 0)) return false;
  if (m.TX > 1 && a.w < 16) return false;
  const int tiles = m.TX * m.TY;
  m.IB = tiles >= 8 ? 1 : (8 + tiles - 1) / tiles;
  if (m.IB > m.B) m.IB = m.B;
  m.items = m.C * ((m.B + m.IB - 1) / m.IB);
  m.x_bytes = (long)a.planes * a.h * a.w;
  m.act = a.act;
  m.alpha = a.alpha;
  if (a.sw == 1) {
    if (out == OUT_I32) launch_dw_mfma_t<1, OUT_I32>(m, s);
    else if (out == OUT_F32) launch_dw_mfma_t<1, OUT_F32>(m, s);
    else launch_dw_mfma_t<1, OUT_I8>(m, s);
  } else {
    if (out == OUT_I32) launch_dw_mfma_t<2, OUT_I32>(m, s);
    else if (out == OUT_F32) launch_dw_mfma_t<2, OUT_F32>(m, s);
    else launch_dw_mfma_t<2, OUT_I8>(m, s);
  }
  return true;
}

}  // namespace plhip
