// depthwise_i8.hip — int8 depthwise convolution (groups == cin == cout), NCHW, fused epilogue.
//
// Replaces (reference, ARM): conv_depthwise_3x3_int8_{fp32,int8} / conv_depthwise_5x5_int8_{fp32,int8}
// (lite/backends/arm/math/conv_impl.cc:798-1184) -> conv3x3s1_depthwise_int8.cc, conv3x3s2_depthwise_int8.cc,
// conv5x5s{1,2}_depthwise_int8.cc, and their epilogue write_int32_nchwc8_to_nchw (conv_block_utils.h:3875-).
// Semantics: y[n,c,oy,ox] = epi( sum_{r,q} x[n,c,oy*s-pt+r*d, ox*s-pl+q*d] * w[c,r,q] ), OOB taps = 0.
//
// MI355X design (HBM-bound op: 9 MAC per output byte)
//   * a workgroup stages a zero-padded band of PB planes into LDS with plain coalesced dword loads, so the
//     compute phase needs no bounds checks and reads only ALIGNED dwords;
//   * each lane produces 4 consecutive outputs of one row: per filter row it reads 3-4 LDS dwords,
//     cuts the 4 sliding windows out with v_alignbyte_b32 and multiplies them with the packed filter row
//     by v_dot4_i32_i8 (2 VALU per output per filter row) — no MFMA reshaping;
//   * results leave as one dword (4 x int8) or one 16-B vector (fp32 / int32) per lane, coalesced along W.
// Fast paths: 3x3 and 5x5, stride 1 and 2, dilation 1, any padding.  Everything else (other k, dilation)
// takes a scalar LDS-byte path in the same kernel.
// 3x3 and 5x5 with stride 1 / 2, dilation 1 and left padding <= 3 do not go through LDS at all: the direct strip kernels
// below (depthwise3x3_direct_kernel / depthwise5x5_direct_kernel: a lane = 4 outputs x RS rows, the input rows fetched
// straight into registers, windows cut by v_alignbyte, v_dot4 per filter row + one more dot4 for a 5x5 row's fifth tap).
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"

namespace plhip {

template <int OUT>
__device__ __forceinline__ void dw_store4(const DwArgs& a, size_t off, int ox0, const int (&acc)[4], float s, float bi) {
  const int room = a.ow - ox0;
  const bool vec = (a.ow & 3) == 0;
  if (OUT == OUT_I32) {
    int* yp = reinterpret_cast<int*>(a.y) + off;
    if (vec) {
      v4i v = {acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<v4i*>(yp) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < room) yp[j] = acc[j];
    }
    return;
  }
  float f[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = epilogue_f32(acc[j], s, bi, a.act, a.alpha);
  if (OUT == OUT_F32) {
    float* yp = reinterpret_cast<float*>(a.y) + off;
    if (vec) {
      v4f v = {f[0], f[1], f[2], f[3]};
      *reinterpret_cast<v4f*>(yp) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < room) yp[j] = f[j];
    }
  } else {
    int8_t* yp = reinterpret_cast<int8_t*>(a.y) + off;
    const int q0 = round_sat_i8(f[0]), q1 = round_sat_i8(f[1]), q2 = round_sat_i8(f[2]), q3 = round_sat_i8(f[3]);
    if (vec) {
      *reinterpret_cast<uint32_t*>(yp) = pack4_i8(q0, q1, q2, q3);
    } else {
      if (0 < room) yp[0] = (int8_t)q0;
      if (1 < room) yp[1] = (int8_t)q1;
      if (2 < room) yp[2] = (int8_t)q2;
      if (3 < room) yp[3] = (int8_t)q3;
    }
  }
}

// window of 4 bytes starting at compile-time byte offset O of the dword array d[ND]
template <int O, int ND>
__device__ __forceinline__ uint32_t window(const uint32_t (&d)[ND]) {
  constexpr int idx = O >> 2, sft = O & 3;
  static_assert(idx < ND, "window start outside the loaded dwords");
  const uint32_t lo = d[idx];
  const uint32_t hi = (idx + 1 < ND) ? d[(idx + 1 < ND) ? idx + 1 : idx] : 0u;
  if (sft == 0) return lo;
  return __builtin_amdgcn_alignbyte(hi, lo, sft);
}

template <int KW, int S, int SHF>
__device__ __forceinline__ void dw_rows_fast(const uint8_t* lds_rows, int pitch, const uint32_t* wpk, int (&acc)[4]) {
  constexpr int MAXO = SHF + 3 * S + (KW - 1);
  constexpr int ND = MAXO / 4 + 1;
#pragma unroll
  for (int r = 0; r < KW; ++r) {
    const uint32_t* rp = reinterpret_cast<const uint32_t*>(lds_rows + r * pitch);
    uint32_t d[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = rp[i];
    const int w0 = (int)wpk[2 * r];
    const int w1 = (int)wpk[2 * r + 1];
    acc[0] = __builtin_amdgcn_sdot4((int)window<SHF + 0 * S, ND>(d), w0, acc[0], false);
    acc[1] = __builtin_amdgcn_sdot4((int)window<SHF + 1 * S, ND>(d), w0, acc[1], false);
    acc[2] = __builtin_amdgcn_sdot4((int)window<SHF + 2 * S, ND>(d), w0, acc[2], false);
    acc[3] = __builtin_amdgcn_sdot4((int)window<SHF + 3 * S, ND>(d), w0, acc[3], false);
    if constexpr (KW == 5) {
      acc[0] = __builtin_amdgcn_sdot4((int)window<SHF + 0 * S + 4, ND>(d), w1, acc[0], false);
      acc[1] = __builtin_amdgcn_sdot4((int)window<SHF + 1 * S + 4, ND>(d), w1, acc[1], false);
      acc[2] = __builtin_amdgcn_sdot4((int)window<SHF + 2 * S + 4, ND>(d), w1, acc[2], false);
      acc[3] = __builtin_amdgcn_sdot4((int)window<SHF + 3 * S + 4, ND>(d), w1, acc[3], false);
    }
  }
}

// FAST: 0 = generic scalar path, else KW*10 + S (31, 32, 51, 52)
template <int OUT, int FAST>
__global__ __launch_bounds__(256) void depthwise_i8_kernel(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int pgroup = blockIdx.x / a.bands;
  const int band = blockIdx.x - pgroup * a.bands;
  const int p0 = pgroup * a.PB;
  const int npl = min(a.PB, a.planes - p0);
  const int oy0 = band * a.OB;
  const int nrows = min(a.OB, a.oh - oy0);
  const int iy_base = oy0 * a.sh - a.pt;
  const int OFF = (a.pl + 3) & ~3;
  const int kk = a.kh * a.kw;

  // LDS carve: [tile PB*in_rows*pitch][wpk PB*kh*2 dwords][scale PB][bias PB][wraw PB*kk bytes]
  const int tile_bytes = a.PB * a.in_rows * a.pitch;
  uint8_t* tile = smem;
  uint32_t* wpk = reinterpret_cast<uint32_t*>(smem + tile_bytes);
  float* lsc = reinterpret_cast<float*>(wpk + a.PB * a.kh * 2);
  float* lbi = lsc + a.PB;
  int8_t* wraw = reinterpret_cast<int8_t*>(lbi + a.PB);

  // ---- stage filters / scales ----
  for (int i = tid; i < npl; i += 256) {
    const int ch = (p0 + i) % a.C;
    lsc[i] = a.scale ? a.scale[ch] : 1.f;
    lbi[i] = a.bias ? a.bias[ch] : 0.f;
  }
  for (int i = tid; i < npl * kk; i += 256) {
    const int pi = i / kk;
    const int ch = (p0 + pi) % a.C;
    wraw[i] = a.wt[(size_t)ch * kk + (i - pi * kk)];
  }
  if (FAST != 0) {
    for (int i = tid; i < npl * a.kh; i += 256) {
      const int pi = i / a.kh, r = i - pi * a.kh;
      const int ch = (p0 + pi) % a.C;
      const int8_t* wr = a.wt + (size_t)ch * kk + r * a.kw;
      uint32_t lo = 0, hi = 0;
      for (int q = 0; q < a.kw && q < 4; ++q) lo |= (uint32_t)(uint8_t)wr[q] << (8 * q);
      if (a.kw > 4) hi = (uint32_t)(uint8_t)wr[4];
      wpk[2 * i] = lo;
      wpk[2 * i + 1] = hi;
    }
  }
  // ---- stage the zero-padded input band ----
  const int pd = a.pitch >> 2;
  const int tile_dw = npl * a.in_rows * pd;
  for (int i = tid; i < tile_dw; i += 256) {
    const int cd = i % pd;
    const int t = i / pd;
    const int r = t % a.in_rows;
    const int pi = t / a.in_rows;
    const int ih = iy_base + r;
    const int iw0 = 4 * cd - OFF;
    uint32_t v = 0;
    if (ih >= 0 && ih < a.h && iw0 + 3 >= 0 && iw0 < a.w) {
      const int8_t* src = a.x + ((size_t)(p0 + pi) * a.h + ih) * a.w + iw0;
      if (iw0 >= 0 && iw0 + 3 < a.w) {
        __builtin_memcpy(&v, src, 4);  // global loads may be unaligned on gfx950 (unaligned access mode)
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (iw0 + j >= 0 && iw0 + j < a.w) v |= (uint32_t)(uint8_t)src[j] << (8 * j);
      }
    }
    reinterpret_cast<uint32_t*>(tile)[(size_t)(pi * a.in_rows + r) * pd + cd] = v;
  }
  __syncthreads();

  // ---- compute: one item = 4 consecutive outputs of one row ----
  const int owq = (a.ow + 3) >> 2;
  const int items = npl * nrows * owq;
  const int shf = (OFF - a.pl) & 3;
  for (int it = tid; it < items; it += 256) {
    const int xq = it % owq;
    const int t = it / owq;
    const int oyl = t % nrows;
    const int pi = t / nrows;
    int acc[4] = {0, 0, 0, 0};
    if constexpr (FAST != 0) {
      constexpr int KW = FAST / 10, S = FAST % 10;
      const int col0 = 4 * xq * S - a.pl + OFF;  // LDS column of tap q=0 for output j=0
      const uint8_t* rows = tile + (size_t)(pi * a.in_rows + oyl * S) * a.pitch + (col0 & ~3);
      const uint32_t* wp = wpk + 2 * pi * KW;
      switch (shf) {
        case 0: dw_rows_fast<KW, S, 0>(rows, a.pitch, wp, acc); break;
        case 1: dw_rows_fast<KW, S, 1>(rows, a.pitch, wp, acc); break;
        case 2: dw_rows_fast<KW, S, 2>(rows, a.pitch, wp, acc); break;
        default: dw_rows_fast<KW, S, 3>(rows, a.pitch, wp, acc); break;
      }
    } else {
      const int8_t* wr = wraw + pi * kk;
      for (int r = 0; r < a.kh; ++r) {
        const uint8_t* row = tile + (size_t)(pi * a.in_rows + oyl * a.sh + r * a.dh) * a.pitch;
        for (int q = 0; q < a.kw; ++q) {
          const int wv = wr[r * a.kw + q];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = (4 * xq + j) * a.sw - a.pl + q * a.dw + OFF;
            acc[j] += (int)(int8_t)row[col] * wv;
          }
        }
      }
    }
    const int oy = oy0 + oyl;
    const size_t off = ((size_t)(p0 + pi) * a.oh + oy) * a.ow + 4 * xq;
    dw_store4<OUT>(a, off, 4 * xq, acc, lsc[pi], lbi[pi]);
  }
}

// =====================================================================================================================
// Fast path for the MobileNet case: 3x3, dilation 1, stride 1 or 2, left pad <= 3, any H/W.
//   * one lane = one column quad (4 consecutive outputs) x a vertical strip of RS output rows;
//   * every input row of the strip is fetched with ONE unaligned global load per lane (8 B for stride 1, 12 B for
//     stride 2) that already contains all taps of the 4 outputs; all (RS*S + 2) row loads are issued before any use,
//     so each lane keeps 10-15 loads in flight (this op is HBM-bound; bytes in flight are what matters);
//   * the 4 sliding windows are cut with v_alignbyte_b32 and hit the packed filter row with v_dot4_i32_i8; each input
//     row's windows are reused for the (up to) 3 output rows it feeds;
//   * no LDS, no barrier, no shuffles; image borders are handled by per-lane byte masks (zero padding).
// One output row of a lane's strip: `rowbase` = the output tensor advanced by the row's offset inside the strip (wave-uniform:
// the stores take it as their scalar base), `off` = the lane's element offset of (plane, strip row 0, quad) — no per-row
// 64-bit address arithmetic; the clamp bound and the rounding constant are scalar kernel arguments (a.hi2, a.ones).
template <int OUT, int ACT>
__device__ __forceinline__ void dw_finish_row(const DwArgs& a, void* rowbase, uint32_t off, int room, const int (&acc)[4], float s, float bi) {
  if (OUT == OUT_I32) {
    int* yp = reinterpret_cast<int*>(rowbase) + off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < room) yp[j] = acc[j];
  } else if (OUT == OUT_F32) {
    float* yp = reinterpret_cast<float*>(rowbase) + off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = __fmaf_rn((float)acc[j], s, bi);
      if (ACT == ACT_RELU) f = fmaxf(f, 0.f);
      if (ACT == ACT_RELU6) f = fminf(fmaxf(f, 0.f), a.alpha);
      if (ACT == ACT_LEAKY) f = f > 0.f ? f : a.alpha * f;
      if (j < room) yp[j] = f;
    }
  } else {
    const float lo2 = (ACT == ACT_RELU || ACT == ACT_RELU6) ? 0.f : -254.f;
    const uint32_t pk = dw_requant4<ACT>(acc, s + s, bi + bi, a.alpha, lo2, a.hi2, a.ones);
    int8_t* yp = reinterpret_cast<int8_t*>(rowbase) + off;
    if (room >= 4) {
      __builtin_memcpy(yp, &pk, 4);  // may be unaligned when OW % 4 != 0: fine for global memory
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < room) yp[j] = (int8_t)((pk >> (8 * j)) & 0xff);
    }
  }
}

// STAGE (int8 output, narrow planes): PMC showed the 14x14 / 7x7 layers bound by the L2 write-REQUEST rate — a store
// instruction there carries 16 row segments of <= 14 bytes, 4.4 B per request.  With 64 % owq == 0 and RS | OH the 64 lanes
// of a wave own one CONTIGUOUS output region ((64/owq) strips x RS rows x OW bytes); the results are assembled in LDS
// and leave as contiguous dwords (256 B per store instruction).
// KS = 3 or 5 (square filter): the window of 4 outputs is 3 S + KS bytes = 2 dwords at stride 1, 3 at stride 2 for both.
template <int OUT, int S, int RS, bool STAGE, int KS, bool FULLROWS>
__device__ __forceinline__ void dw3x3_finish(const DwArgs& a, const uint32_t (&in)[(RS - 1) * S + KS][S == 1 ? 2 : 3], long plane,
                                             int ch, int oy0, int xq, bool live, long gid_in, uint8_t* wlds);

template <int OUT, int S, int RS, bool TAIL, bool STAGE, int KS>
__device__ __forceinline__ void dw3x3_direct_body(const DwArgs& a, long gid_in, bool live, uint8_t* wlds) {
  // a dead lane (staging: the surplus lanes of a wave, the lanes past the end) recomputes its wave's FIRST quad: a live
  // one by the kernel's wave-level exit test, and inside whatever part of the tensor this workgroup's fetch variant is safe for
  const long gid = live ? gid_in : gid_in - (threadIdx.x & 63);
  constexpr int NIN = (RS - 1) * S + KS;  // input rows per strip
  constexpr int ND = S == 1 ? 2 : 3;      // dwords per row load
  const int owq = (a.ow + 3) >> 2;
  const int spp = (a.oh + RS - 1) / RS;  // strips per plane
  // gid -> (plane, strip, quad); shifts when the divisors are powers of two (the common case), wave-uniform choice
  int xq, sidx, ch;
  long plane;
  if (a.fast_div) {
    xq = (int)(gid & (owq - 1));
    const long strip = gid >> a.owq_log2;
    sidx = (int)(strip & (spp - 1));
    plane = strip >> a.spp_log2;
    ch = (int)(plane & (a.C - 1));
  } else {  // magic-number divisions (total_lanes < 2^31, checked by the launcher): no hardware divide sequences
    const uint32_t g32 = (uint32_t)gid;
    const uint32_t strip = fastdiv_u31(g32, a.div_owq_m, a.div_owq_s);
    xq = (int)(g32 - strip * (uint32_t)owq);
    const uint32_t pl32 = fastdiv_u31(strip, a.div_spp_m, a.div_spp_s);
    sidx = (int)(strip - pl32 * (uint32_t)spp);
    plane = (long)pl32;
    ch = (int)(pl32 - fastdiv_u31(pl32, a.div_c_m, a.div_c_s) * (uint32_t)a.C);
  }
  const int oy0 = sidx * RS;
  const int iy0 = oy0 * S - a.pt;
  const int start = 4 * xq * S - a.pl;       // input column of byte 0 of the row window
  const int sh = start < 0 ? -start : 0;     // bytes of left padding inside the window (only xq == 0)
  int lcol = start + sh;                     // first column actually loaded
  if (lcol > a.w - 1) lcol = a.w - 1;        // quads entirely right of the image: any legal address, masks are zero
  const long plane_base = plane * (long)a.h * a.w;
  const int8_t* xplane = a.x + plane_base;
  const long plane_room = (long)a.planes * a.h * a.w - plane_base;

  // per-lane validity mask of the ND*4 window bytes (after the left-pad shift): byte i <-> column start + i
  uint32_t cmask[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = start + 4 * d + i;
      if (col >= 0 && col < a.w) m |= 0xffu << (8 * i);
    }
    cmask[d] = m;
  }

  // ---- issue every row load of the strip ----
  uint32_t in[NIN][ND];
#pragma unroll
  for (int t = 0; t < NIN; ++t) dw_load_row<ND, TAIL>(xplane, iy0 + t, a.h, a.w, lcol, sh, plane_room, cmask, in[t]);

  dw3x3_finish<OUT, S, RS, STAGE, KS, false>(a, in, plane, ch, oy0, xq, live, gid_in, wlds);
}

// Second half of a strip, shared by the general and the fast row fetch: filter / scale fetch, the dot4 accumulation
// over the NIN input rows in registers, requantisation and the store (direct, or staged through LDS).
// FULLROWS: every strip has all its RS rows (RS | OH: the fast-fetch launches): no per-row test.
template <int OUT, int S, int RS, bool STAGE, int KS, bool FULLROWS>
__device__ __forceinline__ void dw3x3_finish(const DwArgs& a, const uint32_t (&in)[(RS - 1) * S + KS][S == 1 ? 2 : 3], long plane,
                                             int ch, int oy0, int xq, bool live, long gid_in, uint8_t* wlds) {
  constexpr int NIN = (RS - 1) * S + KS;
  constexpr int ND = S == 1 ? 2 : 3;
  // filter rows packed (w0, w1, w2, 0): three unaligned dword loads; the last one is taken one byte early and shifted so
  // that it never reads past the end of the filter tensor.  5x5: (w0 .. w3) and (w4, 0, 0, 0) per row, the fifth tap from the
  // top byte of the dword one byte further on (in bounds for the last row of the last channel too)
  uint32_t wr[KS], wr4[KS];
  if constexpr (KS == 5) {
    const int8_t* wp = a.wt + (size_t)ch * 25;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      uint32_t lo, hi;
      __builtin_memcpy(&lo, wp + 5 * r, 4);
      __builtin_memcpy(&hi, wp + 5 * r + 1, 4);
      wr[r] = lo;
      wr4[r] = hi >> 24;
    }
  } else {
    const int8_t* wp = a.wt + (size_t)ch * 9;
    uint32_t w0, w1, w2;
    __builtin_memcpy(&w0, wp, 4);
    __builtin_memcpy(&w1, wp + 3, 4);
    __builtin_memcpy(&w2, wp + 5, 4);
    wr[0] = w0 & 0xffffffu;
    wr[1] = w1 & 0xffffffu;
    wr[2] = w2 >> 8;
  }
  const float sc = a.scale ? a.scale[ch] : 1.f;
  const float bi = a.bias ? a.bias[ch] : 0.f;

  int acc[RS][4];

#pragma unroll
  for (int t = 0; t < NIN; ++t) {
    uint32_t win[4];
    if (S == 1) {
      win[0] = in[t][0];
      win[1] = __builtin_amdgcn_alignbyte(in[t][1], in[t][0], 1);
      win[2] = __builtin_amdgcn_alignbyte(in[t][1], in[t][0], 2);
      win[3] = __builtin_amdgcn_alignbyte(in[t][1], in[t][0], 3);
    } else {
      win[0] = in[t][0];
      win[1] = __builtin_amdgcn_alignbyte(in[t][1], in[t][0], 2);
      win[2] = in[t][1];
      win[3] = __builtin_amdgcn_alignbyte(in[t][ND - 1], in[t][1], 2);
    }
    uint32_t win4[4];  // 5x5: a dword whose byte 0 is the fifth tap's column (byte S j + 4 of the window) of output j
    if constexpr (KS == 5) {
      if (S == 1) {
        win4[0] = in[t][1];
        win4[1] = in[t][1] >> 8;
        win4[2] = in[t][1] >> 16;
        win4[3] = in[t][1] >> 24;
      } else {
        win4[0] = in[t][1];
        win4[1] = in[t][1] >> 16;
        win4[2] = in[t][ND - 1];
        win4[3] = in[t][ND - 1] >> 16;
      }
    }
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      if ((t - r) % S != 0) continue;
      const int o = (t - r) / S;
      if (t - r < 0 || o >= RS) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[o][j] = r == 0 ? sdot4_first(win[j], wr[0]) : __builtin_amdgcn_sdot4((int)win[j], (int)wr[r], acc[o][j], false);
        if constexpr (KS == 5) acc[o][j] = __builtin_amdgcn_sdot4((int)win4[j], (int)wr4[r], acc[o][j], false);
      }
    }
  }

  const size_t obase = ((size_t)plane * a.oh + oy0) * a.ow + 4 * xq;
  const int room = a.ow - 4 * xq;
  if (STAGE && OUT == OUT_I8) {
    // requantise into registers first (one straight-line body per activation), then LDS, then contiguous stores
    uint32_t pk[RS];
    const float s2 = sc + sc, b2 = bi + bi;
#define DW_PK(ACTV)                                                                                                  \
  _Pragma("unroll") for (int o = 0; o < RS; ++o) pk[o] =                                                             \
      dw_requant4<ACTV>(acc[o], s2, b2, a.alpha, (ACTV == ACT_RELU || ACTV == ACT_RELU6) ? 0.f : -254.f, a.hi2, a.ones);
    switch (a.act) {
      case ACT_RELU: DW_PK(ACT_RELU) break;
      case ACT_RELU6: DW_PK(ACT_RELU6) break;
      case ACT_LEAKY: DW_PK(ACT_LEAKY) break;
      default: DW_PK(ACT_NONE) break;
    }
#undef DW_PK
    const int lane = threadIdx.x & 63;
    const int owq_s = (a.ow + 3) >> 2;
    // strip index inside the wave (a wave owns lw / owq whole strips; lw = 64 when owq is a power of two)
    const int strip_l = a.owq_log2 >= 0 ? (lane >> a.owq_log2) : (int)fastdiv_u31((uint32_t)lane, a.div_owq_m, a.div_owq_s);
    const int lofs = strip_l * (RS * a.ow) + 4 * xq;     // byte offset of (row 0, quad) inside the wave's region
    if (live) {
      if ((a.ow & 1) == 0) {
        // even OW: every (row, quad) starts 2-byte aligned -> halfword writes (room is even: 2 or >= 4); uniform branch
        uint8_t* dst = wlds + lofs;
#pragma unroll
        for (int o = 0; o < RS; ++o) {
          *reinterpret_cast<uint16_t*>(dst) = (uint16_t)pk[o];
          if (room >= 4) *reinterpret_cast<uint16_t*>(dst + 2) = (uint16_t)(pk[o] >> 16);
          dst += a.ow;
        }
      } else {
#pragma unroll
        for (int o = 0; o < RS; ++o) {
          uint8_t* dst = wlds + lofs + o * a.ow;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < room) dst[j] = (uint8_t)(pk[o] >> (8 * j));
        }
      }
    }
    // the wave's region in global memory starts at the output offset of its lane 0
    const long gid0 = gid_in - lane;  // a multiple of owq
    const long strip0 = a.owq_log2 >= 0 ? (gid0 >> a.owq_log2) : (long)fastdiv_u31((uint32_t)gid0, a.div_owq_m, a.div_owq_s);
    const size_t wbase = (size_t)strip0 * (RS * a.ow);
    const long lanes_left = a.total_lanes - gid0;
    const int lanes_mine = (int)(lanes_left < a.lw ? lanes_left : a.lw);
    const int nstrips = a.owq_log2 >= 0 ? (lanes_mine >> a.owq_log2) : lanes_mine / owq_s;
    const int region = nstrips * RS * a.ow;  // bytes
    int8_t* yb = reinterpret_cast<int8_t*>(a.y) + wbase;
    if ((((uintptr_t)yb | (uintptr_t)region) & 15) == 0) {
      // the common case (full waves, 16-byte multiple regions): 16 bytes per lane and instruction
      for (int i = lane * 16; i < region; i += 64 * 16)
        *reinterpret_cast<v4i*>(yb + i) = *reinterpret_cast<const v4i*>(wlds + i);
      return;
    }
    const int head = (int)((4 - ((uintptr_t)yb & 3)) & 3);  // bytes before the first aligned dword
    for (int i = lane; i < head && i < region; i += 64) yb[i] = (int8_t)wlds[i];
    const int ndw = region > head ? (region - head) >> 2 : 0;
    for (int i = lane; i < ndw; i += 64) {
      uint32_t v;
      __builtin_memcpy(&v, wlds + head + 4 * i, 4);
      *reinterpret_cast<uint32_t*>(yb + head + 4 * i) = v;
    }
    for (int i = head + 4 * ndw + lane; i < region; i += 64) yb[i] = (int8_t)wlds[i];
    return;
  }
  constexpr int ESZ = OUT == OUT_I8 ? 1 : 4;
  const uint32_t obase32 = (uint32_t)obase;  // < 2^31 elements (launcher check)
#define DW_ROWS(ACTV)                                                                                                   \
  _Pragma("unroll") for (int o = 0; o < RS; ++o) {                                                                      \
    if (FULLROWS || oy0 + o < a.oh)                                                                                     \
      dw_finish_row<OUT, ACTV>(a, reinterpret_cast<uint8_t*>(a.y) + (size_t)(o * a.ow) * ESZ, obase32, room, acc[o], sc, bi); \
  }
  if (OUT == OUT_I32) {
    DW_ROWS(ACT_NONE)
  } else {
    switch (a.act) {
      case ACT_RELU: DW_ROWS(ACT_RELU) break;
      case ACT_RELU6: DW_ROWS(ACT_RELU6) break;
      case ACT_LEAKY: DW_ROWS(ACT_LEAKY) break;
      default: DW_ROWS(ACT_NONE) break;
    }
  }
#undef DW_ROWS
}

// Fast row fetch (the MobileNet geometry): vertical padding <= 1 and RS | OH, so only the first and the last input row
// of a strip can fall outside the image, and the window is read from its true start column (pad bytes included: they
// belong to the neighbouring row / plane and are masked to zero), so no per-row clamp, shift or validity select is
// left: one 32-bit offset add, one load and ND ANDs per row.  (PMC: the general fetch spent ~14 VALU per row and the
// large layers ran at 94 % VALU issue -- this op is VALU-bound before it is HBM-bound.)  The first workgroup (a window
// may start before the tensor) and the last ones (it may end after it) use the guarded general body instead.
template <int OUT, int S, int RS, bool STAGE, int KS>
__device__ __forceinline__ void dw3x3_fast_body(const DwArgs& a, long gid_in, bool live, uint8_t* wlds) {
  // a dead lane (staging: the surplus lanes of a wave, the lanes past the end) recomputes its wave's FIRST quad: a live
  // one by the kernel's wave-level exit test, and inside whatever part of the tensor this workgroup's fetch variant is safe for
  const long gid = live ? gid_in : gid_in - (threadIdx.x & 63);
  constexpr int NIN = (RS - 1) * S + KS;
  constexpr int PV = (KS - 1) / 2;  // rows at either end of a strip that may lie outside the image (vertical padding <= PV)
  constexpr int ND = S == 1 ? 2 : 3;
  const int owq = (a.ow + 3) >> 2;
  const int spp = a.oh / RS;
  int xq, sidx, ch;
  long plane;
  if (a.fast_div) {
    xq = (int)(gid & (owq - 1));
    const long strip = gid >> a.owq_log2;
    sidx = (int)(strip & (spp - 1));
    plane = strip >> a.spp_log2;
    ch = (int)(plane & (a.C - 1));
  } else {  // magic-number divisions (total_lanes < 2^31, checked by the launcher): no hardware divide sequences
    const uint32_t g32 = (uint32_t)gid;
    const uint32_t strip = fastdiv_u31(g32, a.div_owq_m, a.div_owq_s);
    xq = (int)(g32 - strip * (uint32_t)owq);
    const uint32_t pl32 = fastdiv_u31(strip, a.div_spp_m, a.div_spp_s);
    sidx = (int)(strip - pl32 * (uint32_t)spp);
    plane = (long)pl32;
    ch = (int)(pl32 - fastdiv_u31(pl32, a.div_c_m, a.div_c_s) * (uint32_t)a.C);
  }
  const int oy0 = sidx * RS;
  const int iy0 = oy0 * S - a.pt;
  const int start = 4 * xq * S - a.pl;
  uint32_t cmask[ND];
  dw_col_masks<ND>(start, a.w, cmask);
  // rows PV .. NIN-1-PV are always inside the image; an outside row fetches row PV again (a legal address) under a zero mask
  const uint32_t offp = (uint32_t)((int)plane * a.h * a.w + (iy0 + PV) * a.w + start);
  uint32_t in[NIN][ND];
  bool ok[NIN];
#pragma unroll
  for (int t = 0; t < NIN; ++t) {
    ok[t] = true;
    if (t < PV) ok[t] = iy0 + t >= 0;
    if (t >= NIN - PV) ok[t] = iy0 + t < a.h;
    const uint32_t off = ok[t] ? offp + (uint32_t)((t - PV) * a.w) : offp;
    __builtin_memcpy(in[t], a.x + off, 4 * ND);
  }
#pragma unroll
  for (int t = 0; t < NIN; ++t)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      uint32_t m = cmask[d];
      if (t < PV || t >= NIN - PV) m = ok[t] ? m : 0u;
      in[t][d] &= m;
    }
  dw3x3_finish<OUT, S, RS, STAGE, KS, true>(a, in, plane, ch, oy0, xq, live, gid_in, wlds);
}

template <int OUT, int S, int RS, bool STAGE, bool FASTV, int KS>
__device__ __forceinline__ void dw_direct_kernel_body(DwArgs a) {
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.wt); PLHIP_PRELOAD(a.y); PLHIP_PRELOAD(a.scale); PLHIP_PRELOAD(a.bias);
  PLHIP_PRELOAD(a.planes); PLHIP_PRELOAD(a.C); PLHIP_PRELOAD(a.h); PLHIP_PRELOAD(a.w); PLHIP_PRELOAD(a.oh); PLHIP_PRELOAD(a.ow);
  PLHIP_PRELOAD(a.pt); PLHIP_PRELOAD(a.pl); PLHIP_PRELOAD(a.total_lanes); PLHIP_PRELOAD(a.owq_log2); PLHIP_PRELOAD(a.spp_log2);
  PLHIP_PRELOAD(a.fast_div); PLHIP_PRELOAD(a.stage_bytes); PLHIP_PRELOAD(a.act); PLHIP_PRELOAD(a.alpha); PLHIP_PRELOAD(a.lw); PLHIP_PRELOAD(a.nblocks); PLHIP_PRELOAD(a.hi2); PLHIP_PRELOAD(a.ones);
  PLHIP_PRELOAD(a.div_owq_m); PLHIP_PRELOAD(a.div_spp_m); PLHIP_PRELOAD(a.div_c_m); PLHIP_PRELOAD(a.div_owq_s); PLHIP_PRELOAD(a.div_spp_s); PLHIP_PRELOAD(a.div_c_s);
  extern __shared__ __attribute__((aligned(16))) uint8_t dw_stage[];  // STAGE: 4 waves x stage_bytes
  // XCD-contiguous work: workgroups are dealt round-robin over the 8 XCDs (private L2 each); giving XCD x the x-th eighth
  // of the lane space keeps neighbouring strips (which share their 2 halo rows and the cache lines at their edges) on
  // one L2 (PMC: FETCH x2 was 1.3x the input bytes with consecutive blocks on consecutive XCDs).  grid = 8 * per blocks.
  const unsigned nb = (unsigned)a.nblocks, per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;
  // STAGE: a wave owns lw = (64 / owq) * owq lanes = whole strips (lw = 64 when owq divides 64; 63 for 28-wide, 56 for
  // 56-wide planes): the 64 - lw surplus lanes compute nothing and only help with the cooperative copy-out
  const int lane_w = threadIdx.x & 63;
  long gid = STAGE ? (long)(vb * 4 + (threadIdx.x >> 6)) * a.lw + lane_w : (long)vb * 256 + threadIdx.x;
  if (!STAGE && gid >= a.total_lanes) return;
  // STAGE: lanes past the end stay (copy-out) but are wave-uniformly dropped when the whole wave is past the end; a dead
  // lane recomputes the last live quad and skips its LDS writes
  if (STAGE && gid - lane_w >= a.total_lanes) return;
  const bool live = STAGE ? (lane_w < a.lw && gid < a.total_lanes) : true;
  uint8_t* wlds = STAGE ? dw_stage + (threadIdx.x >> 6) * a.stage_bytes : nullptr;
  // only the last workgroups can touch the final bytes of the tensor: they alone pay for the guarded loads
  if (vb + 4 >= nb) dw3x3_direct_body<OUT, S, RS, true, STAGE, KS>(a, gid, live, wlds);
  else if (FASTV && vb != 0) dw3x3_fast_body<OUT, S, RS, STAGE, KS>(a, gid, live, wlds);
  else dw3x3_direct_body<OUT, S, RS, false, STAGE, KS>(a, gid, live, wlds);
}
template <int OUT, int S, int RS, bool STAGE, bool FASTV>
__global__ __launch_bounds__(256) void depthwise3x3_direct_kernel(DwArgs a) {
  dw_direct_kernel_body<OUT, S, RS, STAGE, FASTV, 3>(a);
}
// the same strip kernel with 5 filter rows of (dot4, fifth tap) per output row (conv5x5s{1,2}_depthwise_int8.cc)
template <int OUT, int S, int RS, bool STAGE, bool FASTV>
__global__ __launch_bounds__(256) void depthwise5x5_direct_kernel(DwArgs a) {
  dw_direct_kernel_body<OUT, S, RS, STAGE, FASTV, 5>(a);
}

template <int OUT, int S, int KS>
static void launch_dw_direct_s(const DwArgs& a_in, int rs, hipStream_t s) {
  const long owq = (a_in.ow + 3) >> 2;
  const long spp = (a_in.oh + rs - 1) / rs;
  const long total = (long)a_in.planes * spp * owq;
  DwArgs a = a_in;
  a.total_lanes = total;
  a.hi2 = a.act == ACT_RELU6 ? (a.alpha + a.alpha < 254.f ? a.alpha + a.alpha : 254.f) : 254.f;
  a.ones = 0x01010101u;
  auto lg2 = [](long v) { int l = 0; while ((1L << l) < v) ++l; return (1L << l) == v ? l : -1; };
  a.owq_log2 = lg2(owq);
  a.spp_log2 = lg2(spp);
  a.fast_div = a.owq_log2 >= 0 && a.spp_log2 >= 0 && lg2(a.C) >= 0;
  auto magic = [&](long d, unsigned& m, int& sh) {
    const int l = lg2(d);
    if (l >= 0) { m = 0; sh = l; return; }
    int sc = 0;
    while ((1L << sc) < d) ++sc;  // ceil(log2 d)
    m = (unsigned)(((1ULL << (31 + sc)) / (unsigned long long)d) + 1ULL);
    sh = sc - 1;
  };
  magic(owq, a.div_owq_m, a.div_owq_s);
  magic(spp, a.div_spp_m, a.div_spp_s);
  magic(a.C, a.div_c_m, a.div_c_s);
  // output staging through LDS: int8 output, narrow planes, a wave = whole strips, strips = whole rows of the plane
  const int stage_env = knob("DW_STAGE", 1);
  // (owq a power of two: a wave = 64 lanes = whole strips; otherwise a wave uses (64 / owq) * owq lanes: 63 of 64 on
  // 28-wide planes, 56 of 64 on 56-wide ones — the 14x14 layers went 19.3 -> 11.0 us with staging, and a 28x28 layer
  // moves the same bytes with the same arithmetic)
  const int stage_np2 = knob("DW_STAGE_NP2", 1);
  // measured: 28-wide planes gain ~5 % (dw6 19.7 -> 18.7 us), 56-wide ones lose ~5 %: the store-request granularity that
  // staging cures is a narrow-row effect; stage_np2 = 2 forces it for every width <= 64
  const bool stage = stage_env && OUT == OUT_I8 && a.ow <= 64 && owq <= 64 && a.oh % rs == 0 &&
                     (a.owq_log2 >= 0 || (stage_np2 == 1 && a.ow <= 32) || stage_np2 >= 2);
  a.lw = stage ? (int)((64 / owq) * owq) : 64;
  a.nblocks = stage ? (int)(((total + a.lw - 1) / a.lw + 3) / 4) : (int)((total + 255) / 256);
  const unsigned blocks = (unsigned)((a.nblocks + 7) / 8 * 8);  // 8 XCDs x equal shares (kernel: vb map)
  a.stage_bytes = stage ? (int)(((64 / owq) * rs * a.ow + 15) & ~15) : 0;
  const size_t lds = stage ? (size_t)4 * a.stage_bytes : 0;
  // fast row fetch: only the first / last row of a strip can leave the image, windows start inside the row
  const int fast_env = knob("DW_FASTV", 1);
  constexpr int PV = (KS - 1) / 2;
  const bool fastv = fast_env && a.pt <= PV && (a.oh - 1) * S + KS - 1 - a.pt <= a.h - 1 + PV && a.oh % rs == 0 &&
                     (owq - 1) * 4 * S - a.pl < a.w;
#define PLHIP_DW_LAUNCH(RSV, ST, FV)                                                                               \
  do {                                                                                                             \
    if (KS == 5) hipLaunchKernelGGL((depthwise5x5_direct_kernel<OUT, S, RSV, ST, FV>), dim3(blocks), dim3(256), lds, s, a); \
    else hipLaunchKernelGGL((depthwise3x3_direct_kernel<OUT, S, RSV, ST, FV>), dim3(blocks), dim3(256), lds, s, a);         \
  } while (0)
#define PLHIP_DW_RS(ST, FV)              \
  do {                                   \
    if (rs == 8) PLHIP_DW_LAUNCH(8, ST, FV);      \
    else if (rs == 7) PLHIP_DW_LAUNCH(7, ST, FV); \
    else PLHIP_DW_LAUNCH(4, ST, FV);              \
  } while (0)
  if (stage && fastv) PLHIP_DW_RS(true, true);
  else if (stage) PLHIP_DW_RS(true, false);
  else if (fastv) PLHIP_DW_RS(false, true);
  else PLHIP_DW_RS(false, false);
#undef PLHIP_DW_RS
#undef PLHIP_DW_LAUNCH
}

static bool launch_dw_direct(const DwArgs& a, int out, hipStream_t s) {
  const int k5_env = knob("DW5_DIRECT", 1);  // 0 = 5x5 filters on the LDS-band kernel (A/B runs)
  const bool k3 = a.kh == 3 && a.kw == 3, k5 = a.kh == 5 && a.kw == 5 && k5_env;
  if (!((k3 || k5) && a.dh == 1 && a.dw == 1 && a.sh == a.sw && (a.sw == 1 || a.sw == 2) && a.pl <= 3)) return false;
  if ((long)a.planes * a.h * a.w >= (1L << 31) || (long)a.planes * a.oh * a.ow >= (1L << 31)) return false;
  // rows per strip: amortise the 2-row halo while keeping many lanes (and bytes) in flight
  int rs;
  const int rs1_env = knob("DW_RS1", 0);
  if (a.sw == 1 && (rs1_env == 4 || rs1_env == 7 || rs1_env == 8)) rs = rs1_env;
  else if (a.sw == 1) rs = (a.oh % 8 == 0) ? 8 : (a.oh % 7 == 0 ? 7 : (a.oh >= 8 ? 8 : (a.oh >= 5 ? 7 : 4)));
  else {
    // stride 2 fetches 2 rows per output row: a taller strip amortises the per-row fetch / mask work (VALU-bound op)
    const int rs2_env = knob("DW_RS2", 0);
    if (rs2_env == 4 || rs2_env == 7 || rs2_env == 8) rs = rs2_env;
    else rs = (a.oh % 7 == 0 && a.oh <= 14) ? 7 : 4;  // taller strips measured slower (dw3 33.7 -> 37.1 us): not VALU-bound
  }
  const bool s1 = a.sw == 1;
#define PLHIP_DW_OUT(O)                                                                              \
  do {                                                                                               \
    if (k5) s1 ? launch_dw_direct_s<O, 1, 5>(a, rs, s) : launch_dw_direct_s<O, 2, 5>(a, rs, s);      \
    else s1 ? launch_dw_direct_s<O, 1, 3>(a, rs, s) : launch_dw_direct_s<O, 2, 3>(a, rs, s);         \
  } while (0)
  if (out == OUT_I32) PLHIP_DW_OUT(OUT_I32);
  else if (out == OUT_F32) PLHIP_DW_OUT(OUT_F32);
  else PLHIP_DW_OUT(OUT_I8);
#undef PLHIP_DW_OUT
  return true;
}

template <int OUT>
static void launch_dw_t(const DwArgs& a, int fast, unsigned blocks, size_t lds, hipStream_t s) {
  switch (fast) {
    case 31: hipLaunchKernelGGL((depthwise_i8_kernel<OUT, 31>), dim3(blocks), dim3(256), lds, s, a); break;
    case 32: hipLaunchKernelGGL((depthwise_i8_kernel<OUT, 32>), dim3(blocks), dim3(256), lds, s, a); break;
    case 51: hipLaunchKernelGGL((depthwise_i8_kernel<OUT, 51>), dim3(blocks), dim3(256), lds, s, a); break;
    case 52: hipLaunchKernelGGL((depthwise_i8_kernel<OUT, 52>), dim3(blocks), dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL((depthwise_i8_kernel<OUT, 0>), dim3(blocks), dim3(256), lds, s, a); break;
  }
}

int launch_depthwise(const DwArgs& a, int out, hipStream_t s) {
  if (launch_dw_direct(a, out, s)) return 0;
  int fast = 0;
  if (a.kh == a.kw && (a.kw == 3 || a.kw == 5) && a.sh == a.sw && (a.sw == 1 || a.sw == 2) && a.dh == 1 && a.dw == 1)
    fast = a.kw * 10 + a.sw;
  const size_t lds = (size_t)a.PB * a.in_rows * a.pitch + (size_t)a.PB * a.kh * 8 + (size_t)a.PB * 8 +
                     (((size_t)a.PB * a.kh * a.kw + 15) & ~(size_t)15);
  if (lds > 64 * 1024) return -3;
  const unsigned blocks = (unsigned)(((a.planes + a.PB - 1) / a.PB) * a.bands);
  if (out == OUT_I32) launch_dw_t<OUT_I32>(a, fast, blocks, lds, s);
  else if (out == OUT_F32) launch_dw_t<OUT_F32>(a, fast, blocks, lds, s);
  else launch_dw_t<OUT_I8>(a, fast, blocks, lds, s);
  return 0;
}

}  // namespace plhip
