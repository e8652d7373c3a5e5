// conv_stem7_i8.hip — direct int8 7x7 stride-2 convolution for Cin <= 3 (ResNet50's stem, 3 -> 64 @224x224).
//
// Replaces (reference): the GemmLikeConv route of lite/kernels/arm/conv_compute.cc:87-134 for this shape (im2col of K = 147 +
// gemm_s8); as a GEMM it is K = 147 -> 160 on a phase-split padded copy and an epilogue-bound launch (0.23 ms for 256
// images on the ring kernel, 5x its roofline).  Computed directly, the way conv3x3s2_mfma_kernel does the 3x3 stems:
//   * one lane = a quad of 4 consecutive outputs of one output row; MFMA j of a K-step multiplies the j-th pixels of 32 quads,
//     so a lane ends with 4 consecutive pixels per output channel: gemm_epilogue's layout (one dword / 16-byte store per
//     channel, 128 / 512 contiguous bytes per half-wave), with its fused residual / calib tails;
//   * K order: row cr = ci * 7 + r of the filter = 8 k-values (7 taps + a zero), 4 rows per K-step of the 32x32x32 MFMA, 6
//     K-steps for 21 rows; half h of the wave owns rows 4 ks + 2h, 4 ks + 2h + 1: a lane's B operand of a K-step is the
//     8-byte windows of ITS two rows — no byte permutes: per row ONE unaligned 16-byte global load holds the 13 bytes all
//     four pixels of the quad need (zero padding by byte masks), the windows of pixels 1 and 3 are two v_alignbyte each;
//   * the B operands of all 6 K-steps are built once and stay in registers (96 VGPRs) while the output channels are walked
//     32 at a time (A fragments: 6 x 16 B per lane and m tile, L2-resident).
// The first / last row groups (the only ones whose windows can leave the tensor) fetch bytewise.
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "gemm_epilogue.h"
#include "dw_common.h"

namespace plhip {

#define STEM7_KS 6  // K-steps: 4 filter rows (ci, r) each, Cin * 7 <= 24

bool conv7x7s2_stem_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int n, int h, int w,
                              int oh, int ow, int pl) {
  const int env = knob("STEM7", 1);  // 0 = the ring kernel's implicit GEMM (A/B runs)
  if (!env) return false;
  if (!(groups == 1 && kh == 7 && kw == 7 && sh == 2 && sw == 2 && dh == 1 && dw == 1)) return false;
  if (cin * 7 > 4 * STEM7_KS || pl > 3 || (ow & 3) != 0 || w < 16 || cout < 1) return false;
  const long owq = ow >> 2;
  return (long)n * cin * h * w < (1L << 31) && (long)cout * oh * ow < (1L << 31) && 8 * (owq - 1) - pl < w &&
         ((owq + 31) / 32) * (long)((oh + 3) / 4) * n < (1L << 31) - 8;
}
size_t conv7x7s2_stem_packed_bytes(int cout) { return (size_t)((cout + 31) / 32) * STEM7_KS * 1024; }

// A fragments [m tile][K-step][lane (m % 32, hh)][16 B]: byte 8 rr + s <- W[m][row cr = 4 ks + 2 hh + rr][tap s] (s < 7, cr < 7 Cin)
__global__ void pack_conv7x7s2_stem_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ afrag, int cin, int cout) {
  const int total = ((cout + 31) / 32) * STEM7_KS * 1024;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int q = idx & 15, lane = (idx >> 4) & 63, t = idx >> 10;
    const int ks = t % STEM7_KS, mt = t / STEM7_KS;
    const int m = mt * 32 + (lane & 31), cr = 4 * ks + 2 * (lane >> 5) + (q >> 3), sx = q & 7;
    afrag[idx] = (m < cout && cr < 7 * cin && sx < 7) ? w[((size_t)m * cin * 7 + cr) * 7 + sx] : (int8_t)0;
  }
}
void launch_pack_conv7x7s2_stem(const int8_t* w_oihw, int8_t* wp, int cin, int cout, hipStream_t s) {
  const int total = ((cout + 31) / 32) * STEM7_KS * 1024;
  hipLaunchKernelGGL(pack_conv7x7s2_stem_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_oihw, wp, cin, cout);
}

__device__ __forceinline__ unsigned long long stem7_low_bytes(int n) {  // the low n bytes set, 0 <= n <= 8
  return n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1ull);
}

template <int OUT, bool MFULL, bool GUARD, bool VEC>
__device__ __forceinline__ void stem7_body(const DirectS2Args& a, const int8_t* __restrict__ afrag, float* lsb, int lane, int wave,
                                           int bx, int by, int bz) {
  // block = (quad tile of a row, group of 4 output rows, image): image, output row and with them every row offset / validity
  // are wave-uniform; only the quad index is per lane
  const int c = lane & 31, h = lane >> 5;
  const int owq = a.ow >> 2;  // OW % 4 == 0 here
  const int b = bz;
  const int oy = by * 4 + wave;
  if (oy >= a.oh) return;  // wave-uniform; no barrier in this kernel
  int xq = bx * 32 + c;
  const bool qvalid = xq < owq;
  if (!qvalid) xq = owq - 1;
  const int start = 8 * xq - a.pl;  // input column of byte 0 of the quad's 16-byte row window (13 bytes used)
  const int ncr = a.cin * 7;

  GemmArgs g;
  g.y = a.y;
  g.scale = a.scale;
  g.bias = a.bias;
  g.M = a.cout;
  g.HWY = a.oh * a.ow;
  g.y_bstride = (size_t)a.cout * a.oh * a.ow;
  g.act = a.act;
  g.alpha = a.alpha;
  g.res = a.res; g.res_relu = a.res_relu; g.y2 = a.y2; g.inv_scale2 = a.inv_scale2;
  if (OUT != OUT_I32) stage_scale_bias<1, OUT>(g, 0, lane, lsb);

  // byte i of the window <-> column start + i: kept iff inside the row
  uint32_t cmask[4];
  {
    int lo = start < 0 ? -start : 0, hi = a.w - start;
    lo = lo > 16 ? 16 : lo;
    hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
    const unsigned long long m0 = stem7_low_bytes(hi < 8 ? hi : 8) & ~stem7_low_bytes(lo < 8 ? lo : 8);
    const unsigned long long m1 = stem7_low_bytes(hi > 8 ? hi - 8 : 0) & ~stem7_low_bytes(lo > 8 ? lo - 8 : 0);
    cmask[0] = (uint32_t)m0; cmask[1] = (uint32_t)(m0 >> 32); cmask[2] = (uint32_t)m1; cmask[3] = (uint32_t)(m1 >> 32);
  }
  const int img_base = b * a.cin * a.h * a.w;  // tensor < 2^31 bytes (host check)

  v4i bf[STEM7_KS][4];  // [K-step][pixel of the quad]: (lo, hi) of row 2h, (lo, hi) of row 2h + 1
#pragma unroll
  for (int ks = 0; ks < STEM7_KS; ++ks) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      // row cr = 4 ks + 2 h + rr -> (ci, filter row): both candidates are wave-uniform, the half picks one
      int ro[2], rm[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int cr = 4 * ks + 2 * hh + rr;
        const int ci = cr / 7, r7 = cr % 7;
        const int ih = 2 * oy - a.pt + r7;
        const bool rv = ih >= 0 && ih < a.h && cr < ncr;
        const int ihc = ih < 0 ? 0 : (ih >= a.h ? a.h - 1 : ih);
        const int cic = ci < a.cin ? ci : a.cin - 1;
        ro[hh] = img_base + (cic * a.h + ihc) * a.w;
        rm[hh] = rv ? -1 : 0;
      }
      const uint32_t rmask = (uint32_t)(h ? rm[1] : rm[0]);
      const int rowoff = h ? ro[1] : ro[0];
      uint32_t d[4];
      if (GUARD) {  // bytewise: nothing outside the row is touched
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t v = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int col = start + 4 * i + k;
            if (col >= 0 && col < a.w) v |= (uint32_t)(uint8_t)a.x[rowoff + col] << (8 * k);
          }
          d[i] = v & rmask;
        }
      } else {
        __builtin_memcpy(d, a.x + (rowoff + start), 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] &= cmask[i] & rmask;
      }
      // pixel j of the quad: window bytes 2j .. 2j + 7
      const uint32_t a01 = __builtin_amdgcn_alignbyte(d[1], d[0], 2), a12 = __builtin_amdgcn_alignbyte(d[2], d[1], 2),
                     a23 = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
      bf[ks][0][2 * rr] = (int)d[0]; bf[ks][0][2 * rr + 1] = (int)d[1];
      bf[ks][1][2 * rr] = (int)a01;  bf[ks][1][2 * rr + 1] = (int)a12;
      bf[ks][2][2 * rr] = (int)d[1]; bf[ks][2][2 * rr + 1] = (int)d[2];
      bf[ks][3][2 * rr] = (int)a12;  bf[ks][3][2 * rr + 1] = (int)a23;
    }
  }

  const int hw = oy * a.ow + 4 * xq;
  const int MT = (a.cout + 31) >> 5;
  for (int mt = 0; mt < MT; ++mt) {  // uniform
    v4i af[STEM7_KS];
#pragma unroll
    for (int ks = 0; ks < STEM7_KS; ++ks)
      af[ks] = *reinterpret_cast<const v4i*>(afrag + ((size_t)(mt * STEM7_KS + ks) * 64 + lane) * 16);
    if (mt > 0 && OUT != OUT_I32) stage_scale_bias<1, OUT>(g, mt, lane, lsb);
    v16i acc[1][4];
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0][j], zero, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < STEM7_KS; ++ks) acc[0][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks], bf[ks][j], acc[0][j], 0, 0, 0);
    }
    if (qvalid) {
      if (OUT == OUT_I32) {
        gemm_epilogue<1, OUT, VEC, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw);
      } else {
        switch (a.act) {
          case ACT_RELU: gemm_epilogue<1, OUT, VEC, MFULL, ACT_RELU>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
          case ACT_RELU6: gemm_epilogue<1, OUT, VEC, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
          case ACT_LEAKY: gemm_epilogue<1, OUT, VEC, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
          default: gemm_epilogue<1, OUT, VEC, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
        }
      }
    }
  }
}

template <int OUT, bool MFULL, bool VEC>
__global__ __launch_bounds__(256, 2) void conv7x7s2_stem_kernel(DirectS2Args a, const int8_t* __restrict__ afrag) {
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.y); PLHIP_PRELOAD(a.scale); PLHIP_PRELOAD(a.bias); PLHIP_PRELOAD(afrag);
  PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.cin); PLHIP_PRELOAD(a.h); PLHIP_PRELOAD(a.w); PLHIP_PRELOAD(a.cout); PLHIP_PRELOAD(a.oh);
  PLHIP_PRELOAD(a.ow); PLHIP_PRELOAD(a.pt); PLHIP_PRELOAD(a.pl); PLHIP_PRELOAD(a.act); PLHIP_PRELOAD(a.alpha);
  PLHIP_PRELOAD(a.res); PLHIP_PRELOAD(a.y2); PLHIP_PRELOAD(a.res_relu); PLHIP_PRELOAD(a.inv_scale2);
  __shared__ __attribute__((aligned(16))) float lsb_all[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* lsb = lsb_all[wave];
  // 1-D grid of 8 * per blocks; XCD x (= blockIdx % 8, round-robin dispatch) gets the x-th eighth of the (image, row group,
  // column tile) space, so that row groups sharing input rows sit on one L2.  All of it is wave-uniform.
  const int nx = ((a.ow >> 2) + 31) >> 5, ny = (a.oh + 3) >> 2;
  const unsigned nb = (unsigned)(nx * ny * a.n), per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;
  const int bx = (int)(vb % (unsigned)nx);
  const unsigned t = vb / (unsigned)nx;
  const int by = (int)(t % (unsigned)ny), bz = (int)(t / (unsigned)ny);
  // a 16-byte window starts up to 3 bytes before its row and ends up to 15 after it: only row groups of the first image that
  // reach input row 0 (or above) and row groups of the last image that reach its last row (or below) can touch bytes outside
  // the tensor: those fetch bytewise
  const bool guard = (bz == 0 && 8 * by - a.pt <= 0) || (bz + 1 == a.n && 8 * by + 12 - a.pt >= a.h - 1);
  if (guard) stem7_body<OUT, MFULL, true, VEC>(a, afrag, lsb, lane, wave, bx, by, bz);
  else stem7_body<OUT, MFULL, false, VEC>(a, afrag, lsb, lane, wave, bx, by, bz);
}

// vec_store: y (and the tail operands) aligned for a lane's 4 consecutive outputs (dword / 16-byte accesses)
void launch_conv7x7s2_stem(const DirectS2Args& a, int out, bool vec_store, hipStream_t s) {
  const int owq = a.ow >> 2;
  const long nblk = (long)((owq + 31) / 32) * ((a.oh + 3) / 4) * a.n;
  const dim3 blocks((unsigned)((nblk + 7) / 8 * 8));
  const int8_t* afrag = reinterpret_cast<const int8_t*>(a.wp);
  const bool mfull = a.cout % 32 == 0;
#define PLHIP_STEM7(O)                                                                                                   \
  do {                                                                                                                   \
    if (mfull && vec_store) hipLaunchKernelGGL((conv7x7s2_stem_kernel<O, true, true>), blocks, dim3(256), 0, s, a, afrag);   \
    else if (mfull) hipLaunchKernelGGL((conv7x7s2_stem_kernel<O, true, false>), blocks, dim3(256), 0, s, a, afrag);          \
    else if (vec_store) hipLaunchKernelGGL((conv7x7s2_stem_kernel<O, false, true>), blocks, dim3(256), 0, s, a, afrag);      \
    else hipLaunchKernelGGL((conv7x7s2_stem_kernel<O, false, false>), blocks, dim3(256), 0, s, a, afrag);                    \
  } while (0)
  if (out == OUT_I32) PLHIP_STEM7(OUT_I32);
  else if (out == OUT_F32) PLHIP_STEM7(OUT_F32);
  else PLHIP_STEM7(OUT_I8);
#undef PLHIP_STEM7
}

}  // namespace plhip
