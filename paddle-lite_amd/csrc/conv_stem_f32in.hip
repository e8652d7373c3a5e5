// conv_stem_f32in.hip — calib[fp32_to_int8] + conv 3x3 stride 2 (Cin <= 3) in ONE launch: the first two instructions of the
// MobileNet programs (the fp32 image is quantised with the conv's input scale, then convolved).  As two kernels the int8 image is
// written and read back (19 + 19 MB of the 147 MB the pair moves at batch 128) and the calib launch is a pure stream at the HBM
// roof (16 us) in front of a stem that cannot start before it ends.
// Replaces the instruction pair CalibComputeFp32ToInt8 (lite/kernels/arm/calib_compute.cc:25-40 -> type_trans.cc:34-187) ;
// DirectConv<kInt8,*> (lite/kernels/arm/conv_direct.cc -> conv3x3s2_direct_int8.cc); results bit-identical to the two kernels:
// every input value is quantised exactly as calib_f32_to_i8_kernel does (round_sat_i8(inv_scale * x)), the conv is the MFMA
// form of conv_direct_i8.hip (same A fragments, same epilogue).
//
// block = (image, 4 output rows, 32 column quads = 128 output pixels): the 9 input rows x Cin channels x 264 columns it reads are
// fetched ONCE as aligned 16-byte pieces of fp32, quantised and written to LDS ([Cin * 9 rows][272 B], zeros where the padding
// is: no masks later); a wave then owns one output row: lane (c, h) reads its five row windows as 16 aligned LDS bytes each and
// cuts the four 3-byte windows with v_alignbyte (the window of quad c starts at LDS byte 8 c + 3 for the left padding 1).
#include "gemm_epilogue.h"
#include "plhip_kernels.h"
#include "gemm_tr_common.h"

namespace plhip {

constexpr int SF_PITCH = 272;  // 4 bytes in front of the tile's first column, 256 columns, 12 behind

bool conv3x3s2_f32in_supported(const DirectS2Args& a) {
  return a.cin >= 1 && a.cin <= 3 && (a.ow & 3) == 0 && (a.w & 3) == 0 && a.pl == 1 && (a.pt == 0 || a.pt == 1) && a.cout <= 128 &&
         (long)a.n * a.cin * a.h * a.w < (1L << 31) && (long)(((a.ow >> 2) + 31) / 32) * ((a.oh + 3) / 4) * a.n < (1L << 31) - 8;
}

template <int OUT, bool MFULL>
__global__ __launch_bounds__(256) void conv3x3s2_mfma_f32in_kernel(DirectS2Args a, const int8_t* __restrict__ afrag) {
  PLHIP_PRELOAD(a.xf); PLHIP_PRELOAD(a.y); PLHIP_PRELOAD(a.scale); PLHIP_PRELOAD(a.bias); PLHIP_PRELOAD(afrag);
  PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.cin); PLHIP_PRELOAD(a.h); PLHIP_PRELOAD(a.w); PLHIP_PRELOAD(a.cout); PLHIP_PRELOAD(a.oh);
  PLHIP_PRELOAD(a.ow); PLHIP_PRELOAD(a.pt); PLHIP_PRELOAD(a.act); PLHIP_PRELOAD(a.alpha); PLHIP_PRELOAD(a.x_inv_scale);
  __shared__ __attribute__((aligned(16))) float lsb_all[4][64];
  __shared__ __attribute__((aligned(16))) uint8_t img[27 * SF_PITCH];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* lsb = lsb_all[wave];
  const int nx = ((a.ow >> 2) + 31) >> 5, ny = (a.oh + 3) >> 2;
  const unsigned nb = (unsigned)(nx * ny * a.n), per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;  // block-uniform
  const int bx = (int)(vb % (unsigned)nx);
  const unsigned tq = vb / (unsigned)nx;
  const int by = (int)(tq % (unsigned)ny), b = (int)(tq / (unsigned)ny);

  // ---- stage: input rows 8 by - pt + j (j = 0..8) of every channel, columns 256 bx - 4 .. 256 bx + 259, as int8
  const int c0 = 256 * bx - 4, ih0 = 8 * by - a.pt;
  const int nrows = a.cin * 9;
  const float inv = a.x_inv_scale;
  const size_t img_base = (size_t)b * a.cin * a.h * a.w;
  // (all of a thread's pieces are requested before the first one is converted: fetched and converted one by one the block paid
  // one memory round trip per piece, 7 in a row: 33.8 us for the launch)
  constexpr int NPIECE = (27 * 66 + 255) / 256;
  v4f pv[NPIECE];
  int pdst[NPIECE];
#pragma unroll
  for (int it = 0; it < NPIECE; ++it) {
    const int k = it * 256 + (int)threadIdx.x;
    const int row = k / 66, piece = k - row * 66;
    const int ci = row / 9, j = row - ci * 9;
    const int ih = ih0 + j, col = c0 + 4 * piece;
    pdst[it] = k < nrows * 66 ? row * SF_PITCH + 4 * piece : -1;
    pv[it] = v4f{0.f, 0.f, 0.f, 0.f};
    if (k < nrows * 66 && ih >= 0 && ih < a.h && col >= 0 && col < a.w)  // (w % 4 == 0: a piece is inside or outside as a whole)
      pv[it] = *reinterpret_cast<const v4f*>(a.xf + img_base + ((size_t)ci * a.h + ih) * a.w + col);
  }
#pragma unroll
  for (int it = 0; it < NPIECE; ++it) {
    const v4f v = pv[it];
    const uint32_t pk = pack4_i8(round_sat_i8(inv * v[0]), round_sat_i8(inv * v[1]), round_sat_i8(inv * v[2]), round_sat_i8(inv * v[3]));
    if (pdst[it] >= 0) *reinterpret_cast<uint32_t*>(img + pdst[it]) = pk;  // (a piece outside the image: +0.0 -> 0, the padding)
  }
  // (bytes 264 .. 271 of a row are read by the last quads' 16-byte windows and never used: no need to clear them)

  const int c = lane & 31, h = lane >> 5;
  const int owq = a.ow >> 2;
  const int oy = by * 4 + wave;
  int xq = bx * 32 + c;
  const bool qvalid = xq < owq && oy < a.oh;
  if (xq >= owq) xq = owq - 1;

  GemmArgs g;
  g.y = a.y;
  g.scale = a.scale;
  g.bias = a.bias;
  g.M = a.cout;
  g.HWY = a.oh * a.ow;
  g.y_bstride = (size_t)a.cout * a.oh * a.ow;
  g.act = a.act;
  g.alpha = a.alpha;
  g.res = nullptr; g.res_relu = 0; g.y2 = nullptr; g.inv_scale2 = 0.f;
  const v4i af0 = *reinterpret_cast<const v4i*>(afrag + (size_t)lane * 16);
  if (OUT != OUT_I32) stage_scale_bias<1, OUT>(g, 0, lane, lsb);
  __syncthreads();

  // ---- operands: window cr = 5 h + L -> (channel cr / 3, filter row cr % 3) = staged row 9 ci + 2 wave + r3
  uint32_t win[4][5];
#pragma unroll
  for (int L = 0; L < 5; ++L) {
    const int cr = 5 * h + L;
    const int ci = cr / 3, r3 = cr - ci * 3;
    const bool live = cr < nrows / 3;  // (nrows / 3 = 3 Cin windows exist)
    const int lrow = live ? ci * 9 + 2 * wave + r3 : 0;
    const uint8_t* p = img + lrow * SF_PITCH + 8 * c;
    const v2i lo = *reinterpret_cast<const v2i*>(p), hi = *reinterpret_cast<const v2i*>(p + 8);
    const uint32_t m = live ? 0xffffffffu : 0u;
    const uint32_t d0 = (uint32_t)lo[0] & m, d1 = (uint32_t)lo[1] & m, d2 = (uint32_t)hi[0] & m, d3 = (uint32_t)hi[1] & m;
    win[0][L] = __builtin_amdgcn_alignbyte(d1, d0, 3);  // columns 8 xq - 1 ..
    win[1][L] = __builtin_amdgcn_alignbyte(d2, d1, 1);  // 8 xq + 1 ..
    win[2][L] = __builtin_amdgcn_alignbyte(d2, d1, 3);  // 8 xq + 3 ..
    win[3][L] = __builtin_amdgcn_alignbyte(d3, d2, 1);  // 8 xq + 5 ..
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

// a.xf = the fp32 image, a.x_inv_scale = 1 / calib scale; a.wp = the packed block of launch_pack_conv3x3s2_direct
void launch_conv3x3s2_f32in(const DirectS2Args& a, const int8_t* afrag, int out, hipStream_t s) {
  const int owq = a.ow >> 2;
  const long nblk = (long)((owq + 31) / 32) * ((a.oh + 3) / 4) * a.n;
  const dim3 blocks((unsigned)((nblk + 7) / 8 * 8));
  const bool mfull = a.cout % 32 == 0;
#define PLHIP_STEMF(O)                                                                                          \
  do {                                                                                                          \
    if (mfull) hipLaunchKernelGGL((conv3x3s2_mfma_f32in_kernel<O, true>), blocks, dim3(256), 0, s, a, afrag);   \
    else hipLaunchKernelGGL((conv3x3s2_mfma_f32in_kernel<O, false>), blocks, dim3(256), 0, s, a, afrag);        \
  } while (0)
  if (out == OUT_I32) PLHIP_STEMF(OUT_I32);
  else if (out == OUT_F32) PLHIP_STEMF(OUT_F32);
  else PLHIP_STEMF(OUT_I8);
#undef PLHIP_STEMF
}

}  // namespace plhip
