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
#include "plhip_device.h"
#include "plhip_kernels.h"

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
