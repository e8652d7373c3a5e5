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
#include "plhip_device.h"
#include "plhip_kernels.h"

namespace plhip {

#define DS2_COB 16   // output channels per accumulation pass (64 accumulators)
#define DS2_MAXCIN 4

bool conv3x3s2_direct_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int pl) {
  return groups == 1 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && dh == 1 && dw == 1 && cin <= DS2_MAXCIN && cout <= 128 &&
         pl <= 3;
}

size_t conv3x3s2_direct_packed_bytes(int cin, int cout) { return (size_t)cin * 3 * ((cout + 3) / 4 * 4) * 4; }

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
      uint32_t t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (uint32_t)__builtin_amdgcn_fmed3f(__fmaf_rn((float)acc[j], s2, b2), lo2, hi2);
      const uint32_t p = (t[0] | (t[1] << 8)) | ((t[2] | (t[3] << 8)) << 16);
      pk = ((p + 0x01010101u) >> 1) & 0x7f7f7f7fu;
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
__device__ __forceinline__ void ds2_body(const DirectS2Args& a, const uint32_t* lw, const uint32_t (&win)[DS2_MAXCIN * 3][4],
                                         int b, int oy, int xq) {
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
      const float s = (OUT == OUT_I32) ? 1.f : a.scale[co];
      const float bi = (OUT != OUT_I32 && a.bias) ? a.bias[co] : 0.f;
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
    ds2_body<OUT, ACT_NONE>(a, lw, win, b, oy, xq);
    return;
  }
  switch (a.act) {
    case ACT_RELU: ds2_body<OUT, ACT_RELU>(a, lw, win, b, oy, xq); break;
    case ACT_RELU6: ds2_body<OUT, ACT_RELU6>(a, lw, win, b, oy, xq); break;
    case ACT_LEAKY: ds2_body<OUT, ACT_LEAKY>(a, lw, win, b, oy, xq); break;
    default: ds2_body<OUT, ACT_NONE>(a, lw, win, b, oy, xq); break;
  }
}

void launch_pack_conv3x3s2_direct(const int8_t* w_oihw, uint32_t* wp, int cin, int cout, hipStream_t s) {
  const int coutp = (cout + 3) / 4 * 4;
  const int total = cin * 3 * coutp;
  hipLaunchKernelGGL(pack_conv3x3s2_direct_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_oihw, wp, cin, cout, coutp);
}

void launch_conv3x3s2_direct(const DirectS2Args& a, int out, hipStream_t s) {
  const long owq = (a.ow + 3) >> 2;
  const long total = (long)a.n * a.oh * owq;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  const size_t lds = ((size_t)a.cin * 3 * a.coutp + DS2_COB) * 4;
  if (out == OUT_I32) hipLaunchKernelGGL((conv3x3s2_direct_kernel<OUT_I32>), dim3(blocks), dim3(256), lds, s, a);
  else if (out == OUT_F32) hipLaunchKernelGGL((conv3x3s2_direct_kernel<OUT_F32>), dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv3x3s2_direct_kernel<OUT_I8>), dim3(blocks), dim3(256), lds, s, a);
}

}  // namespace plhip
