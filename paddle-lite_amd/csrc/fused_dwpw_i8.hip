// fused_dwpw_i8.hip — depthwise 3x3 (int8 out) fused with the following pointwise 1x1 convolution.
//
// SURVEY.md §8(f) rank 1.  In the MobileNet graph every depthwise_conv2d [int8_out] feeds exactly one conv2d 1x1; run as
// two kernels the int8 intermediate makes a full HBM round trip (~40 % of the network's traffic) and costs a launch.  Here
// one workgroup computes, for a tile of <= 128 output pixels and ALL channels,
//   phase 1  D[k][n] = requant_dw( sum_{r,q} x[k][..] * wdw[k][r][q] )   (the reference's depthwise kernel,
//            conv_depthwise_3x3_int8_int8, lite/backends/arm/math/conv_impl.cc:909-1018, same arithmetic, same int8 result)
//            and writes each int8 value STRAIGHT INTO THE MFMA B-FRAGMENT it belongs to in LDS;
//   phase 2  Y[m][n] = epi_pw( sum_k Wpw[m][k] * D[k][n] )  — a pure v_mfma_i32_32x32x32_i8 loop: B fragments are
//            lane-linear ds_read_b128, A fragments come pre-packed from L2, no transposes, no barrier inside the loop.
// The result is bit-identical to running the two kernels one after the other (tests/test_gpu_fused.py).
//
// Tile: R = 32 / ceil(OW/4) output rows of the flattened (batch x OH) row space; column n = rl*4*OWQ + ox, so that the
// 32 column quads of the tile are the 32 MFMA lanes c and lane c ends with 4 consecutive ox for every output channel
// (one dword store per row, as in gemm_i8.hip).  LDS: K x 128 bytes of fragments (K = channels, <= 1024) + 20 B per
// channel of depthwise parameters + the pointwise scale/bias of the current pass.
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"
#include "gemm_epilogue.h"

namespace plhip {

template <int ACT>
__device__ __forceinline__ uint32_t fused_dw_requant(const int (&acc)[4], float sc, float bi, float alpha) {
  const float hi2 = ACT == ACT_RELU6 ? fminf(alpha + alpha, 254.f) : 254.f;
  const float lo2 = (ACT == ACT_RELU || ACT == ACT_RELU6) ? 0.f : -254.f;
  return dw_requant4<ACT>(acc, sc + sc, bi + bi, alpha, lo2, hi2);
}

// Phase 1 arithmetic for one channel k whose three row windows are already in registers: 4 depthwise outputs of this
// lane's quad -> 4 bytes into fragments i = 0..3.
template <int S, int ACT>
__device__ __forceinline__ void fused_dw_compute(const FusedArgs& a, uint8_t* frag, const uint32_t* prm, int k, int c,
                                                 const uint32_t (&in)[3][S == 1 ? 2 : 3]) {
  constexpr int ND = S == 1 ? 2 : 3;
  const uint32_t* p = prm + 5 * k;  // (w row0, w row1, w row2, scale, bias)
  int acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    uint32_t win[4];
    if (S == 1) {
      win[0] = in[r][0];
      win[1] = __builtin_amdgcn_alignbyte(in[r][1], in[r][0], 1);
      win[2] = __builtin_amdgcn_alignbyte(in[r][1], in[r][0], 2);
      win[3] = __builtin_amdgcn_alignbyte(in[r][1], in[r][0], 3);
    } else {
      win[0] = in[r][0];
      win[1] = __builtin_amdgcn_alignbyte(in[r][1], in[r][0], 2);
      win[2] = in[r][1];
      win[3] = __builtin_amdgcn_alignbyte(in[r][ND - 1], in[r][1], 2);
    }
    const int wr = (int)p[r];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_sdot4((int)win[j], wr, acc[j], false);
  }
  const uint32_t pk = fused_dw_requant<ACT>(acc, __uint_as_float(p[3]), __uint_as_float(p[4]), a.dw_alpha);
  // fragment (ks = k/32, i), lane (h = (k%32)/16, c), byte k%16
  uint8_t* dst = frag + (size_t)(k >> 5) * 4096 + ((((k >> 4) & 1) * 32 + c) << 4) + (k & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[i * 1024] = (uint8_t)(pk >> (8 * i));
}

// One lane = one column quad of the tile for every 8th channel.  FU channels are processed per round: all 3*FU row
// loads are issued before the first use, so a lane keeps 12 loads in flight (the loop is otherwise one memory latency
// per channel).
#define FUSED_FU 4
template <int S, bool TAIL, int ACT>
__device__ __forceinline__ void fused_phase1_act(const FusedArgs& a, uint8_t* frag, const uint32_t* prm, int c, int kslot, int b,
                                                 int iy0, int lcol, int sh, const uint32_t (&cmask)[S == 1 ? 2 : 3]) {
  constexpr int ND = S == 1 ? 2 : 3;
  const long hw_in = (long)a.h * a.w;
  const long tensor = (long)a.n * a.C * hw_in;
  for (int k0 = kslot; k0 < a.C; k0 += 8 * FUSED_FU) {
    uint32_t in[FUSED_FU][3][ND];
#pragma unroll
    for (int u = 0; u < FUSED_FU; ++u) {
      const int k = k0 + 8 * u < a.C ? k0 + 8 * u : a.C - 1;  // clamped: loaded, never used
      const long pbase = ((long)b * a.C + k) * hw_in;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        dw_load_row<ND, TAIL>(a.x + pbase, iy0 + r, a.h, a.w, lcol, sh, tensor - pbase, cmask, in[u][r]);
    }
#pragma unroll
    for (int u = 0; u < FUSED_FU; ++u)
      if (k0 + 8 * u < a.C) fused_dw_compute<S, ACT>(a, frag, prm, k0 + 8 * u, c, in[u]);
  }
}

template <int S, bool TAIL>
__device__ __forceinline__ void fused_phase1(const FusedArgs& a, uint8_t* frag, const uint32_t* prm, long gr0) {
  constexpr int ND = S == 1 ? 2 : 3;
  const int c = threadIdx.x & 31;       // column quad of the tile == MFMA lane c
  const int kslot = threadIdx.x >> 5;   // 8 channel slots
  const int owq = (a.ow + 3) >> 2;
  const int rl = c / owq, xq = c - rl * owq;
  const long gr = gr0 + rl;             // flattened (image, output row)
  const bool live = c < a.R * owq && gr < (long)a.n * a.oh;
  if (!live) return;                    // dead quads: their fragment bytes are never stored
  const int b = (int)(gr / a.oh), oy = (int)(gr - (long)b * a.oh);
  const int iy0 = oy * S - a.pt;
  const int start = 4 * xq * S - a.pl;
  const int sh = start < 0 ? -start : 0;
  int lcol = start + sh;
  if (lcol > a.w - 1) lcol = a.w - 1;
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
  switch (a.dw_act) {  // block-uniform
    case ACT_RELU: fused_phase1_act<S, TAIL, ACT_RELU>(a, frag, prm, c, kslot, b, iy0, lcol, sh, cmask); break;
    case ACT_RELU6: fused_phase1_act<S, TAIL, ACT_RELU6>(a, frag, prm, c, kslot, b, iy0, lcol, sh, cmask); break;
    case ACT_LEAKY: fused_phase1_act<S, TAIL, ACT_LEAKY>(a, frag, prm, c, kslot, b, iy0, lcol, sh, cmask); break;
    default: fused_phase1_act<S, TAIL, ACT_NONE>(a, frag, prm, c, kslot, b, iy0, lcol, sh, cmask); break;
  }
}

template <int MA, int OUT, int S>
__global__ __launch_bounds__(256, 2) void fused_dwpw_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];  // ONE LDS object: [fragments][dw params][pw scale/bias]
  const int KS = (a.C + 31) >> 5;
  uint8_t* frag = fsm;                                           // KS * 4 KiB
  uint32_t* prm = reinterpret_cast<uint32_t*>(fsm + (size_t)KS * 4096);  // C * 5 dwords
  float* lsb_all = reinterpret_cast<float*>(prm + ((5 * a.C + 3) & ~3));   // 4 waves x 2*MA*32 floats
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const long gr0 = (long)blockIdx.x * a.R;

  // ---- depthwise parameters -> LDS: per channel (w0w1w2 | w3w4w5 | w6w7w8 | scale | bias) ----
  for (int k = threadIdx.x; k < a.C; k += 256) {
    const int8_t* wp = a.dw_w + (size_t)k * 9;
    uint32_t w0, w1, w2;
    __builtin_memcpy(&w0, wp, 4);
    __builtin_memcpy(&w1, wp + 3, 4);
    __builtin_memcpy(&w2, wp + 5, 4);  // one byte early + shift: never reads past the filter tensor
    prm[5 * k + 0] = w0 & 0xffffffu;
    prm[5 * k + 1] = w1 & 0xffffffu;
    prm[5 * k + 2] = w2 >> 8;
    prm[5 * k + 3] = __float_as_uint(a.dw_scale[k]);
    prm[5 * k + 4] = __float_as_uint(a.dw_bias ? a.dw_bias[k] : 0.f);
  }
  __syncthreads();

  // ---- phase 1: depthwise, results land in MFMA B-fragment order ----
  if (!(a.pw.dbg & 1)) {
    if (blockIdx.x + 1 == gridDim.x) fused_phase1<S, true>(a, frag, prm, gr0);
    else fused_phase1<S, false>(a, frag, prm, gr0);
  }
  __syncthreads();

  // ---- phase 2: pointwise GEMM, MT tiles of 32*MA rows, 4 per pass ----
  const int owq = (a.ow + 3) >> 2;
  const int rl = c / owq, xq = c - rl * owq;
  const long gr = gr0 + rl;
  const bool cvalid = c < a.R * owq && gr < (long)a.n * a.oh;
  const int b = cvalid ? (int)(gr / a.oh) : 0;
  const int oy = cvalid ? (int)(gr - (long)b * a.oh) : 0;
  const int hw = oy * a.ow + 4 * xq;
  const int room = cvalid ? a.ow - 4 * xq : 0;
  const GemmArgs& g = a.pw;
  const int MT = (g.M + 32 * MA - 1) / (32 * MA);
  float* lsb = lsb_all + wave * 2 * MA * 32;
  const v4i* bfrag = reinterpret_cast<const v4i*>(frag) + lane;

  for (int pass = 0; pass * 4 < MT; ++pass) {
    const int mt = pass * 4 + wave;
    if (mt >= MT) break;  // wave-uniform; no barrier below
    float my_s = 1.f, my_b = 0.f;
    v4i a_cur[MA], a_nxt[MA];
    load_a<MA>(g.wp, mt, KS, 0, lane, a_cur);
    if (OUT != OUT_I32) load_scale_bias<MA>(g, mt, lane, my_s, my_b);
    v16i acc[MA][4];
#pragma unroll
    for (int aa = 0; aa < MA; ++aa)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[aa][i][r] = 0;
    v4i b_cur[4], b_nxt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b_cur[i] = bfrag[i * 64];
    for (int ks = 0; ks < ((a.pw.dbg & 2) ? 0 : KS); ++ks) {
      const int kn = ks + 1 < KS ? ks + 1 : ks;
      load_a<MA>(g.wp, mt, KS, kn, lane, a_nxt);
#pragma unroll
      for (int i = 0; i < 4; ++i) b_nxt[i] = bfrag[(kn * 4 + i) * 64];
#pragma unroll
      for (int aa = 0; aa < MA; ++aa)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[aa][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[aa], b_cur[i], acc[aa][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) b_cur[i] = b_nxt[i];
#pragma unroll
      for (int aa = 0; aa < MA; ++aa) a_cur[aa] = a_nxt[aa];
    }
    if (OUT != OUT_I32) store_scale_bias<MA, OUT>(lsb, lane, my_s, my_b);
    if (!cvalid || (a.pw.dbg & 4)) continue;
    if (OUT == OUT_I32) {
      gemm_epilogue<MA, OUT, false, false, ACT_NONE>(g, acc, mt, h, b, hw, lsb, room);
    } else {
      switch (g.act) {
        case ACT_RELU: gemm_epilogue<MA, OUT, false, false, ACT_RELU>(g, acc, mt, h, b, hw, lsb, room); break;
        case ACT_RELU6: gemm_epilogue<MA, OUT, false, false, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, room); break;
        case ACT_LEAKY: gemm_epilogue<MA, OUT, false, false, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, room); break;
        default: gemm_epilogue<MA, OUT, false, false, ACT_NONE>(g, acc, mt, h, b, hw, lsb, room); break;
      }
    }
  }
}

size_t fused_dwpw_lds_bytes(int C, int ma) {
  const size_t KS = (C + 31) / 32;
  return KS * 4096 + (size_t)((5 * C + 3) & ~3) * 4 + (size_t)4 * 2 * ma * 32 * 4;
}

bool fused_dwpw_supported(int C, int kh, int kw, int sh, int sw, int dh, int dw, int pl, int ow) {
  return kh == 3 && kw == 3 && sh == sw && (sw == 1 || sw == 2) && dh == 1 && dw == 1 && pl <= 3 && ow <= 128 &&
         fused_dwpw_lds_bytes(C, 2) <= 160 * 1024;
}

template <int MA, int OUT>
static void launch_fused_t(const FusedArgs& a, unsigned blocks, size_t lds, hipStream_t s) {
  if (a.stride == 1) {
    auto kfn = fused_dwpw_kernel<MA, OUT, 1>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, s, a);
  } else {
    auto kfn = fused_dwpw_kernel<MA, OUT, 2>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, s, a);
  }
}

void launch_fused_dwpw(const FusedArgs& a_in, int out, hipStream_t s) {
  FusedArgs a = a_in;
  static int dbg_env = -1;
  if (dbg_env < 0) {
    const char* e = getenv("PLHIP_FUSED_DEBUG");
    dbg_env = e ? atoi(e) : 0;
  }
  a.pw.dbg = dbg_env;
  const int owq = (a.ow + 3) / 4;
  a.R = 32 / owq;
  const long rows = (long)a.n * a.oh;
  const unsigned blocks = (unsigned)((rows + a.R - 1) / a.R);
  const int ma = a.pw.M > 128 ? 2 : 1;
  const size_t lds = fused_dwpw_lds_bytes(a.C, ma);
  if (ma == 1) {
    if (out == OUT_I32) launch_fused_t<1, OUT_I32>(a, blocks, lds, s);
    else if (out == OUT_F32) launch_fused_t<1, OUT_F32>(a, blocks, lds, s);
    else launch_fused_t<1, OUT_I8>(a, blocks, lds, s);
  } else {
    if (out == OUT_I32) launch_fused_t<2, OUT_I32>(a, blocks, lds, s);
    else if (out == OUT_F32) launch_fused_t<2, OUT_F32>(a, blocks, lds, s);
    else launch_fused_t<2, OUT_I8>(a, blocks, lds, s);
  }
}

}  // namespace plhip
