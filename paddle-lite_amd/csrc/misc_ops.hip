// misc_ops.hip — fc (int8), calib (fp32<->int8), global average pool and softmax (fp32) for gfx950.
//
// Replaces (reference, ARM):
//   fc      FcCompute<kInt8,*>::Run   lite/kernels/arm/fc_compute.cc:229-344 -> gemm_s8 / gemv_int8
//   calib   CalibCompute*             lite/kernels/arm/calib_compute.cc:25-57 -> type_trans.cc:34-187, 268-371
//   pool    pooling_global_avg        lite/backends/arm/math/pooling.cc:1006-
//   softmax softmax_inner1            lite/backends/arm/math/softmax.cc
// All four are tiny in the MobileNet graph (FC = 1 MMAC / image); they are kept on device so that the
// whole graph runs without host round trips.  FC uses v_dot4_i32_i8 on a [k/4][n][4] pre-packed weight:
// lanes walk n (coalesced dword per k-quad), the x dwords are wave-uniform.
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"

namespace plhip {

// FC epilogue.  flags bit 0: relu; bit 1: the reference's gemm_s8 + fill_bias_fc route (fc_compute.cc:250-266,
// funcs.cc:24-108: product rounded, then the bias added with a second rounding) instead of the single fused
// multiply-add of its gemv route (gemv_arm_int8.cc:47-56).  The host picks the route like check_fc_use_gemm does.
__device__ __forceinline__ float mul_then_add_two_roundings(float a, float s, float b) {
#pragma clang fp contract(off)  // HIP's __fmul_rn / __fadd_rn are plain operators: hipcc's default contraction fuses them
  const float p = a * s;
  return p + b;
}
__device__ __forceinline__ float fc_epilogue_f32(int acc, float s, float b, int flags) {
  float y = (flags & 2) ? mul_then_add_two_roundings((float)acc, s, b) : __fmaf_rn((float)acc, s, b);
  if (flags & 1) y = y > 0.f ? y : 0.f;
  return y;
}

// w [k][n] -> wp [(k+3)/4][n][4]  (zero padded in k)
__global__ void pack_fc_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, int k, int n) {
  const int k4n = (k + 3) / 4;
  const size_t total = (size_t)k4n * n * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 3;
    const size_t t = idx >> 2;
    const int col = (int)(t % n);
    const int kq = (int)(t / n);
    const int kk = kq * 4 + j;
    wp[idx] = kk < k ? w[(size_t)kk * n + col] : (int8_t)0;
  }
}

// One thread: one output column n for FC_MB consecutive rows m.  x rows must be readable as dwords:
// the tail quad of a row (k % 4 != 0) is assembled bytewise.
#define FC_MB 8
template <int OUT>
__global__ __launch_bounds__(256) void fc_i8_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ wp,
                                                    const float* __restrict__ scale, const float* __restrict__ bias,
                                                    void* __restrict__ y, int m, int k, int n, int relu) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * FC_MB;
  const int k4n = (k + 3) / 4;
  int acc[FC_MB];
#pragma unroll
  for (int i = 0; i < FC_MB; ++i) acc[i] = 0;
  const int colc = col < n ? col : n - 1;
  const uint32_t* wq = reinterpret_cast<const uint32_t*>(wp) + colc;
  for (int kq = 0; kq < k4n; ++kq) {
    const int wv = (int)wq[(size_t)kq * n];
#pragma unroll
    for (int i = 0; i < FC_MB; ++i) {
      const int mi = m0 + i < m ? m0 + i : m - 1;
      const int8_t* xr = x + (size_t)mi * k + kq * 4;
      uint32_t xv = 0;
      if (kq * 4 + 3 < k) {
        __builtin_memcpy(&xv, xr, 4);
      } else {
        for (int j = 0; j < 4; ++j)
          if (kq * 4 + j < k) xv |= (uint32_t)(uint8_t)xr[j] << (8 * j);
      }
      acc[i] = __builtin_amdgcn_sdot4((int)xv, wv, acc[i], false);
    }
  }
  if (col >= n) return;
  const float s = (OUT == OUT_I32) ? 1.f : scale[col];
  const float bi = (OUT != OUT_I32 && bias) ? bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < FC_MB; ++i) {
    if (m0 + i >= m) break;
    const size_t off = (size_t)(m0 + i) * n + col;
    if (OUT == OUT_I32) {
      reinterpret_cast<int*>(y)[off] = acc[i];
    } else {
      const float f = fc_epilogue_f32(acc[i], s, bi, relu);
      if (OUT == OUT_F32) reinterpret_cast<float*>(y)[off] = f;
      else reinterpret_cast<int8_t*>(y)[off] = (int8_t)round_sat_i8(f);
    }
  }
}

// Fast path (k % 16 == 0, x 16-byte aligned): a 256-thread block owns 64 output columns x FCF_MB rows.  The block
// first stages its FCF_MB x-rows into LDS with coalesced 16-byte loads (scalar loads of x were latency bound: 16
// dependent s_load per round).  Its 4 waves then split the k range; per round a lane reads 4 packed weight dwords
// (coalesced: lanes walk n) and each x quad-quad comes from LDS as one broadcast ds_read_b128.  Partial sums of the 4
// waves meet in LDS (the x tile's space is reused after a barrier).
#define FCF_MB 16
template <int OUT>
__global__ __launch_bounds__(256) void fc_i8_fast_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ wp,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         void* __restrict__ y, int m, int k, int n, int relu) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];  // max(FCF_MB * k, 4 * FCF_MB * 64 * 4) bytes
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * FCF_MB;
  const int k4n = k >> 2;
  const int k16n = k >> 4;
  // ---- stage x rows m0 .. m0+FCF_MB-1 ----
  {
    const v4i* src = reinterpret_cast<const v4i*>(x);
    v4i* dst = reinterpret_cast<v4i*>(fsm);
    for (int i = threadIdx.x; i < FCF_MB * k16n; i += 256) {
      const int r = i / k16n, c16 = i - r * k16n;
      const int mi = m0 + r < m ? m0 + r : m - 1;
      dst[i] = src[(size_t)mi * k16n + c16];
    }
  }
  __syncthreads();
  const int g0 = (k16n * wave) / 4, g1 = (k16n * (wave + 1)) / 4;  // this wave's 16-byte k groups
  const int colc = col < n ? col : n - 1;
  const uint32_t* wq = reinterpret_cast<const uint32_t*>(wp) + colc;
  int acc[FCF_MB];
#pragma unroll
  for (int i = 0; i < FCF_MB; ++i) acc[i] = 0;
  for (int gq = g0; gq < g1; ++gq) {
    int wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = (int)wq[(size_t)(gq * 4 + u) * n];
#pragma unroll
    for (int i = 0; i < FCF_MB; ++i) {
      const v4i xv = *reinterpret_cast<const v4i*>(fsm + ((size_t)i * k16n + gq) * 16);  // same address in every lane
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[i] = __builtin_amdgcn_sdot4(xv[u], wv[u], acc[i], false);
    }
  }
  __syncthreads();  // everyone is done with the x tile: reuse the space for the cross-wave reduction
  int* red = reinterpret_cast<int*>(fsm);
#pragma unroll
  for (int i = 0; i < FCF_MB; ++i) red[(wave * FCF_MB + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (col >= n) return;
  const float s = (OUT == OUT_I32) ? 1.f : scale[col];
  const float bi = (OUT != OUT_I32 && bias) ? bias[col] : 0.f;
#pragma unroll
  for (int j = 0; j < FCF_MB / 4; ++j) {
    const int i = wave * (FCF_MB / 4) + j;
    if (m0 + i >= m) break;
    const int a = red[(0 * FCF_MB + i) * 64 + lane] + red[(1 * FCF_MB + i) * 64 + lane] + red[(2 * FCF_MB + i) * 64 + lane] +
                  red[(3 * FCF_MB + i) * 64 + lane];
    const size_t off = (size_t)(m0 + i) * n + col;
    if (OUT == OUT_I32) {
      reinterpret_cast<int*>(y)[off] = a;
    } else {
      const float f = fc_epilogue_f32(a, s, bi, relu);
      if (OUT == OUT_F32) reinterpret_cast<float*>(y)[off] = f;
      else reinterpret_cast<int8_t*>(y)[off] = (int8_t)round_sat_i8(f);
    }
  }
}

// ---- MFMA path (k % 32 == 0): Y^T tile = W^T (32 output features x K) * X^T (K x 32 batch rows).  Both operands are
// K-contiguous per lane as v_mfma_i32_32x32x32_i8 wants them: A = the pre-packed weights (second half of the packed
// block, fragment order), B = 16 consecutive bytes of an x row -- no transposes anywhere.  A lane ends with 4
// consecutive output features of one batch row: one 16-byte fp32 store.  The 4 waves of a block split K (all their
// loads are issued before the first MFMA: the op is one memory round trip long) and meet in LDS.  The dot4 kernel
// above walked K with one dependent global load per 16 k: 16 serial round trips, 15 us for 131 MMAC.
size_t fc_dot4_bytes(int k, int n) { return (((size_t)((k + 3) / 4) * n * 4) + 15) & ~(size_t)15; }
size_t fc_packed_bytes(int k, int n) { return fc_dot4_bytes(k, n) + (size_t)((n + 31) / 32) * ((k + 31) / 32) * 1024; }

__global__ void pack_fc_mfma_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, int k, int n) {
  const int KS = (k + 31) / 32;
  const size_t total = (size_t)((n + 31) / 32) * KS * 1024;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 15, lane = (idx >> 4) & 63;
    const size_t t = idx >> 10;
    const int ks = (int)(t % KS), nt = (int)(t / KS);
    const int nn = nt * 32 + (lane & 31), kk = ks * 32 + 16 * (lane >> 5) + j;
    wp[idx] = (nn < n && kk < k) ? w[(size_t)kk * n + nn] : (int8_t)0;
  }
}

template <int OUT>
__global__ __launch_bounds__(256) void fc_i8_mfma_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ wfrag,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         void* __restrict__ y, int m, int k, int n, int relu) {
  __shared__ int red[4][16][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5;
  const int nt = blockIdx.x, mt = blockIdx.y;
  const int KS = k >> 5;
  const int ks0 = (KS * wave) >> 2, ks1 = (KS * (wave + 1)) >> 2;
  const int mrow = mt * 32 + c < m ? mt * 32 + c : m - 1;
  const int8_t* xb = x + (size_t)mrow * k + 16 * h;
  const int8_t* ab = wfrag + (size_t)nt * KS * 1024 + lane * 16;
  const int n0 = nt * 32 + 8 * wave + 4 * h;  // the 4 features this lane finishes (register group = wave)
  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  for (int ks = ks0; ks < ks1; ks += 8) {
    v4i af[8], bf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = ks + u < ks1 ? ks + u : ks1 - 1;  // surplus slots repeat the last step and are not multiplied
      af[u] = *reinterpret_cast<const v4i*>(ab + (size_t)kk * 1024);
      __builtin_memcpy(&bf[u], xb + (size_t)kk * 32, 16);  // x rows need no alignment
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (ks + u < ks1) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[u], bf[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // wave w finishes register group w: rows (features) nt*32 + 8w + 4h + (0..3), column (batch row) mt*32 + c
  const int mcol = mt * 32 + c;
  if (mcol >= m || n0 >= n) return;
  // scale / bias: 8 independent loads issued together (hoisting them above the operand loads measured 6 us slower)
  float sc4[4] = {1.f, 1.f, 1.f, 1.f}, bi4[4] = {0.f, 0.f, 0.f, 0.f};
  if (OUT != OUT_I32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sc4[e] = scale[n0 + e < n ? n0 + e : n - 1];
    if (bias) {  // uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) bi4[e] = bias[n0 + e < n ? n0 + e : n - 1];
    }
  }
  float f[4];
  int a4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * wave + e;
    a4[e] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    f[e] = fc_epilogue_f32(a4[e], sc4[e], bi4[e], relu);  // fp32-out spec
  }
  const size_t off = (size_t)mcol * n + n0;
  const int cnt = n - n0 < 4 ? n - n0 : 4;
  if (OUT == OUT_I32) {
    int* yp = reinterpret_cast<int*>(y) + off;
    if (cnt == 4) __builtin_memcpy(yp, a4, 16);
    else for (int e = 0; e < cnt; ++e) yp[e] = a4[e];
  } else if (OUT == OUT_F32) {
    float* yp = reinterpret_cast<float*>(y) + off;
    if (cnt == 4) __builtin_memcpy(yp, f, 16);
    else for (int e = 0; e < cnt; ++e) yp[e] = f[e];
  } else {
    int8_t* yp = reinterpret_cast<int8_t*>(y) + off;
    const uint32_t pk = pack4_i8(round_sat_i8(f[0]), round_sat_i8(f[1]), round_sat_i8(f[2]), round_sat_i8(f[3]));
    if (cnt == 4) __builtin_memcpy(yp, &pk, 4);
    else for (int e = 0; e < cnt; ++e) yp[e] = (int8_t)(pk >> (8 * e));
  }
}

// q = clamp(round_half_away(x * (1.f/scale)), -127, 127)   type_trans.cc:45,183-184
__global__ void calib_f32_to_i8_kernel(const float* __restrict__ x, int8_t* __restrict__ y, float inv_scale, int64_t count, int vec) {
  const int64_t nq = vec ? count >> 2 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
    const v4f v = reinterpret_cast<const v4f*>(x)[i];
    reinterpret_cast<uint32_t*>(y)[i] = pack4_i8(round_sat_i8(inv_scale * v[0]), round_sat_i8(inv_scale * v[1]),
                                                 round_sat_i8(inv_scale * v[2]), round_sat_i8(inv_scale * v[3]));
  }
  for (int64_t t = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride)
    y[t] = (int8_t)round_sat_i8(inv_scale * x[t]);
}

// x = q * scale   type_trans.cc:268-371
__global__ void calib_i8_to_f32_kernel(const int8_t* __restrict__ x, float* __restrict__ y, float scale, int64_t count, int vec) {
  const int64_t nq = vec ? count >> 2 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
    const uint32_t v = reinterpret_cast<const uint32_t*>(x)[i];
    v4f o;
    o[0] = scale * (float)(int8_t)(v & 0xff);
    o[1] = scale * (float)(int8_t)((v >> 8) & 0xff);
    o[2] = scale * (float)(int8_t)((v >> 16) & 0xff);
    o[3] = scale * (float)(int8_t)(v >> 24);
    reinterpret_cast<v4f*>(y)[i] = o;
  }
  for (int64_t t = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride)
    y[t] = scale * (float)x[t];
}

// 16 lanes per (n, c) plane, 16 planes per 256-thread block
__global__ __launch_bounds__(256) void global_avg_pool_kernel(const float* __restrict__ x, int nc, int spatial, float* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int plane = blockIdx.x * 16 + (threadIdx.x >> 4);
  float s = 0.f;
  if (plane < nc) {
    const float* p = x + (size_t)plane * spatial;
    for (int i = sub; i < spatial; i += 16) s += p[i];
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
  if (sub == 0 && plane < nc) y[plane] = s / (float)spatial;
}

// one block per row
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, int cols, float* __restrict__ y) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * cols;
  float* yr = y + (size_t)blockIdx.x * cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -3.402823466e38f;
  for (int i = threadIdx.x; i < cols; i += 256) mx = fmaxf(mx, xr[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) s += expf(xr[i] - mx);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = threadIdx.x; i < cols; i += 256) yr[i] = expf(xr[i] - mx) / s;
}

void launch_pack_fc(const int8_t* w_kn, int8_t* wp, int k, int n, hipStream_t s) {
  const size_t total = (size_t)((k + 3) / 4) * n * 4;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_fc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_kn, wp, k, n);
  const size_t ftotal = (size_t)((n + 31) / 32) * ((k + 31) / 32) * 1024;
  size_t fblocks = (ftotal + 255) / 256;
  if (fblocks > 4096) fblocks = 4096;
  hipLaunchKernelGGL(pack_fc_mfma_kernel, dim3((unsigned)fblocks), dim3(256), 0, s, w_kn, wp + fc_dot4_bytes(k, n), k, n);
}

void launch_fc(const int8_t* x, const int8_t* wp, const float* scale, const float* bias, void* y, int m, int k, int n,
               int relu, int out, hipStream_t s) {
  // The MFMA form is NOT the default any more: measured again at the end of round 3 (tools/fcbench.py, one box) the dot4
  // kernel below that stages x through LDS runs the network tails in half its time (m = 128, k = 1024, n = 1000: 8.8 vs
  // 18.7 us; 256 x 2048: 16.2 vs 30.9; 1024 x 1280: 22.3 vs 32.3): a lane of the MFMA form fetches its B operand from its own
  // x row (32 rows 1 KiB apart per load instruction).  PLHIP_FC_MFMA=1 selects it (parity test in a subprocess).
  const int fc_mfma_env = knob("FC_MFMA", 0);
  const size_t lds_f = (size_t)FCF_MB * k > (size_t)4 * FCF_MB * 64 * 4 ? (size_t)FCF_MB * k : (size_t)4 * FCF_MB * 64 * 4;
  const bool fast_ok = (k & 15) == 0 && ((uintptr_t)x & 15) == 0 && lds_f <= 64 * 1024;
  if ((fc_mfma_env || !fast_ok) && (k & 31) == 0) {
    dim3 grid((n + 31) / 32, (m + 31) / 32);
    const int8_t* wfrag = wp + fc_dot4_bytes(k, n);
    if (out == OUT_I32) hipLaunchKernelGGL((fc_i8_mfma_kernel<OUT_I32>), grid, dim3(256), 0, s, x, wfrag, scale, bias, y, m, k, n, relu);
    else if (out == OUT_F32) hipLaunchKernelGGL((fc_i8_mfma_kernel<OUT_F32>), grid, dim3(256), 0, s, x, wfrag, scale, bias, y, m, k, n, relu);
    else hipLaunchKernelGGL((fc_i8_mfma_kernel<OUT_I8>), grid, dim3(256), 0, s, x, wfrag, scale, bias, y, m, k, n, relu);
    return;
  }
  const size_t lds = (size_t)FCF_MB * k > (size_t)4 * FCF_MB * 64 * 4 ? (size_t)FCF_MB * k : (size_t)4 * FCF_MB * 64 * 4;
  if ((k & 15) == 0 && ((uintptr_t)x & 15) == 0 && lds <= 64 * 1024) {
    dim3 grid((n + 63) / 64, (m + FCF_MB - 1) / FCF_MB);
    if (out == OUT_I32) hipLaunchKernelGGL((fc_i8_fast_kernel<OUT_I32>), grid, dim3(256), lds, s, x, wp, scale, bias, y, m, k, n, relu);
    else if (out == OUT_F32) hipLaunchKernelGGL((fc_i8_fast_kernel<OUT_F32>), grid, dim3(256), lds, s, x, wp, scale, bias, y, m, k, n, relu);
    else hipLaunchKernelGGL((fc_i8_fast_kernel<OUT_I8>), grid, dim3(256), lds, s, x, wp, scale, bias, y, m, k, n, relu);
    return;
  }
  dim3 grid((n + 255) / 256, (m + FC_MB - 1) / FC_MB);
  if (out == OUT_I32) hipLaunchKernelGGL((fc_i8_kernel<OUT_I32>), grid, dim3(256), 0, s, x, wp, scale, bias, y, m, k, n, relu);
  else if (out == OUT_F32) hipLaunchKernelGGL((fc_i8_kernel<OUT_F32>), grid, dim3(256), 0, s, x, wp, scale, bias, y, m, k, n, relu);
  else hipLaunchKernelGGL((fc_i8_kernel<OUT_I8>), grid, dim3(256), 0, s, x, wp, scale, bias, y, m, k, n, relu);
}

static unsigned ew_blocks(int64_t quads) {
  int64_t b = (quads + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048 * 4) b = 2048 * 4;
  return (unsigned)b;
}

void launch_calib_f32_to_i8(const float* x, int8_t* y, float scale, int64_t count, hipStream_t s) {
  const float inv = 1.f / scale;  // type_trans.cc:45
  const int vec = (((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 3) == 0) ? 1 : 0;
  hipLaunchKernelGGL(calib_f32_to_i8_kernel, dim3(ew_blocks(vec ? count >> 2 : count)), dim3(256), 0, s, x, y, inv, count, vec);
}

void launch_calib_i8_to_f32(const int8_t* x, float* y, float scale, int64_t count, hipStream_t s) {
  const int vec = (((uintptr_t)y & 15) == 0 && ((uintptr_t)x & 3) == 0) ? 1 : 0;
  hipLaunchKernelGGL(calib_i8_to_f32_kernel, dim3(ew_blocks(vec ? count >> 2 : count)), dim3(256), 0, s, x, y, scale, count, vec);
}

void launch_global_avg_pool(const float* x, int nc, int spatial, float* y, hipStream_t s) {
  hipLaunchKernelGGL(global_avg_pool_kernel, dim3((nc + 15) / 16), dim3(256), 0, s, x, nc, spatial, y);
}

void launch_softmax(const float* x, int rows, int cols, float* y, hipStream_t s) {
  hipLaunchKernelGGL(softmax_kernel, dim3(rows), dim3(256), 0, s, x, cols, y);
}

}  // namespace plhip
