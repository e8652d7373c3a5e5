// eltwise_pool.hip — the fp32 glue ops that sit between the int8 convs of the ResNet50 / MobileNetV2 programs
// (SURVEY.md Appendix D): pool2d (max / avg windows) and elementwise_add (+ fused relu) for gfx950.
//
// Replaces (reference, ARM — both ops exist there in fp32 only, so residual edges de/re-quantise through calib):
//   pool2d                              PoolCompute::Run lite/kernels/arm/pool_compute.cc:36-345 ->
//                                       pooling_basic lite/backends/arm/math/pooling.cc:38-215 (semantics of every
//                                       specialised pooling3x3s2p1_max etc.: max / sum over the window clipped to the image)
//   elementwise_add                     ElementwiseAddCompute lite/kernels/arm/elementwise_compute.cc:85-110 ->
//                                       elementwise_add<float> lite/backends/arm/math/elementwise.cc
//   fusion_elementwise_add_activation   ElementwiseAddActivationCompute (:112-140) -> elementwise_add_relu<float>
// Both are pure HBM streams: 16 bytes per lane, no LDS.
#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"

namespace plhip {

// One lane = 4 consecutive outputs of one output row.  Grid: x = quads of a row x rows (flattened), y = planes, so
// plane / row / validity of a window row are cheap.  Window rows and columns are clipped to the image exactly as
// pooling_basic does (sh/eh/sw/ew), the first valid element initialises the result.
template <bool MAX>
__global__ __launch_bounds__(256) void pool2d_f32_kernel(PoolArgs a) {
  const int owq = (a.ow + 3) >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= owq * a.oh) return;
  const int oy = idx / owq, oxq = idx - oy * owq;
  const size_t plane = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
  if (plane >= (size_t)a.planes) return;
  const float* __restrict__ xp = a.x + plane * (size_t)a.h * a.w;
  float* __restrict__ yp = a.y + plane * (size_t)a.oh * a.ow + (size_t)oy * a.ow;
  int sh = oy * a.sh - a.pt, eh = sh + a.kh;
  sh = sh < 0 ? 0 : sh;
  eh = eh > a.h ? a.h : eh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = oxq * 4 + j;
    if (ox >= a.ow) break;
    int sw = ox * a.sw - a.pl, ew = sw + a.kw;
    sw = sw < 0 ? 0 : sw;
    ew = ew > a.w ? a.w : ew;
    float r = 0.f;
    bool first = true;
    for (int y = sh; y < eh; ++y)
      for (int x = sw; x < ew; ++x) {
        const float v = xp[(size_t)y * a.w + x];
        if (first) { r = v; first = false; }
        else if (MAX) r = r >= v ? r : v;
        else r += v;
      }
    if (!MAX) {
      if (a.exclusive) {
        int div = (ew - sw) * (eh - sh);
        div = div > 0 ? div : 1;
        r /= (float)div;
      } else {
        // pooling.cc:165-205 as written (sw / sh are the CLIPPED window starts there): the full kernel area unless
        // the window touches the right / bottom image edge, where the covered padding is counted
        int bh = a.kh, bw = a.kw;
        if (ew == a.w) {
          bw = (sw + a.kw) >= (a.w + a.pr) ? (a.w + a.pr) : (sw + a.kw);
          bw -= sw;
          if ((sw - a.pl) < 0 && (sw + a.kw) > (a.w + a.pr)) bw += a.pl;
        }
        if (eh == a.h) {
          bh = (sh + a.kh) >= (a.h + a.pb) ? (a.h + a.pb) : (sh + a.kh);
          bh -= sh;
          if ((sh - a.pt) < 0 && (sh + a.kh) > (a.h + a.pb)) bh += a.pt;
        }
        r /= (float)(bh * bw);
      }
    }
    yp[ox] = r;
  }
}

// int8 max pool (kHIP graph fusion: conv -> pool2d(max) -> calib becomes conv+calib -> THIS; max commutes with the
// monotonic quantiser, so the bytes equal calib(pool(x))).  One lane = 4 consecutive outputs, packed into one dword.
__global__ __launch_bounds__(256) void pool2d_max_i8_kernel(PoolArgs a) {
  const int owq = (a.ow + 3) >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= owq * a.oh) return;
  const int oy = idx / owq, oxq = idx - oy * owq;
  const size_t plane = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
  if (plane >= (size_t)a.planes) return;
  const int8_t* __restrict__ xp = reinterpret_cast<const int8_t*>(a.x) + plane * (size_t)a.h * a.w;
  int8_t* __restrict__ yp = reinterpret_cast<int8_t*>(a.y) + plane * (size_t)a.oh * a.ow + (size_t)oy * a.ow;
  int sh = oy * a.sh - a.pt, eh = sh + a.kh;
  sh = sh < 0 ? 0 : sh;
  eh = eh > a.h ? a.h : eh;
  int r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = oxq * 4 + j;
    int sw = ox * a.sw - a.pl, ew = sw + a.kw;
    sw = sw < 0 ? 0 : sw;
    ew = ew > a.w ? a.w : ew;
    int m = -128;
    bool any = false;
    if (ox < a.ow)
      for (int y = sh; y < eh; ++y)
        for (int x = sw; x < ew; ++x) {
          const int v = xp[(size_t)y * a.w + x];
          m = v > m ? v : m;
          any = true;
        }
    r[j] = any ? m : 0;  // a window that covers padding only: 0, like pooling_basic
  }
  const int ox0 = oxq * 4;
  if (ox0 + 3 < a.ow && (((uintptr_t)(yp + ox0)) & 3) == 0) {
    *reinterpret_cast<uint32_t*>(yp + ox0) = pack4_i8(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ox0 + j < a.ow) yp[ox0 + j] = (int8_t)r[j];
  }
}

// 3x3 stride-2 int8 max pool (ResNet50's pool1: 256 x 64 planes of 112x112 per step): the generic kernel above walks
// its window byte by byte (0.27 ms, 0.9 TB/s).  Here a lane = 4 consecutive outputs of one row: 3 row windows of 12 bytes
// (dw_load_row: unaligned 12-byte fetch, row / left-border clamping, byte masks), bytes outside the image become -128,
// then 27 v_bfe_i32 + 9 + 4 v_max3_i32.  A window always holds a real element (pad <= 1 < kernel), so -128 never wins.
template <bool TAIL>
__device__ __forceinline__ void pool3x3s2_max_i8_body(const PoolArgs& a, int idx, int owq, size_t plane) {
  const int oy = idx / owq, oxq = idx - oy * owq;
  const int8_t* __restrict__ xp = reinterpret_cast<const int8_t*>(a.x) + plane * (size_t)a.h * a.w;
  int8_t* __restrict__ yp = reinterpret_cast<int8_t*>(a.y) + plane * (size_t)a.oh * a.ow + (size_t)oy * a.ow;
  const int start = 8 * oxq - a.pl;
  const int sh = start < 0 ? -start : 0;
  int lcol = start + sh;
  lcol = lcol > a.w - 1 ? a.w - 1 : lcol;
  uint32_t cmask[3];
  dw_col_masks<3>(start, a.w, cmask);
  const long room = ((long)a.planes - (long)plane) * a.h * a.w;  // bytes from this plane's start to the end of the tensor
  int v[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    uint32_t d[3];
    dw_load_row<3, TAIL>(xp, oy * 2 - a.pt + r, a.h, a.w, lcol, sh, room, cmask, d);
    const bool rv = oy * 2 - a.pt + r >= 0 && oy * 2 - a.pt + r < a.h;
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] |= rv ? (~cmask[i] & 0x80808080u) : 0x80808080u;  // outside the image: -128
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int b = __builtin_amdgcn_sbfe((int)d[j >> 2], 8 * (j & 3), 8);
      v[j] = r == 0 ? b : (b > v[j] ? b : v[j]);
    }
  }
  int o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = v[2 * i] > v[2 * i + 1] ? v[2 * i] : v[2 * i + 1];
    o[i] = m > v[2 * i + 2] ? m : v[2 * i + 2];
  }
  const int ox0 = oxq * 4;
  if (ox0 + 3 < a.ow && (((uintptr_t)(yp + ox0)) & 3) == 0) {
    *reinterpret_cast<uint32_t*>(yp + ox0) = pack4_i8(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ox0 + j < a.ow) yp[ox0 + j] = (int8_t)o[j];
  }
}

__global__ __launch_bounds__(256) void pool3x3s2_max_i8_kernel(PoolArgs a) {
  const int owq = (a.ow + 3) >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= owq * a.oh) return;
  const size_t plane = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
  if (plane >= (size_t)a.planes) return;
  // only the last plane's fetches can cross the end of the tensor
  if (plane + 1 == (size_t)a.planes) pool3x3s2_max_i8_body<true>(a, idx, owq, plane);
  else pool3x3s2_max_i8_body<false>(a, idx, owq, plane);
}

void launch_pool2d_max_i8(const PoolArgs& a, hipStream_t s) {
  const int owq = (a.ow + 3) >> 2;
  const int gy = a.planes < 32768 ? a.planes : 32768;
  dim3 grid((owq * a.oh + 255) / 256, gy, (a.planes + gy - 1) / gy);
  if (a.kh == 3 && a.kw == 3 && a.sh == 2 && a.sw == 2 && a.pt <= 1 && a.pl <= 1 && a.w >= 4 && (long)a.h * a.w >= 12 &&
      8 * (owq - 1) - a.pl < a.w)  // the last quad's window starts inside the row (dw_col_masks' precondition)
    hipLaunchKernelGGL(pool3x3s2_max_i8_kernel, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(pool2d_max_i8_kernel, grid, dim3(256), 0, s, a);
}

void launch_pool2d(const PoolArgs& a, hipStream_t s) {
  const int owq = (a.ow + 3) >> 2;
  const int gy = a.planes < 32768 ? a.planes : 32768;
  dim3 grid((owq * a.oh + 255) / 256, gy, (a.planes + gy - 1) / gy);
  if (a.is_max) hipLaunchKernelGGL((pool2d_f32_kernel<true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((pool2d_f32_kernel<false>), grid, dim3(256), 0, s, a);
}

// out = x + y  (then max(., 0) when relu): elementwise.cc elementwise_add / elementwise_add_relu, same-shape operands.
template <bool RELU>
__global__ __launch_bounds__(256) void eltwise_add_f32_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              float* __restrict__ o, int64_t count, int vec) {
  const int64_t nq = vec ? count >> 2 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
    const v4f a = reinterpret_cast<const v4f*>(x)[i], b = reinterpret_cast<const v4f*>(y)[i];
    v4f r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = a[j] + b[j];
      if (RELU) r[j] = r[j] > 0.f ? r[j] : 0.f;
    }
    reinterpret_cast<v4f*>(o)[i] = r;
  }
  for (int64_t t = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
    float r = x[t] + y[t];
    if (RELU) r = r > 0.f ? r : 0.f;
    o[t] = r;
  }
}

void launch_eltwise_add(const float* x, const float* y, float* o, int64_t count, int relu, hipStream_t s) {
  const int vec = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)o) & 15) == 0) ? 1 : 0;
  int64_t b = ((vec ? count >> 2 : count) + 255) / 256;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  if (relu) hipLaunchKernelGGL((eltwise_add_f32_kernel<true>), dim3((unsigned)b), dim3(256), 0, s, x, y, o, count, vec);
  else hipLaunchKernelGGL((eltwise_add_f32_kernel<false>), dim3((unsigned)b), dim3(256), 0, s, x, y, o, count, vec);
}

}  // namespace plhip
