// probe_coexec.hip — what does a VALU instruction cost a SIMD while its matrix pipe runs v_mfma_i32_32x32x32_i8?  The fused
// depthwise -> pointwise kernel's rounds take MFMA time + VALU time (4300 cycles = 2050 + 2360: DESIGN 8), as if nothing
// overlapped.  One 512-thread block per CU (two waves per SIMD: w and w + 4), each half running a chosen instruction stream:
//   role 0: nothing (the wave exits at once)         role 1: N MFMAs back to back (4 accumulators)
//   role 2: R x N v_fma_f32 (independent chains)     role 3: R x N v_dot4_i32_i8          role 4: R x N v_perm_b32
//   role 5..7: one MFMA followed by R of fma / dot4 / perm, N times (the interleaved stream of the fused kernel)
// Prints median cycles per wave of each half (s_memtime), for every pairing asked for on the command line.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_coexec.hip -o tools/_probe_coexec ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int ROLE, int R>
__device__ __forceinline__ void run(int n, int lane, unsigned long long* out) {
  v16i acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) { acc0[r] = lane + r; acc1[r] = lane - r; acc2[r] = lane * r; acc3[r] = r; }
  v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 11};
  float f[16];
  int d[16];
  unsigned p[16];
  for (int i = 0; i < 16; ++i) { f[i] = lane + i; d[i] = lane * i; p[i] = lane * 0x01010101u + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define VALU_BLOCK(J)                                                                                                          \
  if (ROLE == 2 || ROLE == 5) {                                                                                                \
    _Pragma("unroll") for (int r = 0; r < R; ++r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(J * R + r) & 15]) : "v"(1.0001f)); \
  }                                                                                                                            \
  if (ROLE == 3 || ROLE == 6) {                                                                                                \
    _Pragma("unroll") for (int r = 0; r < R; ++r) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(d[(J * R + r) & 15]) : "v"(a[0]), "v"(b[0])); \
  }                                                                                                                            \
  if (ROLE == 4 || ROLE == 7) {                                                                                                \
    _Pragma("unroll") for (int r = 0; r < R; ++r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(p[(J * R + r) & 15]) : "v"(a[1]), "v"(0x05010400u)); \
  }
  for (int it = 0; it < n; ++it) {
    if (ROLE == 1 || ROLE >= 5) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
    VALU_BLOCK(0)
    if (ROLE == 1 || ROLE >= 5) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
    VALU_BLOCK(1)
    if (ROLE == 1 || ROLE >= 5) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
    VALU_BLOCK(2)
    if (ROLE == 1 || ROLE >= 5) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
    VALU_BLOCK(3)
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int s = acc0[0] + acc1[15] + acc2[3] + acc3[7];
  for (int i = 0; i < 16; ++i) s += (int)f[i] + d[i] + (int)p[i];
  if (lane == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}

template <int R>
__global__ __launch_bounds__(512, 2) void k(int ra, int rb, int n, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int role = wave < 4 ? ra : rb;
  unsigned long long* o = out + ((size_t)blockIdx.x * 8 + wave) * 2;
  switch (role) {
    case 1: run<1, R>(n, lane, o); break;
    case 2: run<2, R>(n, lane, o); break;
    case 3: run<3, R>(n, lane, o); break;
    case 4: run<4, R>(n, lane, o); break;
    case 5: run<5, R>(n, lane, o); break;
    case 6: run<6, R>(n, lane, o); break;
    case 7: run<7, R>(n, lane, o); break;
    default: if (lane == 0) { o[0] = 0; o[1] = 0; } break;
  }
}

int main(int argc, char** argv) {
  const int n = 64, blocks = 256;  // 4 (MFMA + R VALU) groups per iteration
  unsigned long long* d;
  hipMalloc(&d, blocks * 8 * 2 * 8);
  std::vector<unsigned long long> h(blocks * 8 * 2);
  const char* names[] = {"idle", "mfma", "fma", "dot4", "perm", "mfma+fma", "mfma+dot4", "mfma+perm"};
  auto go = [&](int ra, int rb, int R) {
    for (int rep = 0; rep < 3; ++rep) {
      if (R == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, ra, rb, n, d);
      else hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(512), 0, 0, ra, rb, n, d);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b;
    for (int bl = 0; bl < blocks; ++bl)
      for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back(h[(bl * 8 + w) * 2]);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("R=%2d  waves 0-3: %-10s %7llu cycles (%.1f per group) | waves 4-7: %-10s %7llu cycles (%.1f per group)\n", R, names[ra],
           a[a.size() / 2], a[a.size() / 2] / (4.0 * n), names[rb], b[b.size() / 2], b[b.size() / 2] / (4.0 * n));
  };
  // each stream alone on its SIMD
  for (int r = 1; r <= 7; ++r) go(r, 0, 8);
  // matrix stream beside a VALU stream
  go(1, 2, 8); go(1, 3, 8); go(1, 4, 8);
  // two matrix streams, two VALU streams
  go(1, 1, 8); go(2, 2, 8); go(3, 3, 8); go(4, 4, 8); go(2, 3, 8);
  // the interleaved stream on both halves (the fused kernel's shape: 1 MFMA + ~10 VALU)
  go(5, 5, 8); go(6, 6, 8); go(7, 7, 8); go(5, 5, 10); go(6, 6, 10);
  return 0;
}
