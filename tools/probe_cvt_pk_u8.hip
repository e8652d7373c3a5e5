// probe_cvt_pk_u8.hip — semantics of v_cvt_pk_u8_f32 (rounding, saturation) and v_lerp_u8 on gfx950: can the int8
// requantisation use them (float -> byte written in place; (t + 1) >> 1 on 4 packed bytes in one instruction)?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_pk_u8.hip -o tools/_probe_cvt ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* x, unsigned* o, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  unsigned r;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(x[i]), "v"(0xAABBCCDDu));
  o[2 * i] = r;
  unsigned p = ((unsigned)i * 0x01030507u) ^ 0xfe00fd01u;
  o[2 * i + 1] = __builtin_amdgcn_lerp(p, 0u, 0x01010101u);
}
int main() {
  float h[] = {-3.7f, -0.5f, -0.0f, 0.49f, 0.5f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 126.5f, 253.99f, 254.0f, 254.9f, 255.0f, 255.5f, 256.f, 300.f, 1e9f, NAN, INFINITY, -INFINITY};
  const int n = sizeof(h) / 4;
  float* d; unsigned* o; unsigned ho[64];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 8);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(ho, o, n * 8, hipMemcpyDeviceToHost);
  int lerp_ok = 1;
  for (int i = 0; i < n; ++i) {
    printf("cvt_pk_u8(%g) byte1 = %u, other bytes %s\n", h[i], (ho[2 * i] >> 8) & 0xff, (ho[2 * i] & 0xffff00ffu) == 0xAABB00DDu ? "preserved" : "CHANGED");
    unsigned p = ((unsigned)i * 0x01030507u) ^ 0xfe00fd01u, want = 0;
    for (int b = 0; b < 4; ++b) want |= ((((p >> (8 * b)) & 0xff) + 1) >> 1) << (8 * b);
    if (want != ho[2 * i + 1]) { lerp_ok = 0; printf("lerp(%08x) = %08x, per-byte (p+1)>>1 = %08x\n", p, ho[2 * i + 1], want); }
  }
  printf("v_lerp_u8(p, 0, 0x01010101) == per-byte (p + 1) >> 1 : %s\n", lerp_ok ? "yes" : "NO");
  return 0;
}
