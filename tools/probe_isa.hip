// probe_isa.hip — tiny stand-alone probes of instruction semantics that the guides do not pin down.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_isa.hip -o /tmp/probe_isa ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const float* x, uint32_t* o, int n) {
  int i = threadIdx.x;
  if (i < n) {
    uint32_t r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(x[i]), "v"(0xAABBCCDDu));
    o[i] = r;
  }
}
int main() {
  float hx[] = {0.f, 0.4f, 0.5f, 0.6f, 1.49f, 1.5f, 2.5f, 3.5f, 253.9f, 254.f, 254.5f, 255.f, 255.9f, 300.f, -0.4f, -0.6f, -3.f, 1e9f};
  const int n = sizeof(hx) / sizeof(float);
  float* dx; uint32_t* d; uint32_t ho[64];
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&d, 64 * 4);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, d, n);
  hipMemcpy(ho, d, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) byte1 -> 0x%08x (byte = %u)\n", hx[i], ho[i], (ho[i] >> 8) & 0xff);
  return 0;
}
