// probe_cvt_rtz.hip — does v_cvt_pk_u8_f32 follow MODE.fp_round (bits 1:0 of HW_REG_MODE) on gfx950?  Under the default
// mode it rounds to nearest even (tools/probe_cvt_pk_u8.hip), which the requantisation cannot use; under round-toward-zero
// it would be trunc + saturate + byte insert in ONE instruction.  Also checks that v_fma_f32 / v_cvt_f32_i32 issued after
// the mode is restored round to nearest again, and what a v_fma_f32 issued INSIDE the RTZ window does (it must stay outside).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_rtz.hip -o tools/_probe_rtz ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const float* x, unsigned* o, float* f, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  unsigned rne, rtz;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(rne) : "v"(v), "v"(0u));
  float fin;
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %2, 0, %3\n\t"
      "v_fma_f32 %1, %2, %4, %5\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
      : "=&v"(rtz), "=&v"(fin)
      : "v"(v), "v"(0u), "v"(1.0000001f), "v"(0.3333333f));
  float fout;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(fout) : "v"(v), "v"(1.0000001f), "v"(0.3333333f));
  o[2 * i] = rne;
  o[2 * i + 1] = rtz;
  f[2 * i] = fin;
  f[2 * i + 1] = fout;
}
int main() {
  const int n = 1 << 20;
  float* h = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int kind = i & 7;
    const double u = rand() / (double)RAND_MAX;
    if (kind == 0) h[i] = (float)(rand() % 300 - 20);                       // integers, some outside 0..255
    else if (kind == 1) h[i] = (float)(rand() % 256) + 0.5f;                // exact halves
    else if (kind == 2) h[i] = nextafterf((float)(rand() % 256) + 0.5f, (rand() & 1) ? 1e9f : -1e9f);
    else if (kind == 3) h[i] = nextafterf((float)(rand() % 257), (rand() & 1) ? 1e9f : -1e9f);
    else h[i] = (float)(u * 262.0 - 3.0);
  }
  float *d, *f;
  unsigned* o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, n * 8);
  hipMalloc(&f, n * 8);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, f, n);
  unsigned* ho = (unsigned*)malloc(n * 8);
  float* hf = (float*)malloc(n * 8);
  hipMemcpy(ho, o, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hf, f, n * 8, hipMemcpyDeviceToHost);
  long bad_rtz = 0, bad_rne = 0, fma_in_differs = 0, fma_out_bad = 0;
  for (int i = 0; i < n; ++i) {
    const float v = h[i];
    const float c = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    const unsigned want_rtz = (unsigned)truncf(c), want_rne = (unsigned)nearbyintf(c);
    if ((ho[2 * i + 1] & 0xff) != want_rtz) {
      if (bad_rtz++ < 8) printf("RTZ mismatch: x = %.9g got %u want %u\n", v, ho[2 * i + 1] & 0xff, want_rtz);
    }
    if ((ho[2 * i] & 0xff) != want_rne) bad_rne++;
    const float want = fmaf(v, 1.0000001f, 0.3333333f);
    if (hf[2 * i + 1] != want) fma_out_bad++;
    if (hf[2 * i] != want) fma_in_differs++;
  }
  printf("v_cvt_pk_u8_f32 under MODE.round = RTZ == trunc(sat(x, 0, 255)) on %d values: %s (%ld mismatches)\n", n, bad_rtz ? "NO" : "yes", bad_rtz);
  printf("v_cvt_pk_u8_f32 under the default mode == nearbyint (RNE): %s (%ld mismatches)\n", bad_rne ? "NO" : "yes", bad_rne);
  printf("v_fma_f32 after the mode is restored == fmaf (RNE): %s (%ld mismatches)\n", fma_out_bad ? "NO" : "yes", fma_out_bad);
  printf("v_fma_f32 INSIDE the RTZ window differs from RNE fmaf on %ld of %d values (expected > 0: the fma must stay outside)\n", fma_in_differs, n);
  return bad_rtz ? 1 : 0;
}
