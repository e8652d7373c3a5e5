// probe_tr8.hip — what does ds_read_b64_tr_b8 (gfx950's transposing 8-bit LDS read) deliver, lane by lane?
// LDS is filled with 16-bit-unique tags (byte at address a carries a & 0xff; a second pass carries a >> 8), every lane
// supplies ITS OWN 8-byte-aligned address, and the 8 result bytes of every lane are reported as LDS addresses.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr8.hip -o tools/_probe_tr8 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void k(const int* lane_addr, uint32_t* out, int hi_pass) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) buf[i] = hi_pass ? (uint8_t)(i >> 8) : (uint8_t)(i & 0xff);
  __syncthreads();
  // the array's address must reach the asm statement, or the stores above are dead to the compiler
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)buf + (uint32_t)lane_addr[lane];
  v2i r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[lane * 2] = (uint32_t)r[0];
  out[lane * 2 + 1] = (uint32_t)r[1];
}

static void run(const char* title, const int* addr) {
  int* da; uint32_t* dout;
  hipMalloc(&da, 64 * 4); hipMalloc(&dout, 128 * 4);
  hipMemcpy(da, addr, 64 * 4, hipMemcpyHostToDevice);
  uint32_t lo[128], hi[128];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, 0);
  hipMemcpy(lo, dout, 512, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return; }
  hipMemcpy(hi, dout, 512, hipMemcpyDeviceToHost);
  printf("== %s\n", title);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d (addr %4d):", l, addr[l]);
    for (int b = 0; b < 8; ++b) {
      const int lb = (lo[l * 2 + (b >> 2)] >> (8 * (b & 3))) & 0xff, hb = (hi[l * 2 + (b >> 2)] >> (8 * (b & 3))) & 0xff;
      printf(" %4d", hb * 256 + lb);
    }
    printf("\n");
  }
}

int main() {
  int addr[64];
  // pattern 1: lane l -> 8 l (lane-linear 8-byte chunks)
  for (int l = 0; l < 64; ++l) addr[l] = 8 * l;
  run("addr = 8*lane", addr);
  // pattern 2: a [row][128] image: lane (g = l>>4, j = l&15): row = j>>1, sub-chunk = j&1, group g takes columns 16 g
  for (int l = 0; l < 64; ++l) { const int g = l >> 4, j = l & 15; addr[l] = (j >> 1) * 128 + 16 * g + 8 * (j & 1); }
  run("8 rows x 128 B image, group g -> columns 16g.., lane j -> row j>>1, sub-chunk j&1", addr);
  // pattern 3: alternative guess: lane j -> row j&7, sub-chunk j>>3
  for (int l = 0; l < 64; ++l) { const int g = l >> 4, j = l & 15; addr[l] = (j & 7) * 128 + 16 * g + 8 * (j >> 3); }
  run("8 rows x 128 B image, lane j -> row j&7, sub-chunk j>>3", addr);
  return 0;
}
