// probe_dma_unaligned.hip — does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept source addresses that are
// not 4-byte aligned?  (Needed for a 16-byte-piece B loader on 7x7 planes, whose rows start at odd byte offsets.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_dma_unaligned.hip -o tools/_probe_dma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
__global__ void k(const uint8_t* src, uint8_t* out, int shift, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
  const int lane = threadIdx.x;
  // lane l fetches 16 bytes from src + shift + l*stride into buf[16 l ..]
  __builtin_amdgcn_global_load_lds((glb_ptr)(src + shift + (size_t)lane * stride), (lds_ptr)buf, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = buf[lane * 16 + i];
}
int main() {
  const int N = 1 << 16;
  uint8_t* h = (uint8_t*)malloc(N);
  for (int i = 0; i < N; ++i) h[i] = (uint8_t)((i * 131 + (i >> 8) * 7) & 0xff);
  uint8_t *d, *o;
  hipMalloc(&d, N); hipMalloc(&o, 1024);
  hipMemcpy(d, h, N, hipMemcpyHostToDevice);
  uint8_t ho[1024];
  const int shifts[] = {0, 4, 8, 2, 1, 3, 5, 7, 13};
  const int strides[] = {16, 49, 33};
  int bad = 0;
  for (int st : strides)
    for (int sh : shifts) {
      hipMemset(o, 0xEE, 1024);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d + 4096, o, sh, st);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("shift %d stride %d: HIP error %s\n", sh, st, hipGetErrorString(e)); return 1; }
      hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
      int mism = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i)
          if (ho[l * 16 + i] != h[4096 + sh + l * st + i]) ++mism;
      printf("shift %2d stride %2d: %s (%d mismatching bytes)\n", sh, st, mism ? "WRONG" : "ok", mism);
      bad += mism != 0;
    }
  printf(bad ? "RESULT: unaligned LDS-DMA not usable\n" : "RESULT: unaligned LDS-DMA ok\n");
  return 0;
}
