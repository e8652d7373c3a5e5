// probe_lds_unaligned.hip — does gfx950 serve UNALIGNED LDS reads / writes (ds_read_b32 / b64 / b96 / b128, ds_write_b32 /
// b16 at byte addresses)?  LDS byte a carries (a * 7 + 3) & 0xff; lane l reads at byte address 64 + l (every alignment)
// and the host compares with the expected bytes.  Also: unaligned ds_write_b32 / ds_write_b64 followed by byte reads.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_lds_unaligned.hip -o tools/_probe_lds ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v3i __attribute__((ext_vector_type(3)));
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) buf[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)buf;
  const uint32_t a = base + 64 + lane;
  int r1; v2i r2; v3i r3; v4i r4;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r1) : "v"(a) : "memory");
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(a) : "memory");
  asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r3) : "v"(a) : "memory");
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r4) : "v"(a) : "memory");
  uint32_t* o = out + lane * 16;
  o[0] = r1; o[1] = r2[0]; o[2] = r2[1]; o[3] = r3[0]; o[4] = r3[1]; o[5] = r3[2]; o[6] = r4[0]; o[7] = r4[1]; o[8] = r4[2]; o[9] = r4[3];
  __syncthreads();
  // unaligned writes: lane l writes 0xA0A1A2A3 + l at byte 1024 + 9 * l (all alignments), then bytes are read back
  const uint32_t wa = base + 1024 + 9 * lane;
  const uint32_t val = 0xA3A2A1A0u + 0x01010101u * lane;
  asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(wa), "v"(val) : "memory");
  __syncthreads();
  uint32_t got = 0;
  for (int i = 0; i < 4; ++i) got |= (uint32_t)buf[1024 + 9 * lane + i] << (8 * i);
  o[10] = got;
  o[11] = val;
  const uint32_t wb = base + 2048 + 5 * lane;  // ds_write_b16 at odd addresses
  asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(wb), "v"(val) : "memory");
  __syncthreads();
  o[12] = (uint32_t)buf[2048 + 5 * lane] | ((uint32_t)buf[2048 + 5 * lane + 1] << 8);
}

int main() {
  uint32_t* dout;
  hipMalloc(&dout, 64 * 16 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout);
  if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 1; }
  uint32_t h[64 * 16];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  auto expect = [](int a, int n, uint32_t* e) {
    for (int d = 0; d < n; ++d) {
      e[d] = 0;
      for (int i = 0; i < 4; ++i) e[d] |= (uint32_t)(uint8_t)((a + 4 * d + i) * 7 + 3) << (8 * i);
    }
  };
  int bad[6] = {0, 0, 0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) {
    uint32_t e[4];
    expect(64 + l, 4, e);
    const uint32_t* o = h + l * 16;
    const bool b32 = o[0] == e[0], b64 = o[1] == e[0] && o[2] == e[1], b96 = o[3] == e[0] && o[4] == e[1] && o[5] == e[2];
    const bool b128 = o[6] == e[0] && o[7] == e[1] && o[8] == e[2] && o[9] == e[3];
    const bool w32 = o[10] == o[11], w16 = o[12] == (o[11] & 0xffff);
    if (l < 8) printf("lane %d (addr%%4 = %d): read b32 %d b64 %d b96 %d b128 %d | write b32 %d (addr%%4 = %d) b16 %d\n", l, l & 3, b32, b64, b96, b128, w32, (9 * l) & 3, w16);
    bad[0] += !b32; bad[1] += !b64; bad[2] += !b96; bad[3] += !b128; bad[4] += !w32; bad[5] += !w16;
  }
  printf("mismatching lanes of 64: read b32 %d, b64 %d, b96 %d, b128 %d; write b32 %d, b16 %d\n", bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
  return 0;
}
