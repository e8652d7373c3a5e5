// ingest_bench.hip — how many bytes per clock can ONE CU take in, by load form?  (Round 3: the pointwise GEMM's K loop
// is bound by ~21 B/clk/CU of operand ingest; this probe measures the ceiling for the forms the kernel could use.)
//   mode 0: LDS-DMA (global_load_lds_dwordx4), 1 KiB contiguous per wave-instruction
//   mode 1: global_load_dwordx4 into VGPRs, 1 KiB contiguous per wave-instruction
//   mode 2: alternating 0 / 1 (the kernel's mix: activations by DMA, weight fragments into registers)
//   mode 3: LDS-DMA, 8 rows x 128 B per wave-instruction, row pitch `pitch` bytes (NCHW activation rows)
// source: "shared" = every workgroup reads the same `span` bytes (weights: L2 hits), "private" = workgroup b reads its own
// span (activations: first touch from HBM / Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/ingest_bench.hip -o tools/_ingest_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <signal.h>
#include <algorithm>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// every wave issues `iters` wave-instructions of 1 KiB, DEPTH in flight
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void ingest(const uint8_t* src, size_t wg_stride, unsigned span, int iters, int pitch,
                                              unsigned long long* stamps, int* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const uint8_t* base = src + (size_t)blockIdx.x * wg_stride;
  uint8_t* slot = lds + wave * (DEPTH * 1024);
  unsigned off = wave * 1024u;  // byte offset of this wave's next piece inside the span
  const unsigned step = nw * 1024u;
  const bool rows = MODE >= 3;
  const unsigned lane_off = MODE == 3 ? (unsigned)((lane >> 3) * pitch + (lane & 7) * 16)    // 8 rows x 128 B, row-major lanes
                          : MODE == 4 ? (unsigned)((lane & 7) * pitch + (lane >> 3) * 16)    // 8 rows x 128 B, row index fastest (the tr kernel's order)
                          : MODE == 5 ? (unsigned)((lane >> 4) * pitch + (lane & 15) * 16)   // 4 rows x 256 B
                          : lane * 16u;
  const unsigned rpi = MODE == 5 ? 4u : 8u;  // rows per wave-instruction
  const unsigned step3 = rows ? (unsigned)(nw * rpi * pitch) : step;
  if (rows) off = wave * rpi * pitch;
  v4i r[DEPTH];
  for (int d = 0; d < DEPTH; ++d) r[d] = v4i{0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  int acc = 0;
  auto issue = [&](int d, bool dma) {
    const uint8_t* p = base + off + lane_off;
    if (dma) __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(slot + d * 1024), 16, 0, 0);
    else r[d] = *reinterpret_cast<const v4i*>(p);
    off += step3;
    if (off + (rows ? rpi * pitch : 1024u) > span) off -= span / step3 * step3;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, MODE != 1);
  for (int i = DEPTH; i < iters; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const bool dma = MODE != 1;
      if (dma) wait_vm<DEPTH - 1>();
      else acc ^= r[d][0] ^ r[d][1] ^ r[d][2] ^ r[d][3];
      issue(d, dma);
    }
  }
  wait_vm<0>();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    acc ^= r[d][3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    unsigned long long* s = stamps + ((size_t)blockIdx.x * 16 + wave) * 4;
    s[0] = t0; s[1] = t1; s[2] = rt0; s[3] = rt1;
  }
  if (acc == 0x12345678) *sink = acc + lds[lane];
}

struct Res { double bclk, gbs_cu, wall_us, clk_ghz; };

template <int MODE, int DEPTH>
static Res run(const uint8_t* src, bool shared_src, unsigned span, int waves, int iters, int pitch, unsigned long long* dstamps, int* sink) {
  const int blocks = 256;
  const size_t lds = (size_t)waves * DEPTH * 1024;
  auto kfn = ingest<MODE, DEPTH>;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> walls, bclks, clks;
  std::vector<unsigned long long> h(256 * 16 * 4);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(64 * waves), lds, 0, src, shared_src ? (size_t)0 : (size_t)span, span, iters, pitch, dstamps, sink);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("HIP error\n"); exit(1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per_block, clk;
    for (int b = 0; b < blocks; ++b) {
      unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
      for (int w = 0; w < waves; ++w) {
        const unsigned long long* s = &h[((size_t)b * 16 + w) * 4];
        t0 = std::min(t0, s[0]); t1 = std::max(t1, s[1]); r0 = std::min(r0, s[2]); r1 = std::max(r1, s[3]);
      }
      per_block.push_back((double)waves * iters * 1024.0 / (double)(t1 - t0));
      clk.push_back((double)(t1 - t0) / ((double)(r1 - r0) * 10.0));  // cycles per ns = GHz (realtime = 100 MHz)
    }
    std::sort(per_block.begin(), per_block.end()); std::sort(clk.begin(), clk.end());
    walls.push_back(ms * 1e3); bclks.push_back(per_block[blocks / 2]); clks.push_back(clk[blocks / 2]);
  }
  std::sort(walls.begin(), walls.end());
  Res r; r.wall_us = walls[2]; r.bclk = bclks[2]; r.clk_ghz = clks[2]; r.gbs_cu = r.bclk * r.clk_ghz;
  return r;
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // run only the lines of this form
  signal(SIGPIPE, SIG_IGN);
  const size_t big = (size_t)256 * (2 << 20);  // 2 MiB private span per workgroup at most
  uint8_t* d; unsigned long long* st; int* sink;
  hipMalloc(&d, big + (1 << 20)); hipMemset(d, 1, big + (1 << 20));
  hipMalloc(&st, 256 * 16 * 4 * 8); hipMalloc(&sink, 4);
  const char* names[] = {"dma-1KiB", "vgpr-1KiB", "-", "dma-rows128", "dma-rows128T", "dma-rows256"};
  printf("%-12s %-8s %6s %6s %5s | %8s %8s %8s %6s\n", "form", "source", "span", "waves", "depth", "B/clk/CU", "GB/s/CU", "wall us", "GHz");
  auto line = [&](int mode, const char* srcname, unsigned span, int waves, int depth, Res r) {
    printf("%-12s %-8s %5uK %6d %5d | %8.1f %8.1f %8.1f %6.2f\n", names[mode], srcname, span >> 10, waves, depth, r.bclk, r.gbs_cu, r.wall_us, r.clk_ghz);
    fflush(stdout);
  };
  // bytes per CU per launch: waves * iters KiB; keep ~2 MiB per CU so that the run is long against launch effects
#define RUN(MODE, DEPTH, SH, SPAN, WAVES, PITCH)                                                                      \
  if (only < 0 || only == MODE) line(MODE, SH ? "shared" : "private", SPAN, WAVES, DEPTH,                                                          \
       run<MODE, DEPTH>(d, SH, SPAN, WAVES, (2048 / WAVES) / DEPTH * DEPTH, PITCH, st, sink))
  // weights-like: 256 KiB shared by all workgroups
  RUN(0, 4, true, 256u << 10, 4, 0);  RUN(0, 8, true, 256u << 10, 8, 0);
  RUN(1, 4, true, 256u << 10, 4, 0);  RUN(1, 8, true, 256u << 10, 4, 0);  RUN(1, 16, true, 256u << 10, 4, 0);
  RUN(1, 4, true, 256u << 10, 8, 0);  RUN(1, 8, true, 256u << 10, 8, 0);  RUN(1, 4, true, 256u << 10, 16, 0);
  RUN(3, 8, true, 256u << 10, 8, 196); RUN(4, 8, true, 256u << 10, 8, 196); RUN(5, 8, true, 256u << 10, 8, 196);
  RUN(3, 8, true, 256u << 10, 8, 784); RUN(4, 8, true, 256u << 10, 8, 784); RUN(5, 8, true, 256u << 10, 8, 784);
  RUN(3, 8, true, 256u << 10, 8, 49); RUN(4, 8, true, 256u << 10, 8, 49);
  RUN(3, 8, true, 256u << 10, 8, 256); RUN(4, 8, true, 256u << 10, 8, 256);
  // activations-like: private 2 MiB spans streamed once (HBM)
  RUN(0, 8, false, 2048u << 10, 8, 0); RUN(1, 8, false, 2048u << 10, 8, 0); RUN(1, 8, false, 2048u << 10, 16, 0);
  RUN(3, 8, false, 2048u << 10, 8, 196); RUN(4, 8, false, 2048u << 10, 8, 196);
  return 0;
}
