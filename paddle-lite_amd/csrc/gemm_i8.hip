// gemm_i8.hip — int8 x int8 -> int32 GEMM on v_mfma_i32_32x32x32_i8 with NCHW-native operands and a fused
// per-channel dequant/requant + bias + activation epilogue, plus the weight pre-pack and im2col kernels.
//
// Replaces (reference, ARM): gemm_prepack_int8 (lite/backends/arm/math/gemm_prepacked_int8.cc:5263-5457,
// hot loop :2582-2744, epilogue :643-796), prepackA_int8 (:109-224), packb_int8 (:3285),
// im2col<int8_t> (lite/backends/arm/math/conv_impl.cc:103-153) and the batch/group driver loops of
// conv1x1s1_gemm_int8 / conv_im2col_gemm_int8 (conv_impl.cc:260-331, 490-598).
//
// Shape mapping (conv_impl.cc:275-299): per group  Y[b] (M x N) = W (M x K) * X[b] (K x N),
// M = cout/g, K = cin/g*kh*kw, N = oh*ow.  Unlike the reference, the batch is folded into N
// (n = b*HWX + hw) so that late layers (N = 49) still fill whole MFMA tiles.
//
// MI355X design
//   * one wave owns a (32*MA) x 128 output tile: MA A-fragments x 4 B-fragments of the 32x32x32 MFMA,
//     accumulators live in 64*MA VGPRs (VGPR form: the epilogue reads every value);
//   * A (weights) is pre-packed once into MFMA fragment order, so a wave's A fragment is one fully
//     coalesced 1 KiB dwordx4 load served by L2;
//   * B (activations, K x N with N contiguous = NCHW slab) is transposed IN REGISTERS with v_perm_b32 into the
//     K-contiguous 16-byte-per-lane operand the MFMA wants: lane (c = lane&31, h = lane>>5) ends up
//     with, for i = 0..3, column n = 4c+i, k = 16h..16h+15.  MFMA i therefore computes columns
//     {4c+i}, so each lane finishes with 4 CONSECUTIVE n for every output row and the int8 result is
//     stored as one dword per row (32 lanes x 4 B = 128 contiguous bytes of an NCHW row);
//   * four kernels share this scheme and differ in how B reaches the registers:
//       gemm_i8_nchw_kernel  private tiles, no LDS, no barrier (M <= 64 or one K-step: purely streaming layers);
//       gemm_i8_lds_kernel   4 waves split M and share B through LDS in fragment order (register-staged loads);
//       gemm_i8_dma_kernel   LDS-DMA ring (16-byte pieces, counted vmcnt, one barrier per K-step), A through LDS or
//                            (NG > 0) straight into a register ring; also the implicit-GEMM route of dense k x k convs;
//       gemm_i8_ws_kernel    wave-specialised experiment (opt-in).
//     All of them map block b to the (b % 8)-th eighth of the N tiles (XCD-contiguous work, xcd_tile_map).
// The MFMA's k-slot <-> (lane>>5, byte) map never matters: A and B use the same one.
#include <stdlib.h>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include <type_traits>

#include "gemm_epilogue.h"
#include "dw_common.h"

namespace plhip {

// ALIGNED: every row dword is 4-byte aligned and inside the tensor.  Otherwise (dense slabs with HW % 4 != 0, e.g. the
// 7x7 layers, or a misaligned base) the dwords are read unaligned — legal for global memory on gfx950 — and the one
// dword that would cross the end of the tensor is assembled bytewise.  Columns >= HW of a 4-column group then hold
// bytes of the next row: harmless, a GEMM column only ever feeds its own (discarded) output column.
// XCD-aware tile map of the shared-B kernels.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each), so
// block b runs on XCD b % 8.  (1) The mtb_n blocks that share one B tile get ids equal mod 8 and adjacent in dispatch
// order: the tile crosses the fabric once instead of mtb_n times (FETCH_SIZE was 3.4x the algorithmic bytes on the M = 512
// layers).  (2) Each XCD owns a CONTIGUOUS range of N tiles: a tile's 128-byte row segments are not cache-line aligned
// (row pitch HW = 196, 784, 3136 ...), so neighbouring tiles share most of their cache lines; with neighbours on
// different XCDs every such line was fetched twice (FETCH x2 was still 2.2-2.4x the input bytes after (1)).
__device__ __forceinline__ void xcd_tile_map(int b, int mtb_n, int NT, int& mtb, int& nt) {
  const int ntx = (NT + 7) >> 3;  // N tiles per XCD
  const int x = b & 7, q = b >> 3;
  const int j = q / mtb_n;
  mtb = q - j * mtb_n;
  nt = x * ntx + j;  // >= NT for the padding blocks of the last XCDs: callers return
}

template <bool ALIGNED>
__device__ __forceinline__ void load_b(const int8_t* __restrict__ xb, int ks, int h, int K, int XP, long room, uint32_t (&raw)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    int k = ks * 32 + 16 * h + j;
    k = k < K ? k : K - 1;  // rows >= K meet zero-padded weights; only the address must stay legal
    const long off = (long)k * XP;
    if (ALIGNED) {
      raw[j] = *reinterpret_cast<const uint32_t*>(xb + off);
    } else if (off + 4 <= room) {
      uint32_t v;
      __builtin_memcpy(&v, xb + off, 4);
      raw[j] = v;
    } else {
      uint32_t v = 0;
      for (int i = 0; i < 4; ++i)
        if (off + i < room) v |= (uint32_t)(uint8_t)xb[off + i] << (8 * i);
      raw[j] = v;
    }
  }
}

template <int MA, int OUT, bool VEC_STORE, bool MFULL, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_i8_nchw_kernel(GemmArgs g) {
  PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.x); PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias);
  PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.K); PLHIP_PRELOAD(g.KS); PLHIP_PRELOAD(g.HWX); PLHIP_PRELOAD(g.HWY); PLHIP_PRELOAD(g.XP);
  PLHIP_PRELOAD(g.NB); PLHIP_PRELOAD(g.x_bstride); PLHIP_PRELOAD(g.y_bstride); PLHIP_PRELOAD(g.MT); PLHIP_PRELOAD(g.NT);
  PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha); PLHIP_PRELOAD(g.dbg);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler too
  const uint32_t wid = blockIdx.x * 4u + (uint32_t)wave;  // MT * NT < 2^31 (launcher)
  if (wid >= (uint32_t)g.MT * (uint32_t)g.NT) return;  // wave-uniform; the kernel uses no barrier
  const int nt = (int)(wid / (uint32_t)g.MT);
  const int mt = (int)(wid - (uint32_t)nt * (uint32_t)g.MT);
  __shared__ __attribute__((aligned(16))) float lsb_all[4][2 * MA * 32];
  float* lsb = lsb_all[wave];
  const int c = lane & 31, h = lane >> 5;
  const int ntot = g.NB * g.HWX;  // multiple of 4 by construction

  int n4 = nt * 128 + 4 * c;
  const bool nvalid = n4 < ntot;
  if (!nvalid) n4 = 0;
  const int b = n4 / g.HWX;
  const int hw = n4 - b * g.HWX;
  const int8_t* xb = g.x + (size_t)b * g.x_bstride + hw;
  const long room = g.x_bytes - ((long)b * (long)g.x_bstride + hw);  // bytes from xb to the end of the tensor

  v16i acc[MA][4];  // first written by K-step 0 (zero addend): no separate zeroing of the 64*MA registers
  uint32_t raw[16];
  v4i af[MA];
  load_b<ALIGNED>(xb, 0, h, g.K, g.XP, room, raw);
  load_a<MA>(g.wp, mt, g.KS, 0, lane, af);
  float my_s = 1.f, my_b = 0.f;  // this lane's share of the tile's scale / bias, issued behind the first operand loads
  if (OUT != OUT_I32) load_scale_bias<MA>(g, mt, lane, my_s, my_b);

  auto kbody = [&](int ks, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    v4i bf[4];
#pragma unroll
    for (int jg = 0; jg < 4; ++jg) {
      uint32_t o0, o1, o2, o3;
      transpose4x4_b8(raw[4 * jg], raw[4 * jg + 1], raw[4 * jg + 2], raw[4 * jg + 3], o0, o1, o2, o3);
      bf[0][jg] = (int)o0;
      bf[1][jg] = (int)o1;
      bf[2][jg] = (int)o2;
      bf[3][jg] = (int)o3;
    }
    v4i ac[MA];
#pragma unroll
    for (int a = 0; a < MA; ++a) ac[a] = af[a];
    if (ks + 1 < g.KS) {  // prefetch the next K-step under this step's MFMAs
      load_b<ALIGNED>(xb, ks + 1, h, g.K, g.XP, room, raw);
      load_a<MA>(g.wp, mt, g.KS, ks + 1, lane, af);
    }
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[a][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ac[a], bf[i], FIRST ? zero : acc[a][i], 0, 0, 0);
  };
  kbody(0, std::integral_constant<bool, true>{});  // (PLHIP_GEMM_DEBUG bit 1 cannot skip the first step any more)
  for (int ks = 1; ks < ((g.dbg & 2) ? 0 : g.KS); ++ks) kbody(ks, std::integral_constant<bool, false>{});

  if (OUT != OUT_I32) store_scale_bias<MA, OUT>(lsb, lane, my_s, my_b);
  if (!nvalid || (g.dbg & 1)) return;
  if (OUT == OUT_I32) {
    gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw);
    return;
  }
  switch (g.act) {  // wave-uniform: one straight-line epilogue per activation
    case ACT_RELU: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    case ACT_RELU6: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    case ACT_LEAKY: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    default: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
  }
}

// =====================================================================================================================
// LDS-shared variant for M >= 128: a 256-thread block owns a (4 * 32*MA) x 128 tile; its 4 waves split M and SHARE the
// B tile.  Per stage (4 K-steps = 128 k) wave w fetches + transposes K-step w of the stage (16 coalesced dword loads,
// 32 v_perm) and writes its four 1-KiB MFMA fragments to LDS in fragment order (ds_write_b128, conflict free); after
// ONE barrier per stage every wave reads all 16 fragments back (ds_read_b128, lane-linear) for its own 4*MA MFMAs per
// K-step.  B therefore crosses L2 -> CU once per block instead of once per wave (4x less), LDS is double buffered
// (2 x 16 KiB) and the next stage's global loads are in flight under the current stage's MFMAs.  A fragments stay
// private (fragment-ordered, one coalesced 1-KiB load each) and are prefetched two K-steps ahead in a 4-deep
// register ring with static indices.
template <int MA, int OUT, bool VEC_STORE, bool MFULL, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_i8_lds_kernel(GemmArgs g) {
  PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.x); PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias);
  PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.K); PLHIP_PRELOAD(g.KS); PLHIP_PRELOAD(g.HWX); PLHIP_PRELOAD(g.HWY); PLHIP_PRELOAD(g.XP);
  PLHIP_PRELOAD(g.NB); PLHIP_PRELOAD(g.x_bstride); PLHIP_PRELOAD(g.y_bstride); PLHIP_PRELOAD(g.MT); PLHIP_PRELOAD(g.NT);
  PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha); PLHIP_PRELOAD(g.dbg);
  __shared__ __attribute__((aligned(16))) v4i bs[2][4][4][64];  // [buf][kstep][i][lane] : 32 KiB
  __shared__ __attribute__((aligned(16))) float lsb_all[4][2 * MA * 32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mtb_n = (g.MT + 3) >> 2;  // blocks along M
  int mtb, nt;
  xcd_tile_map(blockIdx.x, mtb_n, g.NT, mtb, nt);
  if (nt >= g.NT) return;  // block-uniform (grid is padded to 8 N-tiles)
  const int mt = mtb * 4 + wave;
  const bool mactive = mt < g.MT;  // wave-uniform; inactive waves still load their share of B and hit the barriers
  const int mtc = mactive ? mt : g.MT - 1;
  float* lsb = lsb_all[wave];
  const int c = lane & 31, h = lane >> 5;
  const int ntot = g.NB * g.HWX;

  int n4 = nt * 128 + 4 * c;
  const bool nvalid = n4 < ntot;
  if (!nvalid) n4 = 0;
  const int b = n4 / g.HWX;
  const int hw = n4 - b * g.HWX;
  const int8_t* xb = g.x + (size_t)b * g.x_bstride + hw;
  const long room = g.x_bytes - ((long)b * (long)g.x_bstride + hw);

  v16i acc[MA][4];  // first written by K-step 0 (zero addend)

  const int KS = g.KS;
  const int S = (KS + 3) >> 2;
  uint32_t raw[16];
  v4i a0[MA], a1[MA], a2[MA], a3[MA];

  auto stage_write = [&](int buf) {  // transpose raw (this wave's K-step) into 4 fragments of bs[buf][wave]
    v4i bf[4];
#pragma unroll
    for (int jg = 0; jg < 4; ++jg) {
      uint32_t o0, o1, o2, o3;
      transpose4x4_b8(raw[4 * jg], raw[4 * jg + 1], raw[4 * jg + 2], raw[4 * jg + 3], o0, o1, o2, o3);
      bf[0][jg] = (int)o0;
      bf[1][jg] = (int)o1;
      bf[2][jg] = (int)o2;
      bf[3][jg] = (int)o3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[buf][wave][i][lane] = bf[i];
  };
  auto kstep_t = [&](int buf, int kk, const v4i (&af)[MA], auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bf[i] = bs[buf][kk][i][lane];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[a][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[a], bf[i], FIRST ? zero : acc[a][i], 0, 0, 0);
  };
  auto kstep = [&](int buf, int kk, const v4i (&af)[MA]) { kstep_t(buf, kk, af, std::integral_constant<bool, false>{}); };
  auto load_a_c = [&](int ks, v4i (&af)[MA]) {  // clamped: K-steps past the end reload the last one (never used)
    load_a<MA>(g.wp, mtc, KS, ks < KS ? ks : KS - 1, lane, af);
  };

  // prologue: stage 0 into LDS, A ring primed with K-steps 0 and 1
  if (wave < KS) load_b<ALIGNED>(xb, wave, h, g.K, g.XP, room, raw);
  load_a_c(0, a0);
  load_a_c(1, a1);
  float my_s = 1.f, my_b = 0.f;
  if (OUT != OUT_I32) load_scale_bias<MA>(g, mtc, lane, my_s, my_b);
  if (wave < KS) stage_write(0);
  __syncthreads();

  for (int s = 0; s < ((g.dbg & 2) ? 0 : S); ++s) {
    const int buf = s & 1;
    const int ks0 = 4 * s;
    const bool more = s + 1 < S;
    const int myks = ks0 + 4 + wave;  // the K-step this wave stages for the next round
    if (more && myks < KS) load_b<ALIGNED>(xb, myks, h, g.K, g.XP, room, raw);
    // 4 K-steps, A ring: slot kk holds K-step ks0+kk; refill two steps ahead
    load_a_c(ks0 + 2, a2);
    if (mactive) {
      if (s == 0) kstep_t(buf, 0, a0, std::integral_constant<bool, true>{});  // uniform
      else kstep(buf, 0, a0);
    }
    load_a_c(ks0 + 3, a3);
    if (mactive && ks0 + 1 < KS) kstep(buf, 1, a1);
    load_a_c(ks0 + 4, a0);
    if (mactive && ks0 + 2 < KS) kstep(buf, 2, a2);
    load_a_c(ks0 + 5, a1);
    if (mactive && ks0 + 3 < KS) kstep(buf, 3, a3);
    if (more && myks < KS) stage_write(buf ^ 1);
    __syncthreads();
  }

  if (OUT != OUT_I32) store_scale_bias<MA, OUT>(lsb, lane, my_s, my_b);
  if (!nvalid || !mactive || (g.dbg & 1)) return;
  if (OUT == OUT_I32) {
    gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw);
    return;
  }
  switch (g.act) {
    case ACT_RELU: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    case ACT_RELU6: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    case ACT_LEAKY: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    default: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
  }
}

// =====================================================================================================================
// ---- diagnostic timeline (PLHIP_GEMM_DEBUG & 32; never set in production): per-wave s_memtime stamps of the LDS-DMA
// kernel, kept in LDS during the run and flushed to this buffer at the end (plhip_debug_read_stamps reads it).
constexpr int STAMP_SLOTS = 32;
__device__ unsigned long long g_stamps[1024 * 4 * STAMP_SLOTS];
#define PLHIP_STAMP(i)                                                                        \
  do {                                                                                        \
    if (diag && lane == 0) lstamp[i] = __builtin_amdgcn_s_memtime();                          \
  } while (0)

// LDS-DMA ring variant (the fast path for MFMA-heavy layers: M >= 256-ish, K >= 128, 4-byte aligned rows).
// PMC on the register-staged kernel showed MFMA busy ~13 % per wave and one full memory latency per stage: register
// staging cannot keep enough K-steps in flight (VGPR-bound), and the in-order vmcnt couples the short A loads to the
// long B loads.  Here NOTHING in the K loop loads into VGPRs from global memory:
//   * per K-step the block needs a raw B slab (32 k-rows x 128 columns = 4 KiB, NCHW rows as they lie in memory) and
//     each wave its own MA fragment-ordered A tiles (1 KiB each).  Both are fetched by LDS-DMA
//     (global_load_lds dword / dwordx4): asynchronous, no VGPRs, D-1 K-steps in flight per wave in an NS-slot ring;
//   * a wave waits with a COUNTED s_waitcnt vmcnt for its own share of K-step ks, one raw s_barrier makes the whole
//     slot visible, then every wave reads the raw rows (16 ds_read_b32), transposes them in registers (32 v_perm) and
//     reads its A tiles (ds_read_b128) for 4*MA MFMAs.  The transposes are redundant across the 4 waves but run on the
//     VALU beside the MFMA pipe (128 of 256 MFMA cycles per K-step at MA = 2).
// One barrier per K-step; slot reuse distance NS - (D-1) = 2 iterations, so a slot is rewritten only after every wave
// has passed the barrier that follows its last read.

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// GD_D = K-steps in flight (including the one being consumed); ring slots GD_NS = GD_D + 1.
// AREG: the A fragments go straight from L2 into a 4-deep REGISTER ring (they are already in fragment order in memory) and
// never touch LDS: per block and K-step the LDS sees 20 KB (4 KB of B written once, read by 4 waves) instead of 36 KB --
// with two blocks per CU the old traffic alone (72 KB / 128 B per clock = 562 cycles) exceeded the MFMA time of a K-step
// pair (512).  Needs KS % 4 == 0 (static register names: the loop is unrolled by the ring size).
// NG = KS / 4 as a template constant: the K loop is then straight-line code.  With a loop back-edge the compiler cannot
// count the vector-memory operations in flight and protects the loop-carried fragment registers with s_waitcnt vmcnt(0)
// at the top of every group, which drains the whole pipeline.  NG == 0: A through LDS (any KS).
template <int MA, int OUT, bool VEC_STORE, bool MFULL, int GD_D, int NG>
__global__ __launch_bounds__(256, GD_D <= 4 ? 2 : 1) void gemm_i8_dma_kernel(GemmArgs g) {
  constexpr bool AREG = NG > 0;
  PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.x); PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias);
  PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.K); PLHIP_PRELOAD(g.KS); PLHIP_PRELOAD(g.HWX); PLHIP_PRELOAD(g.HWY); PLHIP_PRELOAD(g.XP);
  PLHIP_PRELOAD(g.NB); PLHIP_PRELOAD(g.x_bstride); PLHIP_PRELOAD(g.y_bstride); PLHIP_PRELOAD(g.MT); PLHIP_PRELOAD(g.NT);
  PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha); PLHIP_PRELOAD(g.dbg);
  PLHIP_PRELOAD(g.im_kw); PLHIP_PRELOAD(g.im_khkw); PLHIP_PRELOAD(g.im_c); PLHIP_PRELOAD(g.im_ph); PLHIP_PRELOAD(g.im_pw); PLHIP_PRELOAD(g.im_oh);
  constexpr int GD_NS = GD_D + 1;
  constexpr int SLOT = AREG ? 4096 : 4096 + 4 * MA * 1024;
  constexpr int PER = 1 + MA;  // vector-memory instructions per wave per K-step (1 B piece + MA A fragments)
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];  // GD_NS * SLOT ring + 4 waves x scale/bias (ONE LDS object)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mtb_n = (g.MT + 3) >> 2;
  int mtb, nt;
  xcd_tile_map(blockIdx.x, mtb_n, g.NT, mtb, nt);
  if (nt >= g.NT) return;  // block-uniform (grid is padded to 8 N-tiles)
  const int mt = mtb * 4 + wave;
  const bool mactive = mt < g.MT;
  const int mtc = mactive ? mt : g.MT - 1;
  const int c = lane & 31, h = lane >> 5;
  // Column space: every image's HWX columns are padded to HWP = roundup(HWX, 16) so that the B tile moves as 16-byte
  // pieces (ONE 1-KiB LDS-DMA instruction per wave and K-step instead of four 256-byte ones: the loop was bound by the
  // number of vector-memory instructions, not by bytes).  The last piece of an image is END-aligned (source columns
  // HWX-16 .. HWX-1), so nothing is read outside the plane; its first 16-r columns duplicate earlier ones and are
  // never stored (r = HWX % 16; with HWX % 4 != 0 one lane per image holds both kinds: `skip`).  Neither the rows nor
  // the pieces need any alignment: LDS-DMA takes byte-aligned global addresses (tools/probe_dma_unaligned.hip).
  const int HWP = (g.HWX + 15) & ~15, full16 = g.HWX & ~15, rem16 = g.HWX & 15;
  int n4 = nt * 128 + 4 * c;
  int b = n4 / HWP;
  int j = n4 - b * HWP;
  // lane's columns j .. j+3: real if below full16, or (last piece) from column HWP - rem16 on; `skip` leading duplicates
  int skip = j < full16 ? 0 : HWP - rem16 - j;
  skip = skip < 0 ? 0 : skip;
  const bool nvalid = b < g.NB && skip < 4;
  if (!nvalid) { b = 0; j = 0; skip = 0; }
  int hw = j < full16 ? j : j + rem16 - 16;
  const bool implicit = g.im_kw > 0;  // wave-uniform
  if (implicit) {  // "image" = output row (batch image, oh): the epilogue wants the batch image and hw = oh*OW + column
    const int bi = b / g.im_oh;
    hw += (b - bi * g.im_oh) * g.HWX;
    b = bi;
  }
  // my 16-byte piece of the B tile: row 8*wave + lane/8 of the K-step, columns 16*(lane&7) ...
  const int prow = 8 * wave + (lane >> 3);
  const int8_t* xb;
  {
    const int np = nt * 128 + 16 * (lane & 7);
    int pb = np / HWP;
    int pj = np - pb * HWP;
    if (pb >= g.NB) { pb = 0; pj = 0; }
    const int pcol = pj < full16 ? pj : g.HWX - 16;
    if (implicit) {
      const int bi = pb / g.im_oh, oh = pb - bi * g.im_oh;
      xb = g.x + ((size_t)bi * g.im_c * g.im_ph + oh) * g.im_pw + pcol;
    } else {
      xb = g.x + (size_t)pb * g.x_bstride + pcol;
    }
  }
  // implicit GEMM: (channel, tap) of this lane's K-row, advanced by 32 rows per issued K-step (issue() is called with
  // consecutive K-steps), so that no division sits in the loop
  int kc = 0, krs = 0;
  if (implicit) {
    kc = prow / g.im_khkw;
    krs = prow - kc * g.im_khkw;
  }
  const int kc_step = implicit ? 32 / g.im_khkw : 0, krs_step = implicit ? 32 - kc_step * g.im_khkw : 0;
  const int8_t* ab = g.wp + (size_t)mtc * MA * g.KS * 1024 + lane * 16;
  const int KS = g.KS;
  float* lsb = reinterpret_cast<float*>(ring + GD_NS * SLOT) + wave * 2 * MA * 32;
  const bool diag = (g.dbg & 32) != 0;
  unsigned long long* lstamp = reinterpret_cast<unsigned long long*>(ring + GD_NS * SLOT + 4 * 2 * MA * 32 * 4) + wave * STAMP_SLOTS;
  if (diag && lane == 0) {
    lstamp[0] = __builtin_amdgcn_s_memrealtime();
    lstamp[1] = __builtin_amdgcn_s_memtime();
    lstamp[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }

  auto issue = [&](int ks, int slot) {
    uint8_t* sb = ring + slot * SLOT;
    size_t koff;
    if (implicit) {
      // rows past K meet zero-padded weights: any in-bounds address will do (the last real tap)
      const int c = kc < g.im_c ? kc : g.im_c - 1, rs = kc < g.im_c ? krs : g.im_khkw - 1;
      const int r = (rs * ((65536 + g.im_kw - 1) / g.im_kw)) >> 16;  // rs / kw, exact for rs < 128, kw <= 11
      koff = ((size_t)c * g.im_ph + r) * g.im_pw + (rs - r * g.im_kw);
      kc += kc_step;
      krs += krs_step;
      if (krs >= g.im_khkw) {
        krs -= g.im_khkw;
        ++kc;
      }
    } else {
      int k = ks * 32 + prow;
      k = k < g.K ? k : g.K - 1;  // rows past K meet zero-padded weights
      koff = (size_t)k * g.XP;
    }
    __builtin_amdgcn_global_load_lds((glb_ptr)(xb + koff), (lds_ptr)(sb + wave * 1024), 16, 0, 0);
    if (!AREG) {
#pragma unroll
      for (int a = 0; a < MA; ++a)
        __builtin_amdgcn_global_load_lds((glb_ptr)(ab + ((size_t)a * KS + ks) * 1024), (lds_ptr)(sb + 4096 + (wave * MA + a) * 1024), 16, 0, 0);
    }
  };
  // AREG: fragment order in memory == register layout.  The load is issued through inline asm ON PURPOSE: next to LDS-DMA
  // operations the compiler's wait-count pass protects every register written by an ordinary load with
  // s_waitcnt vmcnt(0) (seen in the ISA even in straight-line code), which would drain the whole DMA pipeline at each
  // first use.  The asm load is invisible to that pass; the counted waits of the pipeline (wait_vmcnt, in-order vmcnt)
  // already guarantee that K-step ks's fragments have arrived one iteration before they are multiplied.  The price is a
  // build-time obligation: no register copy of an in-flight fragment may be inserted between the load and its MFMAs --
  // the parity suite (random operands) fails on any such copy, and tools/check_areg_isa.py checks the ISA.
  auto load_a_regs = [&](int ks, v4i (&dst)[MA]) {
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      const int8_t* p = ab + ((size_t)a * KS + ks) * 1024;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[a]) : "v"(p) : "memory");
    }
  };

  // ---- pipeline ----
  // Ring of GD_NS = GD_D + 1 slots, AHEAD = GD_D K-steps in flight.  Iteration ks:
  //   wait(my pieces of K-step ks+1 landed) ; barrier ; issue the LDS reads of K-step ks+1 (raw B rows + my A fragments) ;
  //   MFMAs of K-step ks from registers, and IN THEIR SHADOW: the DMA of K-step ks+AHEAD (its slot held K-step ks-1,
  //   which every wave finished reading before this barrier), then the 32 v_perm that turn the raw rows of ks+1 into B
  //   fragments.  An in-order wave stalls at the first consumer of an LDS read, so the first MFMAs carry the DMA issue
  //   and the transposes start only after them (timeline stamps: the former order -- waitcnt lgkmcnt(0) in front of the
  //   first MFMA, DMA issue and a run-time vmcnt switch in front of the barrier -- cost ~1190 cycles per K-step for
  //   256 cycles of MFMA).
  constexpr int AHEAD = GD_D;
  // prologue: K-steps 0 .. AHEAD-1 in flight (the launcher guarantees KS >= AHEAD); the accumulators are zeroed
  // behind the issue, while the first bytes travel
  v4i aring[AREG ? 4 : 1][MA];
  static_assert(!AREG || GD_D == 4, "the register ring is as deep as the DMA look-ahead");
#pragma unroll
  for (int p = 0; p < AHEAD; ++p) {
    issue(p, p);
    if (AREG) load_a_regs(p, aring[AREG ? p : 0]);
  }
  float my_s = 1.f, my_b = 0.f;
  if (OUT != OUT_I32) load_scale_bias<MA>(g, mtc, lane, my_s, my_b);  // 2 more vmcnt entries, younger than the prologue DMA
  __builtin_amdgcn_sched_barrier(0);
  v16i acc[MA][4];
  if (!AREG) {  // AREG: straight-line K loop, K-step 0 multiplies into a zero addend instead (no 64*MA v_mov)
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][r] = 0;
  }

  auto read_slot = [&](int slot, uint32_t (&raw)[16], v4i (&af)[MA]) {
    const uint8_t* sb = ring + slot * SLOT;
#pragma unroll
    for (int j = 0; j < 16; ++j) raw[j] = *reinterpret_cast<const uint32_t*>(sb + (16 * h + j) * 128 + 4 * c);
    if (!AREG) {
#pragma unroll
      for (int a = 0; a < MA; ++a) af[a] = *reinterpret_cast<const v4i*>(sb + 4096 + (wave * MA + a) * 1024 + lane * 16);
    }
  };
  auto transpose = [&](const uint32_t (&raw)[16], v4i (&bf)[4]) {
#pragma unroll
    for (int jg = 0; jg < 4; ++jg) {
      uint32_t o0, o1, o2, o3;
      transpose4x4_b8(raw[4 * jg], raw[4 * jg + 1], raw[4 * jg + 2], raw[4 * jg + 3], o0, o1, o2, o3);
      bf[0][jg] = (int)o0;
      bf[1][jg] = (int)o1;
      bf[2][jg] = (int)o2;
      bf[3][jg] = (int)o3;
    }
  };

  PLHIP_STAMP(3);
  uint32_t raw[16];
  v4i af_cur[MA], af_nxt[MA], bf_cur[4], bf_nxt[4];
  // K-step 0 into registers: AHEAD-1 younger K-steps of PER pieces each.  The scale / bias loads behind them are NOT
  // counted: there are 0, 1 or 2 of them (no bias pointer, int32 output), and a count that is too high by one lets the
  // wave run ahead of its own last piece (seen as an intermittent wrong 32-row tile); too low only waits a little longer.
  wait_vmcnt<(AHEAD - 1) * PER>();
  __builtin_amdgcn_s_barrier();
  read_slot(0, raw, af_cur);
  transpose(raw, bf_cur);

  // one iteration; YOUNGER = my K-steps issued after ks+1 that may still be in flight at the wait, ISSUE / NEXT: whether
  // K-step ks+AHEAD / ks+1 exists (compile-time in the steady state and in the peeled tail)
  int rslot = 1, islot = AHEAD % GD_NS;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto step = [&](int ks, auto younger_c, auto issue_c, auto next_c, auto ring_c, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;  // K-step 0 of the straight-line (AREG) loop
    constexpr int RI = AREG ? decltype(ring_c)::value : 0;  // AREG: K-step ks lives in aring[ks % 4]
    constexpr int YOUNGER = decltype(younger_c)::value;
    constexpr bool ISSUE = decltype(issue_c)::value;
    constexpr bool NEXT = decltype(next_c)::value;
    if (ks < STAMP_SLOTS - 8) PLHIP_STAMP(4 + ks);
    const bool sub = diag && (g.dbg & 64) && ks == 6;  // sub-stamps of one steady-state K-step (they perturb it: the
                                                        // s_memtime results force lgkmcnt(0), i.e. wait for the LDS reads)
    if (NEXT) {
      wait_vmcnt<YOUNGER * PER>();
      if (sub && lane == 0) lstamp[20] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();  // K-step ks+1 complete for everyone; nobody reads K-step ks-1's slot any more
      if (sub && lane == 0) lstamp[21] = __builtin_amdgcn_s_memtime();
      read_slot(rslot, raw, af_nxt);
      rslot = rslot + 1 == GD_NS ? 0 : rslot + 1;
      if (sub && lane == 0) lstamp[22] = __builtin_amdgcn_s_memtime();
    }
    if (ISSUE) {
      issue(ks + AHEAD, islot);
      islot = islot + 1 == GD_NS ? 0 : islot + 1;
    }
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[a][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AREG ? aring[RI][a] : af_cur[a], bf_cur[i],
                                                          (AREG && FIRST) ? zero16 : acc[a][i], 0, 0, 0);
    if (NEXT) transpose(raw, bf_nxt);
    // schedule: [MFMA + one DMA piece] x (pieces issued here), bare MFMAs, then the remaining MFMAs share the transposes
    constexpr int NM = 4 * MA;
    constexpr int NDMA = AREG ? 1 : PER;  // LDS-DMA pieces issued among the first MFMAs
    constexpr int LEAD = PER;             // MFMAs in front of the first transpose (LDS read latency)
#pragma unroll
    for (int q = 0; q < LEAD; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ISSUE && q < NDMA) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // address arithmetic of the piece
      if (ISSUE && q < NDMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
#pragma unroll
    for (int q = LEAD; q < NM; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (32 + NM - LEAD - 1) / (NM - LEAD), 0);
    }
    if (sub && lane == 0) lstamp[23] = __builtin_amdgcn_s_memtime();
    // AREG: the fragments of K-step ks+AHEAD replace the ones just consumed.  At the END of the step: an asm statement
    // closes the scheduling region, and the MFMA / v_perm interleave above must stay in one region.
    if (AREG && ISSUE) load_a_regs(ks + AHEAD, aring[RI]);
    if (NEXT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bf_cur[i] = bf_nxt[i];
      if (!AREG) {
#pragma unroll
        for (int a = 0; a < MA; ++a) af_cur[a] = af_nxt[a];
      }
    }
  };
  using std::integral_constant;
  typedef integral_constant<bool, true> T_;
  typedef integral_constant<bool, false> F_;
  if (AREG) {
    // unrolled by the ring size: K-step ks uses aring[ks % 4]; KS == 4 * NG, so the last group is the peeled tail
    if (!(g.dbg & 2)) {
      int ks = 0;
      // (the very first step multiplies into a zero addend: FIRST)
      if (NG > 1) {
        step(0, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 0>{}, T_{});
        step(1, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 1>{}, F_{});
        step(2, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 2>{}, F_{});
        step(3, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 3>{}, F_{});
        ks = 4;
      }
#pragma unroll
      for (int gi = 1; gi + 1 < NG; ++gi, ks += 4) {
        step(ks, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 0>{}, F_{});
        step(ks + 1, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 1>{}, F_{});
        step(ks + 2, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 2>{}, F_{});
        step(ks + 3, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, 3>{}, F_{});
      }
      if (NG > 1) step(ks, integral_constant<int, 2>{}, F_{}, T_{}, integral_constant<int, 0>{}, F_{});
      else step(ks, integral_constant<int, 2>{}, F_{}, T_{}, integral_constant<int, 0>{}, T_{});
      step(ks + 1, integral_constant<int, 1>{}, F_{}, T_{}, integral_constant<int, 1>{}, F_{});
      step(ks + 2, integral_constant<int, 0>{}, F_{}, T_{}, integral_constant<int, 2>{}, F_{});
      step(ks + 3, integral_constant<int, 0>{}, F_{}, F_{}, integral_constant<int, 3>{}, F_{});
    }
  } else {
    const int kmain = (g.dbg & 2) ? 0 : KS - AHEAD;
    for (int ks = 0; ks < kmain; ++ks) step(ks, integral_constant<int, AHEAD - 2>{}, T_{}, T_{}, integral_constant<int, 0>{}, F_{});
    if (!(g.dbg & 2)) {
      // peeled tail: K-steps KS-AHEAD .. KS-1, nothing left to issue, the in-flight count shrinks
      static_assert(AHEAD >= 2 && AHEAD <= 8, "tail is written for 2..8 K-steps ahead");
      int ks = KS - AHEAD;
#define PLHIP_TAIL(T)                                                                                                    \
  if (AHEAD - 1 > T) {                                                                                                   \
    step(ks, integral_constant<int, (AHEAD - 2 - T > 0 ? AHEAD - 2 - T : 0)>{}, F_{}, T_{}, integral_constant<int, 0>{}, F_{}); \
    ++ks;                                                                                                                \
  }
      PLHIP_TAIL(0) PLHIP_TAIL(1) PLHIP_TAIL(2) PLHIP_TAIL(3) PLHIP_TAIL(4) PLHIP_TAIL(5) PLHIP_TAIL(6)
#undef PLHIP_TAIL
      step(ks, integral_constant<int, 0>{}, F_{}, F_{}, integral_constant<int, 0>{}, F_{});
    }
  }

  if (OUT != OUT_I32) store_scale_bias<MA, OUT>(lsb, lane, my_s, my_b);
  PLHIP_STAMP(STAMP_SLOTS - 4);
  if (nvalid && mactive && !(g.dbg & 1)) {
    if (OUT == OUT_I32) {
      gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw, skip);
    } else {
      switch (g.act) {
        case ACT_RELU: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU>(g, acc, mt, h, b, hw, lsb, g.HWY - hw, skip); break;
        case ACT_RELU6: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, g.HWY - hw, skip); break;
        case ACT_LEAKY: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, g.HWY - hw, skip); break;
        default: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw, skip); break;
      }
    }
  }
  if (diag) {  // wave-uniform
    PLHIP_STAMP(STAMP_SLOTS - 3);  // epilogue instructions issued
    wait_vmcnt<0>();
    if (lane == 0) {
      lstamp[STAMP_SLOTS - 2] = __builtin_amdgcn_s_memtime();  // stores acknowledged
      lstamp[STAMP_SLOTS - 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (blockIdx.x < 1024 && lane < STAMP_SLOTS) g_stamps[((size_t)blockIdx.x * 4 + wave) * STAMP_SLOTS + lane] = lstamp[lane];
  }
}

// =====================================================================================================================
// Wave-specialised variant (producer / consumer) for the MFMA-heavy layers.
// PMC + ablation on the ring kernel: per K-step every wave ran the chain  barrier -> 18 LDS reads -> 32 v_perm -> 8 MFMA,
// and with <= 2 waves per SIMD nothing overlapped it (MFMA busy 13 %); a deeper ring changed nothing.  Here the roles are
// split so that the matrix pipes only ever see ds_read_b128 + MFMA:
//   * waves 4,5 = PRODUCERS.  Producer p owns k-rows {8p..8p+7, 16+8p..16+8p+7} of every K-step: it fetches them by
//     LDS-DMA into a PRIVATE ring (5 K-steps in flight, counted vmcnt, no cross-wave dependency on raw data), reads
//     them back, transposes (16 v_perm) and writes its 8-byte half of the four B fragments into the shared, double
//     buffered fragment area;
//   * waves 0-3 = CONSUMERS (64 x 128 outputs each): 4 ds_read_b128 for B, A fragments straight from L2 through a
//     4-deep register ring (their only vector-memory traffic, so the in-order vmcnt never couples to anything slow),
//     8 MFMAs per K-step;
//   * one s_barrier per K-step hands fragment buffer (ks+1)&1 to the consumers and buffer ks&1 back to the producers.
#ifdef PLHIP_EXPERIMENTS  // make EXPERIMENTS=1 (PLHIP_GEMM_VARIANT=5): slower than the ring kernel, kept as a record
#define WS_D 5   // K-steps a producer keeps in flight
#define WS_NS 6  // slots of its private raw ring

template <int OUT, bool VEC_STORE, bool MFULL>
__global__ __launch_bounds__(384, 2) void gemm_i8_ws_kernel(GemmArgs g) {
  constexpr int MA = 2;
  // ONE LDS object: [2 fragment buffers x 4 KiB][2 producers x WS_NS x 2 KiB raw rings][4 consumers x scale/bias]
  __shared__ __attribute__((aligned(16))) uint8_t sm[2 * 4096 + 2 * WS_NS * 2048 + 4 * 2 * MA * 32 * 4];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mtb_n = (g.MT + 3) >> 2;
  int mtb, nt;
  xcd_tile_map(blockIdx.x, mtb_n, g.NT, mtb, nt);
  if (nt >= g.NT) return;  // block-uniform
  const int c = lane & 31, h = lane >> 5;
  const int ntot = g.NB * g.HWX;
  int n4 = nt * 128 + 4 * c;
  const bool nvalid = n4 < ntot;
  if (!nvalid) n4 = 0;
  const int b = n4 / g.HWX;
  const int hw = n4 - b * g.HWX;
  const int KS = g.KS;
  v4i* frags = reinterpret_cast<v4i*>(sm);  // [buf][i][lane]

  if (wave >= 4) {
    // ------------------------------------------------------------------ producer
    const int p = wave - 4;
    uint8_t* ring = sm + 2 * 4096 + p * WS_NS * 2048;
    const int8_t* xb = g.x + (size_t)b * g.x_bstride + hw;
    auto issue = [&](int ks, int slot) {
      uint8_t* sb = ring + slot * 2048;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int lrow = 2 * q;                                   // local rows lrow, lrow+1 (lanes 0-31 / 32-63)
        const int krow = (q < 4 ? 8 * p : 16 + 8 * p) + 2 * (q & 3);  // k-row inside the K-step
        int k = ks * 32 + krow + h;
        k = k < g.K ? k : g.K - 1;
        __builtin_amdgcn_global_load_lds((glb_ptr)(xb + (size_t)k * g.XP), (lds_ptr)(sb + lrow * 128), 4, 0, 0);
      }
    };
    auto transpose_to = [&](int slot, int buf) {
      const uint8_t* sb = ring + slot * 2048;
      uint32_t raw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const uint32_t*>(sb + (8 * h + j) * 128 + 4 * c);  // k = 16h + 8p + j
      uint32_t o[2][4];
      transpose4x4_b8(raw[0], raw[1], raw[2], raw[3], o[0][0], o[0][1], o[0][2], o[0][3]);
      transpose4x4_b8(raw[4], raw[5], raw[6], raw[7], o[1][0], o[1][1], o[1][2], o[1][3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // dwords 2p, 2p+1 of lane's 16-byte entry of fragment i
        uint2 v = make_uint2(o[0][i], o[1][i]);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(frags + (buf * 4 + i) * 64 + lane) + 8 * p) = v;
      }
    };
    auto wait_landed = [&](int younger) {
      switch (younger) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<8>(); break;
        case 2: wait_vmcnt<16>(); break;
        case 3: wait_vmcnt<24>(); break;
        default: wait_vmcnt<32>(); break;
      }
    };
#pragma unroll
    for (int t = 0; t < WS_D; ++t)
      if (t < KS) issue(t, t);
    {
      const int last = WS_D - 1 < KS - 1 ? WS_D - 1 : KS - 1;
      wait_landed(last);
      transpose_to(0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int islot = WS_D % WS_NS, rslot = 1;
    const int KSP = (KS + 3) & ~3;  // same number of hand-over barriers as the consumers
    for (int ks = 0; ks < KSP; ++ks) {
      if (ks + WS_D < KS) issue(ks + WS_D, islot);
      islot = islot + 1 == WS_NS ? 0 : islot + 1;
      if (ks + 1 < KS && !(g.dbg & 16)) {
        const int last = ks + WS_D < KS - 1 ? ks + WS_D : KS - 1;
        wait_landed(last - (ks + 1));
        transpose_to(rslot, (ks + 1) & 1);
        rslot = rslot + 1 == WS_NS ? 0 : rslot + 1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // -------------------------------------------------------------------- consumer
  const int mt = mtb * 4 + wave;
  const bool mactive = mt < g.MT;
  const int mtc = mactive ? mt : g.MT - 1;
  float* lsb = reinterpret_cast<float*>(sm + 2 * 4096 + 2 * WS_NS * 2048) + wave * 2 * MA * 32;
  v16i acc[MA][4];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0;
  v4i a0[MA], a1[MA], a2[MA], a3[MA];
  auto load_a_c = [&](int ks, v4i (&af)[MA]) {
    if ((g.dbg & 8) && ks > 1) return;  // timing experiment: no A traffic inside the loop
    load_a<MA>(g.wp, mtc, KS, ks < KS ? ks : KS - 1, lane, af);
  };
  load_a_c(0, a0);
  load_a_c(1, a1);
  float my_s = 1.f, my_b = 0.f;
  if (OUT != OUT_I32) load_scale_bias<MA>(g, mtc, lane, my_s, my_b);
  auto kstep = [&](int buf, const v4i (&af)[MA]) {
    v4i bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bf[i] = frags[(buf * 4 + i) * 64 + lane];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[a][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[a], bf[i], acc[a][i], 0, 0, 0);
  };
  __builtin_amdgcn_s_barrier();  // fragment buffer 0 is ready
  // K loop unrolled by 4 so that the A ring uses static register names; every K-step ends with the hand-over barrier.
  // Both roles run KSP = roundup(KS, 4) rounds (the surplus ones are empty) so that the barrier counts always match.
  const int KSP = (KS + 3) & ~3;
  for (int ks0 = 0; ks0 < KSP; ks0 += 4) {
    load_a_c(ks0 + 2, a2);
    kstep(0, a0);  // ks0 < KS always
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_a_c(ks0 + 3, a3);
    if (ks0 + 1 < KS) kstep(1, a1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_a_c(ks0 + 4, a0);
    if (ks0 + 2 < KS) kstep(0, a2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_a_c(ks0 + 5, a1);
    if (ks0 + 3 < KS) kstep(1, a3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (OUT != OUT_I32) store_scale_bias<MA, OUT>(lsb, lane, my_s, my_b);
  if (!nvalid || !mactive || (g.dbg & 1)) return;
  if (OUT == OUT_I32) {
    gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw);
    return;
  }
  switch (g.act) {
    case ACT_RELU: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    case ACT_RELU6: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_RELU6>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    case ACT_LEAKY: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_LEAKY>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
    default: gemm_epilogue<MA, OUT, VEC_STORE, MFULL, ACT_NONE>(g, acc, mt, h, b, hw, lsb, g.HWY - hw); break;
  }
}

#endif  // PLHIP_EXPERIMENTS

// ---- weight pre-pack: [G][Mg][Kg] row-major (OIHW flattened) -> [G][MT32][KS][64 lanes][16 B] ----
// lane (r = lane&31, h = lane>>5), byte j  <-  W[g][mt32*32 + r][ks*32 + 16h + j]   (0 outside).
__global__ void pack_weights_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, int G, int Mg, int Kg,
                                    int MT32, int KS) {
  const size_t total = (size_t)G * MT32 * KS * 1024;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 15;
    const int lane = (idx >> 4) & 63;
    size_t t = idx >> 10;
    const int ks = t % KS;
    t /= KS;
    const int mt32 = t % MT32;
    const int grp = (int)(t / MT32);
    const int m = mt32 * 32 + (lane & 31);
    const int k = ks * 32 + 16 * (lane >> 5) + j;
    int8_t v = 0;
    if (m < Mg && k < Kg) v = w[((size_t)grp * Mg + m) * Kg + k];
    wp[idx] = v;
  }
}

// ---- im2col: x NCHW -> col[b][g][Kg][Np], Np = roundup(oh*ow, 4), pad columns and OOB taps = 0 ----
// Row index k = c*kh*kw + r*kw + q (conv_impl.cc:103-153).  One thread writes one dword (4 columns).
// grid = (column-quad tiles, Kg, batch*groups): the row (image, group, channel, tap) is block-uniform, so its decode runs
// on the scalar unit; a thread does ONE 32-bit division (its first column -> (oy, ox)) and walks the other three columns
// with a carry.  Stride-1 quads that stay inside one input row are fetched as one unaligned dword.  (The former
// 1-D form decoded everything per thread with 64-bit divisions: 128 us for the 57.8 MB buffer of BASELINE config #2.)
__global__ __launch_bounds__(256) void im2col_i8_kernel(Im2colArgs a) {
  const int np4 = a.Np >> 2;
  const int q4 = blockIdx.x * 256 + threadIdx.x;
  if (q4 >= np4) return;
  const int k = blockIdx.y;
  const int bg = blockIdx.z;
  const int b = bg / a.G, grp = bg - b * a.G;
  const int khkw = a.kh * a.kw;
  const int ci = k / khkw, rs = k - ci * khkw;
  const int kr = rs / a.kw, kq = rs - kr * a.kw;
  const int8_t* xp = a.x + ((size_t)b * a.cin + (size_t)grp * a.cin_g + ci) * a.h * a.w;
  const size_t row = (size_t)bg * a.Kg + k;
  int n = q4 * 4;
  int oy = (int)((uint32_t)n / (uint32_t)a.ow), ox = n - oy * a.ow;
  uint32_t out = 0;
  const int ih0 = oy * a.sh - a.pt + kr * a.dh, iw0 = ox * a.sw - a.pl + kq * a.dw;
  if (a.sw == 1 && n + 3 < a.N && ox + 3 < a.ow && ih0 >= 0 && ih0 < a.h && iw0 >= 0 && iw0 + 3 < a.w) {
    __builtin_memcpy(&out, xp + (size_t)ih0 * a.w + iw0, 4);
  } else if (a.sw == 2 && n + 3 < a.N && ox + 3 < a.ow && ih0 >= 0 && ih0 < a.h && iw0 >= 0 && iw0 + 7 < a.w) {
    // stride 2 (ResNet50's 1x1 stride-2 shortcuts): columns iw0, +2, +4, +6 of one row: one unaligned 8-byte fetch,
    // every other byte kept (the byte-by-byte walk below ran the three shortcut copies at ~1 TB/s)
    uint32_t d[2];
    __builtin_memcpy(d, xp + (size_t)ih0 * a.w + iw0, 8);
    out = __builtin_amdgcn_perm(d[1], d[0], 0x06040200u);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (n + i < a.N) {
        const int ih = oy * a.sh - a.pt + kr * a.dh, iw = ox * a.sw - a.pl + kq * a.dw;
        if (ih >= 0 && ih < a.h && iw >= 0 && iw < a.w) out |= (uint32_t)(uint8_t)xp[(size_t)ih * a.w + iw] << (8 * i);
      }
      if (++ox == a.ow) {
        ox = 0;
        ++oy;
      }
    }
  }
  *reinterpret_cast<uint32_t*>(a.col + row * a.Np + (size_t)q4 * 4) = out;
}

// ---- host-side launchers (called from plhip_capi.hip) ----
int debug_read_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(unsigned long long) * 1024 * 4 * STAMP_SLOTS) bytes = sizeof(unsigned long long) * 1024 * 4 * STAMP_SLOTS;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

static int gemm_variant() {  // PLHIP_GEMM_VARIANT: 0 auto, 1 private-tile kernel, 2 register-staged LDS kernel, 3 LDS-DMA ring, 5 wave-specialised
  const int v = knob("GEMM_VARIANT", 0);
  return v;
}

template <int MA, int OUT>
static void launch_gemm_t(const GemmArgs& g_in, bool vec_store, bool aligned, hipStream_t s) {
  GemmArgs g = g_in;
  const bool mfull = g.M % (32 * MA) == 0;
  const int var = gemm_variant();
#ifdef PLHIP_EXPERIMENTS
  const bool use_ws = MA == 2 && aligned && var == 5 && g.im_kw == 0;  // experiment, opt-in: slower than the ring
  if (use_ws) {
    const unsigned blocks = (unsigned)(((g.MT + 3) / 4) * (long)((g.NT + 7) / 8 * 8));
    if (vec_store && mfull)
      hipLaunchKernelGGL((gemm_i8_ws_kernel<OUT, true, true>), dim3(blocks), dim3(384), 0, s, g);
    else if (vec_store)
      hipLaunchKernelGGL((gemm_i8_ws_kernel<OUT, true, false>), dim3(blocks), dim3(384), 0, s, g);
    else
      hipLaunchKernelGGL((gemm_i8_ws_kernel<OUT, false, false>), dim3(blocks), dim3(384), 0, s, g);
    return;
  }
#endif
  // (32-row wave tiles with a short K -- e.g. 128->128 at 56x56 -- run faster on the register-staged kernel: 24.8 vs 26.6 us)
  const bool use_dma = g.im_kw > 0 ||  // the implicit-GEMM route exists only in the ring kernel (conv_geom checked the shape)
                       (g.HWX >= 16 && g.KS >= 4 && (var == 3 || (var == 0 && g.MT >= 4 && (MA == 2 || g.KS >= 8))));
  if (use_dma) {
    // dense NCHW slabs whose rows are not a multiple of 4 bytes (HW = 49: the 7x7 layers) arrive with HWX rounded up to 4
    // for the dword kernels; this kernel moves END-aligned 16-byte pieces and must know the TRUE row length, or the last
    // piece of a row reaches HWX - XP bytes into the next row -- and past the end of the tensor on its last row
    if (g.im_kw == 0 && g.XP > 0 && g.XP < g.HWX) g.HWX = g.XP;
    g.NT = (int)(((long)g.NB * ((g.HWX + 15) & ~15) + 127) / 128);  // 16-byte padded column space of this kernel
    const unsigned blocks = (unsigned)(((g.MT + 3) / 4) * (long)((g.NT + 7) / 8 * 8));
    const int areg_env = knob("GEMM_AREG", 1);
    const int ng = (areg_env && (g.KS & 3) == 0 && mfull && MA == 2) ? g.KS >> 2 : 0;
    const bool areg = ng == 1 || ng == 2 || ng == 4 || ng == 8;  // K = 128 / 256 / 512 / 1024
    const size_t lds = (size_t)(4 + 1) * (areg ? 4096 : 4096 + 4 * MA * 1024) + 4 * 2 * MA * 32 * 4 + 4 * STAMP_SLOTS * 8;
#define PLHIP_LAUNCH_DMA2(VS, MF, NGV)                                                                            \
  do {                                                                                                            \
    auto kfn = gemm_i8_dma_kernel<MA, OUT, VS, MF, 4, NGV>;                                                       \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, s, g);                                                  \
  } while (0)
#define PLHIP_LAUNCH_DMA(VS, MF)                                            \
  do {                                                                      \
    if (MA == 2 && MF && areg) {                                            \
      if (ng == 1) PLHIP_LAUNCH_DMA2(VS, MF, (MA == 2 && MF) ? 1 : 0);      \
      else if (ng == 2) PLHIP_LAUNCH_DMA2(VS, MF, (MA == 2 && MF) ? 2 : 0); \
      else if (ng == 4) PLHIP_LAUNCH_DMA2(VS, MF, (MA == 2 && MF) ? 4 : 0); \
      else PLHIP_LAUNCH_DMA2(VS, MF, (MA == 2 && MF) ? 8 : 0);              \
    } else {                                                                \
      PLHIP_LAUNCH_DMA2(VS, MF, 0);                                         \
    }                                                                       \
  } while (0)
    if (vec_store && mfull) PLHIP_LAUNCH_DMA(true, true);
    else if (vec_store) PLHIP_LAUNCH_DMA(true, false);
    else if (mfull) PLHIP_LAUNCH_DMA(false, true);
    else PLHIP_LAUNCH_DMA(false, false);
#undef PLHIP_LAUNCH_DMA2
#undef PLHIP_LAUNCH_DMA
    return;
  }
  const bool use_lds = var == 2 || (var == 0 && g.MT >= 2 && g.KS >= 2);
  if (use_lds) {
    const unsigned blocks = (unsigned)(((g.MT + 3) / 4) * (long)((g.NT + 7) / 8 * 8));
    if (!aligned)
      hipLaunchKernelGGL((gemm_i8_lds_kernel<MA, OUT, false, false, false>), dim3(blocks), dim3(256), 0, s, g);
    else if (vec_store && mfull)
      hipLaunchKernelGGL((gemm_i8_lds_kernel<MA, OUT, true, true, true>), dim3(blocks), dim3(256), 0, s, g);
    else if (vec_store)
      hipLaunchKernelGGL((gemm_i8_lds_kernel<MA, OUT, true, false, true>), dim3(blocks), dim3(256), 0, s, g);
    else
      hipLaunchKernelGGL((gemm_i8_lds_kernel<MA, OUT, false, false, true>), dim3(blocks), dim3(256), 0, s, g);
    return;
  }
  const long waves = (long)g.MT * g.NT;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  if (!aligned)
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, false, false, false>), dim3(blocks), dim3(256), 0, s, g);
  else if (vec_store && mfull)
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, true, true, true>), dim3(blocks), dim3(256), 0, s, g);
  else if (vec_store)
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, true, false, true>), dim3(blocks), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_i8_nchw_kernel<MA, OUT, false, false, true>), dim3(blocks), dim3(256), 0, s, g);
}

// returns 0, or -3 when the shape exists on one kernel only and that kernel declines it (nothing is launched)
int launch_gemm_i8(const GemmArgs& g_in, int ma, int out, bool vec_store, bool aligned_loads, hipStream_t s) {
  if (!aligned_loads) vec_store = false;
  // The packed layout is a sequence of 32-row fragment tiles, so a layer packed for MA = 2 can also run with MA = 1
  // (32-row wave tiles): do so for M <= 128, where 64-row tiles would leave waves of the 4-wave block without work.
  const int ma_env = knob("GEMM_MA", 0);
  GemmArgs g = g_in;
  const int dbg_env = knob("GEMM_DEBUG", 0);
  g.dbg = dbg_env;
  // The transposed-read ring kernel (gemm_tr_i8.hip) is the implicit-GEMM engine (any M > 32, rows down to 7 columns).
  // For plain 1x1 / im2col GEMMs it is opt-in (PLHIP_GEMM_TR=2): measured on MobileNetV1's pointwise layers it ties
  // the first-generation ring kernel at batch 128 and loses at batch 256 (both are bound by the
  // ~21 B/clk a CU ingests through LDS-DMA and by the non-overlapped epilogue, not by the K loop's instruction mix).
  // It moves END-aligned 16-byte pieces and must know the TRUE row length of a dense slab (HW = 49).
  // Third generation (gemm_wide_i8.hip): plain 1x1 GEMMs with M >= 256 and K in {128, 256, 512, 1024}: one 256 x (128..256)
  // tile per CU, the weight panel read once per CU, every operand byte in flight before the first MFMA.
  if (g.im_kw == 0 && gemm_variant() == 0 && (dbg_env & ~32) == 0) {
    GemmArgs t = g;
    if (t.XP > 0 && t.XP < t.HWX) t.HWX = t.XP;  // the TRUE row length of a dense slab
    if (launch_gemm_wide(t, out, s)) return 0;
  }
  // (stride-2 / short-row implicit GEMMs exist on the transposed-read kernel ONLY: the timing bits of PLHIP_GEMM_DEBUG must
  // not send them to a first-generation kernel, which would read outside its operands: a GPU memory fault, seen once)
  const bool tr_only = g.im_kw > 0 && (g.im_s == 2 || g.HWX < 16);
  if (g.M > 32 && (g.im_kw > 0 || (gemm_variant() == 0 && gemm_tr_enabled() >= 2)) && ((dbg_env & ~96) == 0 || tr_only)) {
    GemmArgs t = g;
    if (t.im_kw == 0 && t.XP > 0 && t.XP < t.HWX) t.HWX = t.XP;
    if (launch_gemm_tr(t, out, s)) return 0;
    // stride-2 / short-row implicit GEMMs exist on that kernel ONLY: the first-generation kernels would read outside their
    // operands for these shapes.  conv_geom admits them under the same column-space bound launch_gemm_tr checks
    // (plhip_capi.hip), so this is a defensive error, not a fallback
    if (tr_only) return -3;
  }
  if (ma == 2 && ((ma_env == 0 && g.M <= 128 && g.M > 64) || (ma_env == 1 && g.im_kw == 0))) ma = 1;
  // 64-row tiles whose last tile is at most half full (M = 144: 192 rows computed and stored-checked for 144), and the
  // streaming shapes with K <= 64 and M > 64 (MobileNetV2's expand convs: 24 -> 144 ran at 3.0 TB/s with 64-row tiles,
  // 3.9 with 32-row ones; 64 -> 384 @14x14 12.1 -> 9.8 us): 32-row wave tiles
  // (K = 32, M = 64 — MobileNetV1's first pointwise conv — too: 28.7 -> 27.4 us at batch 128)
  if (ma == 2 && ma_env == 0 && g.im_kw == 0 && (((g.M & 63) != 0 && (g.M & 63) <= 32) || (g.KS <= 2 && g.M > 64) || (g.KS == 1 && g.M == 64))) ma = 1;
  g.MT = (g.M + 32 * ma - 1) / (32 * ma);
  if (ma == 1) {
    if (out == OUT_I32) launch_gemm_t<1, OUT_I32>(g, vec_store, aligned_loads, s);
    else if (out == OUT_F32) launch_gemm_t<1, OUT_F32>(g, vec_store, aligned_loads, s);
    else launch_gemm_t<1, OUT_I8>(g, vec_store, aligned_loads, s);
  } else {
    if (out == OUT_I32) launch_gemm_t<2, OUT_I32>(g, vec_store, aligned_loads, s);
    else if (out == OUT_F32) launch_gemm_t<2, OUT_F32>(g, vec_store, aligned_loads, s);
    else launch_gemm_t<2, OUT_I8>(g, vec_store, aligned_loads, s);
  }
  return 0;
}

void launch_pack_weights(const int8_t* w, int8_t* wp, int G, int Mg, int Kg, int MT32, int KS, hipStream_t s) {
  const size_t total = (size_t)G * MT32 * KS * 1024;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wp, G, Mg, Kg, MT32, KS);
}

// Zero-padded copy of the input for the implicit-GEMM route: xp[plane][ph][pw] = x[plane][ph - pt][pw - pl] or 0.
// One thread = one aligned dword of the flat padded buffer (two divisions, then carry propagation byte by byte).
__global__ void pad_input_i8_kernel(PadArgs a) {
  const long nq = a.total >> 2;
  const int plane_sz = a.ph * a.pw;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    // the padded buffer is < 2^31 bytes (conv_geom): magic-number divisions (two hardware divide sequences per dword made
    // this copy VALU-bound: 35 us for ResNet50's 55 MB res2 planes)
    const uint32_t o = (uint32_t)q << 2;
    int plane = (int)fastdiv_u31(o, a.div_plane_m, a.div_plane_s);
    const int rem = (int)(o - (uint32_t)plane * (uint32_t)plane_sz);
    int ph = (int)fastdiv_u31((uint32_t)rem, a.div_pw_m, a.div_pw_s), pw = rem - ph * a.pw;
    uint32_t v = 0;
    {  // interior dword (the common case): one unaligned 4-byte load
      const int ih = ph - a.pt, iw = pw - a.pl;
      if (plane < a.planes && pw + 3 < a.pw && ih >= 0 && ih < a.h && iw >= 0 && iw + 3 < a.w) {
        __builtin_memcpy(&v, a.x + ((size_t)plane * a.h + ih) * a.w + iw, 4);
        reinterpret_cast<uint32_t*>(a.xp)[q] = v;
        continue;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ih = ph - a.pt, iw = pw - a.pl;
      if (plane < a.planes && ih >= 0 && ih < a.h && iw >= 0 && iw < a.w)
        v |= (uint32_t)(uint8_t)a.x[((size_t)plane * a.h + ih) * a.w + iw] << (8 * i);
      if (++pw == a.pw) {
        pw = 0;
        if (++ph == a.ph) {
          ph = 0;
          ++plane;
        }
      }
    }
    reinterpret_cast<uint32_t*>(a.xp)[q] = v;
  }
}

// Phase-split padded copy for the stride-2 implicit GEMM: xp[plane][p][q][y][x] = padded[plane][2y + p][2x + q].
// One thread = one aligned dword (4 consecutive x of one phase row): 4 source bytes at stride 2.
__global__ void pad_input_phase2_i8_kernel(PadArgs a) {
  const long nq = a.total >> 2;
  const int pwq = a.pw >> 2;  // launcher: phase rows are padded to a multiple of 4 columns
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    // q < 2^29 (the buffer is < 2^31 bytes): 32-bit magic-number divisions, no 64-bit divide sequences
    const uint32_t t1 = fastdiv_u31((uint32_t)q, a.div_pwq_m, a.div_pwq_s);
    const int xq = (int)((uint32_t)q - t1 * (uint32_t)pwq);
    const uint32_t t2 = fastdiv_u31(t1, a.div_ph_m, a.div_ph_s);
    const int y = (int)(t1 - t2 * (uint32_t)a.ph);
    const int ph = (int)(t2 & 3);
    const long plane = (long)(t2 >> 2);
    uint32_t v = 0;
    if (plane < a.planes) {
      const int iy = 2 * y + (ph >> 1) - a.pt;
      if (iy >= 0 && iy < a.h) {
        const int8_t* row = a.x + ((size_t)plane * a.h + iy) * a.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ix = 2 * (4 * xq + i) + (ph & 1) - a.pl;
          if (ix >= 0 && ix < a.w) v |= (uint32_t)(uint8_t)row[ix] << (8 * i);
        }
      }
    }
    reinterpret_cast<uint32_t*>(a.xp)[q] = v;
  }
}

static void pad_magic(long d, unsigned& m, int& sh) {  // fastdiv_u31's (magic, shift) for divisor d (dw_common.h)
  int l = 0;
  while ((1L << l) < d) ++l;
  if ((1L << l) == d) {
    m = 0;
    sh = l;
    return;
  }
  m = (unsigned)(((1ULL << (31 + l)) / (unsigned long long)d) + 1ULL);
  sh = l - 1;
}

void launch_pad_input(const PadArgs& a_in, hipStream_t s) {
  PadArgs a = a_in;
  pad_magic((long)a.ph * a.pw, a.div_plane_m, a.div_plane_s);
  pad_magic(a.pw, a.div_pw_m, a.div_pw_s);
  pad_magic(a.pw >> 2 > 0 ? a.pw >> 2 : 1, a.div_pwq_m, a.div_pwq_s);
  pad_magic(a.ph, a.div_ph_m, a.div_ph_s);
  if (a.stride == 2) {
    long blocks2 = ((a.total >> 2) + 255) / 256;
    if (blocks2 > 65536) blocks2 = 65536;
    hipLaunchKernelGGL(pad_input_phase2_i8_kernel, dim3((unsigned)blocks2), dim3(256), 0, s, a);
    return;
  }
  long blocks = ((a.total >> 2) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(pad_input_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// 1x1 stride-2 convs (ResNet50's downsampling shortcuts: lite/backends/arm/math/conv_impl.cc:490-598 runs them through
// im2col too): the "im2col" is a strided gather, col[b][c][oy * ow + ox] = x[b][c][2 oy][2 ox].  One thread = 16 output
// bytes = 4 quads, each ONE unaligned 8-byte fetch with every other byte kept, one 16-byte store (the dword-per-thread
// form above ran the three shortcut copies of a ResNet50 step at 2.3 TB/s, 0.33 ms per 256 images: 4-byte stores).
// Needs kh = kw = 1, no padding in use, sw = 2.
__global__ __launch_bounds__(256) void subsample2_1x1_i8_kernel(Im2colArgs a) {
  // flat index -> (row = (image, group, channel), 16-byte chunk): a (chunks, channels, images) grid of mostly empty 256-thread
  // blocks (49 chunks per 28x28 row) was bound by the workgroup dispatch rate: 65 k blocks for 51 MB
  const int nch = (a.Np + 15) >> 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.rows * (size_t)nch) return;
  const uint32_t rowi = (uint32_t)(idx / (uint32_t)nch);
  const int q16 = (int)(idx - (size_t)rowi * nch);
  const int bg = (int)(rowi / (uint32_t)a.Kg), k = (int)(rowi - (uint32_t)bg * a.Kg);
  const int b = bg / a.G, grp = bg - b * a.G;
  const int8_t* xp = a.x + ((size_t)b * a.cin + (size_t)grp * a.cin_g + k) * a.h * a.w;
  const size_t row = (size_t)bg * a.Kg + k;
  const int n0 = q16 * 16;
  int oy = (int)((uint32_t)n0 / (uint32_t)a.ow), ox = n0 - oy * a.ow;
  uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int n = n0 + 4 * d;
    if (n + 3 < a.N && ox + 3 < a.ow && ox * 2 + 7 < a.w) {  // the quad inside one output row, its 8 source bytes inside the input row
      uint32_t dd[2];
      __builtin_memcpy(dd, xp + (size_t)(oy * a.sh) * a.w + ox * 2, 8);
      out[d] = __builtin_amdgcn_perm(dd[1], dd[0], 0x06040200u);
      ox += 4;
      if (ox >= a.ow) {
        ox -= a.ow;
        ++oy;
      }
    } else {  // a quad across two output rows (14- and 7-wide planes), the row's last quad when w is odd, the plane's tail
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (n + i < a.N) out[d] |= (uint32_t)(uint8_t)xp[(size_t)(oy * a.sh) * a.w + ox * 2] << (8 * i);
        if (++ox == a.ow) {
          ox = 0;
          ++oy;
        }
      }
    }
  }
  int8_t* dst = a.col + row * a.Np + (size_t)n0;
  if (n0 + 16 <= a.Np) {
    const v4i v = {(int)out[0], (int)out[1], (int)out[2], (int)out[3]};
    __builtin_memcpy(dst, &v, 16);  // (rows are 4-byte aligned: Np % 4 == 0)
  } else {
#pragma unroll
    for (int d = 0; d < 4; ++d)
      if (n0 + 4 * d < a.Np) __builtin_memcpy(dst + 4 * d, &out[d], 4);
  }
}

void launch_im2col(const Im2colArgs& a, hipStream_t s) {
  // rows = batch * G * Kg; Kg and batch*G ride on grid.y / grid.z (<= 65535 each, checked by the caller)
  const unsigned bg = (unsigned)(a.rows / (size_t)a.Kg);
  const int sub_env = knob("SUBSAMPLE_1X1", 1);  // 0 = the generic im2col kernel (A/B runs)
  if (sub_env && a.kh == 1 && a.kw == 1 && a.pt == 0 && a.pl == 0 && a.sw == 2 && a.Kg == a.cin_g &&
      (a.oh - 1) * a.sh < a.h && (a.ow - 1) * 2 < a.w) {  // (no tap in a bottom / right padding)
    const size_t threads = a.rows * (size_t)((a.Np + 15) >> 4);
    if (threads < ((size_t)1 << 31) * 256) {
      hipLaunchKernelGGL(subsample2_1x1_i8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
      return;
    }
  }
  hipLaunchKernelGGL(im2col_i8_kernel, dim3((unsigned)(((a.Np >> 2) + 255) / 256), (unsigned)a.Kg, bg), dim3(256), 0, s, a);
}

}  // namespace plhip
