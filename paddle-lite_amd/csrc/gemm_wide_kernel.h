// gemm_wide_kernel.h — the wide-tile GEMM kernel template and its per-tile launchers; included by the per-tile translation
// units gemm_wide_n4.hip / gemm_wide_n7.hip / gemm_wide_n8.hip (one per NTT, so that they compile in parallel) and
// described in gemm_wide_i8.hip.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "gemm_tr_common.h"
#include "dw_common.h"

namespace plhip {

constexpr int WIDE_STAMP_SLOTS = 48;  // 0-9 phases, 10 + ks: top of K-step ks (behind its barrier)
#define PLHIP_WIDE_STAMP(i)                                                \
  do {                                                                     \
    if (diag && lane == 0) lstamp[i] = __builtin_amdgcn_s_memtime();       \
  } while (0)

// NTT 32-column n tiles per block (4, 7, 8), KS K-steps of 32 (K = 32 KS exactly); A0 K-steps are issued before the loop and
// R more inside every K-step until all KS are in flight (issuing everything first cost 4.4 k cycles in front of the first
// MFMA: 240 KB per CU at the ~58 B/clk the address path takes in).  NONNEG: relu / relu6 (the packed requantisation).
//
// Two phases (timeline of the one-phase form, profiles/r03_wide_timeline_v1_pw8.txt: K loop 10.2 k cycles with the VALU
// idle, then 7.5 k cycles of epilogue with the matrix pipe idle; two passes with swapped wave roles, ..._v2_pw8.txt: 17.4 k):
//   phase 1 = K-steps [0, S1) of ALL n tiles, K-outer: the steps during which operands are still arriving; it carries the
//     counted vmcnt waits and the one barrier per K-step that make them visible, and ends with everything landed;
//   phase 2 = K-steps [S1, KS), tile by tile, NO wait on memory and NO barrier: while tile t multiplies, the finished tile
//     t-1 is requantised in the MFMA shadows, SPG accumulator registers per MFMA (a slice = int -> float, fma, clamp,
//     float -> byte; every 4th packs a dword; the 16th swaps halves, stages 32 rows x 32 bytes in this wave's LDS image and
//     stores them).  Only the last tile's epilogue is exposed.
template <int NTT, int KS, int OUT, int A0, int R, bool NONNEG>
__global__ __launch_bounds__(512, 2) void gemm_i8_wide_kernel(GemmArgs g) {
  constexpr int NCH = 2 * NTT;                     // 16-column chunks per tile
  constexpr int C1 = NCH > 8 ? NCH - 8 : 0;        // chunk slots of group 1 (0: one group, two waves share a DMA piece)
  constexpr int P1 = C1 * 128;                     // bytes of a group-1 piece (group 0: 1024)
  constexpr int KSTEP = 4 * (1024 + P1);           // LDS bytes of one K-step: [group 0: 4 kg x 1024][group 1: 4 kg x P1]
  constexpr int TILE_BYTES = KS * KSTEP;
  constexpr int SP = 48;                           // staging row pitch: 32 bytes + 16 (odd multiple of 16: conflict-free row writes)
  // phase 1 = K-steps [0, S1) of the tiles [0, TK), K-outer, while the operands arrive; phase 2 = everything else, tile by
  // tile.  Measured on MobileNetV1's layers (profiles/r03_wide_timeline_v3/v5_pw8.txt, r03_ab_pw_v3_v5.txt): K >= 512: all
  // tiles, half the K-steps (phase 1 keeps the matrix pipe fed while the loads issue; 8 MFMAs per tile are left to carry
  // the epilogue slices); K <= 256: two tiles, the whole K (few K-steps: a phase 1 over all tiles would leave phase 2
  // 2-4 MFMAs per tile for 16 slices).
  constexpr int ALL_ISSUED = (KS - A0 + R - 1) / R;  // first K-step that starts with every load issued
  constexpr bool DEEP = KS >= 16;
  constexpr int TK = DEEP ? NTT : (NTT >= 6 ? 2 : 1);
  constexpr int S1 = DEEP ? (ALL_ISSUED + 2 < KS ? ALL_ISSUED + 2 : KS) : KS;
  constexpr int Q = TK * (KS - S1) + (NTT - TK) * KS;                        // MFMAs of phase 2
  // flat phase-2 index -> tile (tile-major) and K-step
  constexpr auto p2tile = [](int i) {
    int t = 0;
    while (t < NTT - 1 && i >= KS - (t < TK ? S1 : 0)) {
      i -= KS - (t < TK ? S1 : 0);
      ++t;
    }
    return t;
  };
  constexpr auto p2ks = [](int i) {
    int t = 0;
    while (t < NTT - 1 && i >= KS - (t < TK ? S1 : 0)) {
      i -= KS - (t < TK ? S1 : 0);
      ++t;
    }
    return (t < TK ? S1 : 0) + i;
  };
  constexpr int HID = 16 * (NTT - 1);              // epilogue slices to hide behind them (all tiles but the last)
  // slices done when phase-2 MFMA gi has been issued: spread evenly, never ahead of the finished tiles (the tiles below
  // the one MFMA gi - 1 belongs to)
  constexpr auto cursor = [](int gi) {
    if (gi <= 0 || Q == 0) return 0;
    int want = (int)(((long)gi * HID + Q - 1) / Q);
    int t = 0, i = gi - 1;
    while (t < NTT - 1 && i >= KS - (t < TK ? S1 : 0)) {
      i -= KS - (t < TK ? S1 : 0);
      ++t;
    }
    const int cap = 16 * t;
    return want < cap ? want : cap;
  };
  static_assert((C1 == 0 || C1 == 6 || C1 == 8) && NTT >= 2 && NTT <= 8 && A0 >= 2 && A0 <= KS && R >= 1 && R <= TK + 1 && S1 >= 2 && Q > 0, "tile");
  constexpr auto issued_before = [](int ks) { return A0 + R * ks < KS ? A0 + R * ks : KS; };  // K-steps issued when step ks starts
  PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.x); PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias);
  PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.HWX); PLHIP_PRELOAD(g.HWY); PLHIP_PRELOAD(g.XP); PLHIP_PRELOAD(g.NB);
  PLHIP_PRELOAD(g.x_bstride); PLHIP_PRELOAD(g.y_bstride); PLHIP_PRELOAD(g.MT); PLHIP_PRELOAD(g.NT);
  PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha); PLHIP_PRELOAD(g.dbg); PLHIP_PRELOAD(g.cpi_m); PLHIP_PRELOAD(g.cpi_s);
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];  // [tile: TILE_BYTES][8 staging images][stamps]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int mb, nb;
  tr_xcd_tile_map(blockIdx.x, g.MT, g.NT, mb, nb);  // g.MT / g.NT = blocks along M / N (launcher)
  if (nb >= g.NT) return;                            // block-uniform (grid padded to 8 N blocks)
  const int c = lane & 31, h = lane >> 5;
  const bool diag = (g.dbg & 32) != 0;
  unsigned long long* lstamp = reinterpret_cast<unsigned long long*>(ring + TILE_BYTES + 8 * 32 * SP) + wave * WIDE_STAMP_SLOTS;
  if (diag && lane == 0) {
    lstamp[0] = __builtin_amdgcn_s_memrealtime();
    lstamp[1] = __builtin_amdgcn_s_memtime();
    lstamp[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }

  // ---- column space: every image's HWX columns padded to HWP = roundup(HWX, 16); the last chunk of an image is
  // END-aligned (source columns HWX-16 .. HWX-1): its leading 16 - HWX%16 columns repeat the previous chunk's last ones.
  // They are computed from the same bytes, so they come out bit-identical and the chunk is STORED whole as well.
  const int HWP = (g.HWX + 15) & ~15, full16 = g.HWX & ~15;
  const int CPI = HWP >> 4;  // chunks per image

  // ---- this lane's scale / bias: ordinary loads, first and alone; consumed in phase 2
  const int mt = mb * 8 + wave;  // my 32-row m tile (wave-uniform)
  const int mrow = mt * 32 + c;
  float sc = 1.f, bi = 0.f;
  if (OUT != OUT_I32 && mrow < g.M) {
    sc = g.scale[mrow];
    if (g.bias) bi = g.bias[mrow];
  }

  // ---- my DMA piece of every K-step.  Two groups: wave w moves (group w & 1, kg = w >> 1); one group: waves 2 kg and
  // 2 kg + 1 move rows 0-3 / 4-7 of piece kg (32 lanes each).  Group 0 (and an 8-slot group 1): lane -> row q = lane >> 3,
  // slot s = lane & 7, chunk j = s ^ 2 (q >> 1).  6-slot group 1: lane < 48 -> row q = lane / 6, chunk 8 + lane % 6.
  const int kg = wave >> 1;
  int q, ch, a_ldsoff;
  bool dma_lane = true;
  if (C1 == 0) {
    q = (wave & 1) * 4 + (lane >> 3);
    dma_lane = lane < 32;
    ch = (lane & 7) ^ (2 * ((q & 7) >> 1));
    if (ch >= NCH) ch -= 2;  // NTT < 4: spare slots re-fetch a neighbour (never read)
    a_ldsoff = kg * 1024 + (wave & 1) * 512;
  } else if ((wave & 1) == 0 || C1 == 8) {
    q = lane >> 3;
    ch = (wave & 1) * 8 + ((lane & 7) ^ (2 * (q >> 1)));
    a_ldsoff = (wave & 1) * 4096 + kg * 1024;
  } else {
    q = (lane * 43) >> 8;  // lane / 6 for lane < 64
    ch = 8 + lane - q * 6;
    dma_lane = lane < 48;
    a_ldsoff = 4096 + kg * P1;
  }
  q &= 7;
  const uint8_t* asrc;
  {
    const uint32_t J = (uint32_t)nb * NCH + ch;
    uint32_t pb = fastdiv_u31(J, g.cpi_m, g.cpi_s);
    int pj = (int)(J - pb * CPI) << 4;
    if (pb >= (uint32_t)g.NB) { pb = 0; pj = 0; }  // past the last image: any legal bytes (their columns are never stored)
    const int pcol = pj < full16 ? pj : g.HWX - 16;
    asrc = reinterpret_cast<const uint8_t*>(g.x) + (size_t)pb * g.x_bstride + (size_t)(kg * 8 + q) * (uint32_t)g.XP + pcol;
  }
  const size_t astep = (size_t)32 * (uint32_t)g.XP;
  // ---- my weight fragments: [mt][ks][64 lanes][16 B]; tiles past M: any packed tile (their rows are never stored)
  const int MT32 = (g.M + 31) >> 5;
  const uint8_t* wbase = reinterpret_cast<const uint8_t*>(g.wp) + (size_t)(mt < MT32 ? mt : MT32 - 1) * (KS * 1024);  // wave-uniform
  const uint32_t wlane = lane * 16;

  v4i w[KS];
  auto issue = [&](int ks) __attribute__((always_inline)) {
    // weights: inline asm (next to LDS-DMA the compiler guards ordinary loads with vmcnt(0)); the counted waits below
    // order them.  Obligation: no register copy of an in-flight fragment (tools/check_wide_isa.py).
    const uint32_t vo = wlane + (uint32_t)(ks >> 2) * 4096u;
    switch (ks & 3) {
      case 0: asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[ks]) : "v"(vo), "s"(wbase) : "memory"); break;
      case 1: asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[ks]) : "v"(vo), "s"(wbase) : "memory"); break;
      case 2: asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(w[ks]) : "v"(vo), "s"(wbase) : "memory"); break;
      default: asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(w[ks]) : "v"(vo), "s"(wbase) : "memory"); break;
    }
    if (dma_lane)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc + (size_t)ks * astep), (lds_ptr_t)(ring + ks * KSTEP + a_ldsoff), 16, 0, 0);
  };
#pragma unroll
  for (int ks = 0; ks < A0; ++ks) issue(ks);
  PLHIP_WIDE_STAMP(3);

  // ---- transposed-read addresses: tile t <-> chunk pair (2t, 2t+1); lane 2q'+p of a 16-lane group -> row q', sub-chunk
  // p; 16-lane group parity -> chunk parity; k half h -> kg {2h, 2h+1}
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
  uint32_t tr0[4], tr1;
  {
    const int qr = (lane & 15) >> 1, par = (lane >> 4) & 1;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      tr0[tt] = ring_addr + (h * 2) * 1024 + qr * 128 + ((2 * (tt ^ (qr >> 1)) + par) * 16) + (lane & 1) * 8;
    tr1 = ring_addr + 4096 + (h * 2) * P1 + qr * 96 + par * 16 + (lane & 1) * 8;  // 6-slot group 1
  }
  constexpr int NF = TK > 3 ? TK : 3;  // fragment registers: one set per tile in phase 1, a ring of 3 sets in phase 2
  v2i lo[NF], hi[NF];
  v16i acc[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0;

  // every offset and wait count below is a constant after unrolling; fragment of (K-step KS_, tile T_) -> register set F_
#define PLHIP_WIDE_READ(KS_, T_, F_)                                                                                      \
  do {                                                                                                                     \
    constexpr bool g1_ = (T_) >= 4 && C1 == 6;                                                                             \
    constexpr int o_lo_ = (KS_) * KSTEP + (g1_ ? ((T_) - 4) * 32 : ((T_) >= 4 ? 4096 : 0));                                \
    constexpr int o_hi_ = o_lo_ + (g1_ ? P1 : 1024);                                                                       \
    const uint32_t b_ = g1_ ? tr1 : tr0[(T_) & 3];                                                                         \
    const uint32_t a_lo_ = b_ + (uint32_t)(o_lo_ & ~0xffff), a_hi_ = b_ + (uint32_t)(o_hi_ & ~0xffff);                     \
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(lo[F_]) : "v"(a_lo_), "n"(o_lo_ & 0xffff) : "memory");        \
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(hi[F_]) : "v"(a_hi_), "n"(o_hi_ & 0xffff) : "memory");        \
  } while (0)

  using std::integral_constant;
  // ---------------------------------------------------------------------------------------------------------------
  // phase 1: the whole K of the tiles [0, TK), K-outer.  Top of step ks: K-step ks+1 has landed (my loads of it: counted
  // vmcnt; everyone's: barrier); when it ends every operand byte of the block is in LDS / registers: phase 2 has no wait on
  // memory and no barrier.
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(2 * (A0 - 1)) : "memory");
  __builtin_amdgcn_s_barrier();
  PLHIP_WIDE_STAMP(4);
  {
    auto kstep = [&](auto self, auto ks_c) __attribute__((always_inline)) -> void {
      constexpr int ks = decltype(ks_c)::value;
      if constexpr (ks == 0) {  // fragments of K-step 0
        auto rd = [&](auto rself, auto t_c) __attribute__((always_inline)) -> void {
          constexpr int t = decltype(t_c)::value;
          if constexpr (t < TK) {
            PLHIP_WIDE_READ(0, t, t);
            rself(rself, integral_constant<int, t + 1>{});
          }
        };
        rd(rd, integral_constant<int, 0>{});
      }
      if constexpr (ks < S1) {
        constexpr bool NEXT = ks + 1 < S1;
        if constexpr (NEXT) {
          constexpr int younger = 2 * (issued_before(ks) - (ks + 2));
          static_assert(younger >= 0 && 2 * (issued_before(ks + 1) - ks) <= 60, "vmcnt range");
          asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[ks + 1]) : "n"(younger) : "memory");
          __builtin_amdgcn_s_barrier();
        } else if constexpr (S1 < KS) {
          static_assert(issued_before(ks) == KS, "every load is issued before the last step of phase 1");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = S1; i < KS; ++i) asm volatile("" : "+v"(w[i]));  // no use of a later fragment above this wait
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (ks < 20) PLHIP_WIDE_STAMP(10 + ks);
        auto mm = [&](auto mself, auto t_c) __attribute__((always_inline)) -> void {
          constexpr int t = decltype(t_c)::value;
          if constexpr (t < TK) {
            // fragment t of this K-step: reads issued behind it = tiles t+1.. of this step and 0..t-1 of the next
            constexpr int yl = NEXT ? 2 * (TK - 1) : 2 * (TK - 1 - t);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo[t]), "+v"(hi[t]) : "n"(yl) : "memory");
            const v4i a = {lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w[ks], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // the MFMA stays between its fragment's wait and the next read
            if constexpr (NEXT) PLHIP_WIDE_READ(ks + 1, t, t);
            if constexpr (issued_before(ks) + t < issued_before(ks + 1)) issue(issued_before(ks) + t);
            if constexpr (t == TK - 1 && issued_before(ks) + TK < issued_before(ks + 1)) issue(issued_before(ks) + TK);
            __builtin_amdgcn_sched_barrier(0);
            mself(mself, integral_constant<int, t + 1>{});
          }
        };
        mm(mm, integral_constant<int, 0>{});
        self(self, integral_constant<int, ks + 1>{});
      }
    };
    kstep(kstep, integral_constant<int, 0>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last counted wait already was vmcnt(0))
  PLHIP_WIDE_STAMP(5);

  // ---------------------------------------------------------------------------------------------------------------
  // phase 2: the flat MFMA sequence i = t * NG + j (tile t, K-step S1 + j); fragments two MFMAs ahead in a ring of 3 sets;
  // behind MFMA i the slices [j SPG, (j+1) SPG) of tile t-1's epilogue.
  const float hi2 = g.act == ACT_RELU6 ? fminf(g.alpha + g.alpha, 254.f) : 254.f;
  const float lo2 = NONNEG ? 0.f : -254.f;
  const float s2 = sc + sc, b2 = bi + bi;
  const float leak = g.act == ACT_LEAKY ? g.alpha : 1.f;                              // int8, !NONNEG: none = leaky with slope 1
  const float fcap = g.act == ACT_RELU6 ? g.alpha : __builtin_huge_valf();            // fp32: relu6 cap
  const float flo = (g.act == ACT_RELU || g.act == ACT_RELU6) ? 0.f : -__builtin_huge_valf();
  uint8_t* stg = ring + TILE_BYTES + wave * (32 * SP);
  uint32_t ebytes[4] = {0, 0, 0, 0};  // int8: the 4 results of a register group; then the packed dwords of the tile
  uint32_t edw[4] = {0, 0, 0, 0};
  float ef[4] = {0.f, 0.f, 0.f, 0.f};  // fp32: the 4 results of a register group
  size_t eoff = 0;                     // 32-bit outputs: element offset of this lane's chunk
  bool eok = false;
  auto slice = [&](auto t_c, auto r_c) __attribute__((always_inline)) {
    constexpr int t = decltype(t_c)::value, r = decltype(r_c)::value;
    if constexpr (OUT == OUT_I8) {
      float y2 = __fmaf_rn((float)acc[t][r], s2, b2);
      if constexpr (NONNEG) {
        ef[r & 3] = y2;  // the 4th slice of a register group converts and packs all four (pack4_nn_rtz: 4.25 VALU per output)
        if constexpr ((r & 3) == 3) edw[r >> 2] = pack4_nn_rtz(ef[0], ef[1], ef[2], ef[3], hi2);
      } else {
        y2 = y2 > 0.f ? y2 : leak * y2;
        const int tq = (int)__builtin_amdgcn_fmed3f(y2, lo2, hi2);
        ebytes[r & 3] = (uint32_t)((tq + 1 + (tq >> 31)) >> 1);
        if constexpr ((r & 3) == 3) edw[r >> 2] = pack4_i8((int)ebytes[0], (int)ebytes[1], (int)ebytes[2], (int)ebytes[3]);
      }
      if constexpr (r == 15) {
        // half exchange: every lane gets 16 consecutive columns of its channel row (h = 0: 32t + 0..15, h = 1: + 16..31)
        auto s02 = __builtin_amdgcn_permlane32_swap(edw[0], edw[2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(edw[1], edw[3], false, false);
        const v4i v = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
        *reinterpret_cast<v4i*>(stg + c * SP + h * 16) = v;
        // store: lane -> (row lane >> 1, chunk lane & 1): 32 rows x 32 bytes, one 16-byte piece per lane
        const int row = lane >> 1, cj = lane & 1;
        const uint32_t J = (uint32_t)nb * NCH + 2 * t + cj;
        const uint32_t b = fastdiv_u31(J, g.cpi_m, g.cpi_s);
        const int pj = (int)(J - b * CPI) << 4;
        const int hw0 = pj < full16 ? pj : g.HWX - 16;
        const int m = mt * 32 + row;
        const v4i o = *reinterpret_cast<const v4i*>(stg + row * SP + cj * 16);
        if (b < (uint32_t)g.NB && m < g.M) {
          int8_t* yp = reinterpret_cast<int8_t*>(g.y) + (size_t)b * g.y_bstride + (size_t)m * (uint32_t)g.HWY + hw0;  // possibly unaligned: fine
          __builtin_memcpy(yp, &o, 16);  // (write-through `sc1` stores measured slower: 13.07 vs 12.43 us on the 512 -> 512 layer)
        }
      }
    } else {
      // 32-bit outputs: register group gq = r >> 2 is 4 consecutive columns: chunk 2t + (gq >> 1), column 8 (gq & 1) + 4h.
      // The duplicate columns of an end-aligned chunk are rewritten with equal values.
      if constexpr ((r & 7) == 0) {
        const uint32_t J = (uint32_t)nb * NCH + 2 * t + (r >> 3);
        const uint32_t b = fastdiv_u31(J, g.cpi_m, g.cpi_s);
        const int pj = (int)(J - b * CPI) << 4;
        eok = b < (uint32_t)g.NB && mrow < g.M;
        eoff = (size_t)b * g.y_bstride + (size_t)mrow * (uint32_t)g.HWY + (pj < full16 ? pj : g.HWX - 16) + 4 * h;
      }
      if constexpr (OUT == OUT_F32) {
        float y = __fmaf_rn((float)acc[t][r], sc, bi);
        if (g.act == ACT_LEAKY) y = y > 0.f ? y : g.alpha * y;  // kernel-uniform
        ef[r & 3] = fminf(fmaxf(y, flo), fcap);
        if constexpr ((r & 3) == 3) {
          const v4f v = {ef[0], ef[1], ef[2], ef[3]};
          if (eok) __builtin_memcpy(reinterpret_cast<float*>(g.y) + eoff + 8 * ((r >> 2) & 1), &v, 16);
        }
      } else if constexpr ((r & 3) == 3) {
        const v4i v = {acc[t][r - 3], acc[t][r - 2], acc[t][r - 1], acc[t][r]};
        if (eok) __builtin_memcpy(reinterpret_cast<int*>(g.y) + eoff + 8 * ((r >> 2) & 1), &v, 16);
      }
    }
  };
  auto runslices = [&](auto self, auto e_c, auto eend_c) __attribute__((always_inline)) -> void {
    constexpr int e = decltype(e_c)::value, eend = decltype(eend_c)::value;
    if constexpr (e < eend) {
      slice(integral_constant<int, e / 16>{}, integral_constant<int, e % 16>{});
      self(self, integral_constant<int, e + 1>{}, integral_constant<int, eend>{});
    }
  };
  {
    PLHIP_WIDE_READ(p2ks(0), p2tile(0), 0);
    if constexpr (Q > 1) PLHIP_WIDE_READ(p2ks(1), p2tile(1), 1);
    auto mm2 = [&](auto self, auto i_c) __attribute__((always_inline)) -> void {
      constexpr int i = decltype(i_c)::value;
      if constexpr (i < Q) {
        constexpr int t = p2tile(i), j = p2ks(i), f = i % 3;
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo[f]), "+v"(hi[f]) : "n"(i + 1 < Q ? 2 : 0) : "memory");
        const v4i a = {lo[f][0], lo[f][1], hi[f][0], hi[f][1]};
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w[j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + 2 < Q) PLHIP_WIDE_READ(p2ks(i + 2), p2tile(i + 2), (i + 2) % 3);
        runslices(runslices, integral_constant<int, cursor(i)>{}, integral_constant<int, cursor(i + 1)>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j == KS - 1 && t < 16) PLHIP_WIDE_STAMP(30 + t);
        self(self, integral_constant<int, i + 1>{});
      }
    };
    mm2(mm2, integral_constant<int, 0>{});
    PLHIP_WIDE_STAMP(7);
    runslices(runslices, integral_constant<int, cursor(Q)>{}, integral_constant<int, 16 * NTT>{});  // what is left: the last tile at least
  }
#undef PLHIP_WIDE_READ
  if (diag) {  // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      lstamp[8] = __builtin_amdgcn_s_memtime();  // stores acknowledged
      lstamp[9] = __builtin_amdgcn_s_memrealtime();
    }
    if (g.stamps && blockIdx.x < 512 && lane < WIDE_STAMP_SLOTS)
      g.stamps[((size_t)blockIdx.x * 8 + wave) * WIDE_STAMP_SLOTS + lane] = lstamp[lane];
  }
}

static inline void cpi_magic(long d, unsigned& m, int& sh) {  // fastdiv_u31's (magic, shift) for divisor d (dw_common.h)
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

template <int NTT, int KS, int OUT>
static inline void launch_wide_t(GemmArgs g, hipStream_t s) {
  constexpr int A0 = 4, R = 2;
  constexpr int C1 = 2 * NTT > 8 ? 2 * NTT - 8 : 0, KSTEP = 4 * (1024 + C1 * 128);
  constexpr int LDS_MAIN = KS * KSTEP + 8 * 32 * 48;
  static_assert(LDS_MAIN + 8 * WIDE_STAMP_SLOTS * 8 <= 160 * 1024, "LDS");
  const int CPI = (g.HWX + 15) >> 4;
  const long chunks = (long)g.NB * CPI;
  g.NT = (int)((chunks + 2 * NTT - 1) / (2 * NTT));
  g.MT = (g.M + 255) / 256;
  cpi_magic(CPI, g.cpi_m, g.cpi_s);
  const unsigned blocks = (unsigned)((long)g.MT * ((g.NT + 7) / 8 * 8));
  const size_t lds = (size_t)LDS_MAIN + 8 * WIDE_STAMP_SLOTS * 8;
  const bool nonneg = g.act == ACT_RELU || g.act == ACT_RELU6;
  if (OUT == OUT_I8 && !nonneg) {
    auto kfn = gemm_i8_wide_kernel<NTT, KS, OUT, A0, R, false>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, g);
  } else {
    auto kfn = gemm_i8_wide_kernel<NTT, KS, OUT, A0, R, true>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, g);
  }
}

template <int NTT, int KS>
static inline void launch_wide_o(const GemmArgs& g, int out, hipStream_t s) {
  if (out == OUT_I32) launch_wide_t<NTT, KS, OUT_I32>(g, s);
  else if (out == OUT_F32) launch_wide_t<NTT, KS, OUT_F32>(g, s);
  else launch_wide_t<NTT, KS, OUT_I8>(g, s);
}

}  // namespace plhip
