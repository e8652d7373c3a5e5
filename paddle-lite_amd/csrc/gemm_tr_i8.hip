// gemm_tr_i8.hip — the MFMA-heavy int8 GEMM of the 1x1 / implicit-GEMM convolutions, second generation:
// LDS-DMA ring + gfx950's TRANSPOSING LDS read (ds_read_b64_tr_b8) + swapped MFMA roles.
//
// Replaces the same reference code as gemm_i8.hip (gemm_prepack_int8, lite/backends/arm/math/gemm_prepacked_int8.cc:
// 2582-2744 hot loop, :643-796 epilogue; packb_int8 :3285; the batch loop of conv1x1s1_gemm_int8, conv_impl.cc:260-331).
//
// Why a second kernel (round-1 timeline of gemm_i8_dma_kernel, profiles/r01_final_gemm_timeline_pw8.txt): per 32-deep
// K-step a wave spent ~1000 cycles for 256 cycles of MFMA: 16 ds_read_b32 + 32 v_perm to turn N-contiguous NCHW rows into
// the K-contiguous MFMA operand (repeated by all 4 waves), and an epilogue of 32 dword stores per lane.  Here
//   * the activation tile stays RAW in LDS ([k][n] rows exactly as the DMA delivers them) and ds_read_b64_tr_b8 hands
//     every lane 8 K-consecutive bytes of ITS column: 2 reads per operand, no VALU at all in the K loop.  Semantics
//     (tools/probe_tr8.hip, run on the device): per 16-lane group, lane 2q+p supplies the address of row q, 8-byte
//     sub-chunk p; lane i receives byte (i&7) of sub-chunk (i>>3) of rows 0..7 in its bytes 0..7;
//   * LDS image of one K-step of activations = [128-column group][kg = k/8][j = 16-byte column chunk][kr = k%8][16 B]:
//     one DMA instruction (64 lanes x 16 B, LDS destination lane-linear) fills one (group, kg): lane -> (j = lane>>3,
//     kr = lane&7), the per-lane SOURCE address does the permutation.  A half-wave's transposed read then covers 256
//     contiguous bytes: conflict free;
//   * the MFMA roles are SWAPPED: activations are the A operand (rows = n), weights the B operand (columns = m), so the
//     32x32 result has m on the lanes and n in the registers: a lane owns ONE output channel (its scale / bias are two
//     scalars, no LDS staging) and, per register group, 4 consecutive n.  Two v_permlane32_swap per 32-column tile give
//     every lane 16 consecutive int8 results of its channel row: ONE 16-byte store per 32x32 tile and lane (was 4 dword
//     stores), 32 bytes contiguous per row and instruction;
//   * block tile up to 256 (n) x 256 (m) with 8 waves (2 per SIMD): per K-step 8 KiB of activations + 8 KiB of weights
//     (fragment order, as packed by pack_weights_kernel: the natural k order of the transposed read needs no new packing)
//     feed 64 MFMAs: 32 B/clk/CU from L2, 96 B/clk of LDS reads.
// Column space, end-aligned last 16-byte piece, `skip`, implicit GEMM addressing: as in gemm_i8_dma_kernel (gemm_i8.hip).
#include <stdlib.h>

#include <type_traits>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "gemm_tr_common.h"

namespace plhip {

// ---- diagnostic timeline (PLHIP_GEMM_DEBUG & 32; never set in production): per-wave s_memtime stamps kept in LDS during
// the run and flushed to this buffer at the end (plhip_debug_read_stamps reads it; tools/gemm_timeline.py --tr)
constexpr int TR_STAMP_SLOTS = 32;
__device__ unsigned long long g_tr_stamps[1024 * 8 * TR_STAMP_SLOTS];
#define PLHIP_TR_STAMP(i)                                                 \
  do {                                                                    \
    if (diag && lane == 0) lstamp[i] = __builtin_amdgcn_s_memtime();      \
  } while (0)

// WN x WM waves; every wave owns 128 (n) x 64 (m) outputs = 4 x 2 MFMA tiles.  D K-steps in flight, D + 1 ring slots.
template <int WN, int WM, int OUT, int D, bool IM>
__global__ __launch_bounds__(64 * WN * WM, 2) void gemm_i8_tr_kernel(GemmArgs g) {
  constexpr int NW = WN * WM;
  constexpr int BN = WN * 128, BM = WM * 64;
  constexpr int ACT_BYTES = BN * 32, W_BYTES = BM * 32, SLOT = ACT_BYTES + W_BYTES;
  constexpr int APC = WN * 4, WPC = WM * 2, PT = APC + WPC;  // 1-KiB DMA pieces per K-step: activations, weights
  constexpr int PW = (PT + NW - 1) / NW;                      // ... per wave (surplus ones repeat the wave's previous piece)
  constexpr int NS = D + 1;
  PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.x); PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias);
  PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.K); PLHIP_PRELOAD(g.KS); PLHIP_PRELOAD(g.HWX); PLHIP_PRELOAD(g.HWY); PLHIP_PRELOAD(g.XP);
  PLHIP_PRELOAD(g.NB); PLHIP_PRELOAD(g.x_bstride); PLHIP_PRELOAD(g.y_bstride); PLHIP_PRELOAD(g.MT); PLHIP_PRELOAD(g.NT);
  PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha); PLHIP_PRELOAD(g.dbg);
  PLHIP_PRELOAD(g.im_kw); PLHIP_PRELOAD(g.im_khkw); PLHIP_PRELOAD(g.im_c); PLHIP_PRELOAD(g.im_ph); PLHIP_PRELOAD(g.im_pw); PLHIP_PRELOAD(g.im_oh); PLHIP_PRELOAD(g.im_s);
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];  // NS * SLOT, ONE LDS object
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave / WM, wm = wave - wn * WM;  // wave-uniform
  int mb, nb;
  tr_xcd_tile_map(blockIdx.x, g.MT, g.NT, mb, nb);  // g.MT = blocks along M, g.NT = blocks along N (set by the launcher)
  if (nb >= g.NT) return;                            // block-uniform (grid padded to 8 N blocks)
  // experiment (PLHIP_TR_DELAY, units of 64 clocks): the second block of a CU starts late, so that its K loop runs
  // beside the first block's epilogue (VALU + stores) instead of beside its K loop
  if ((g.dbg >> 8) > 0 && blockIdx.x >= 256) {
    for (int i = 0; i < (g.dbg >> 8); i += 100) __builtin_amdgcn_s_sleep(100);
  }
  const int c = lane & 31, h = lane >> 5;
  const int KS = g.KS;
  const int MT32 = (g.M + 31) >> 5;
  const bool diag = (g.dbg & 32) != 0;
  unsigned long long* lstamp = reinterpret_cast<unsigned long long*>(ring + NS * SLOT) + wave * TR_STAMP_SLOTS;
  if (diag && lane == 0) {
    lstamp[0] = __builtin_amdgcn_s_memrealtime();
    lstamp[1] = __builtin_amdgcn_s_memtime();
    lstamp[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
  constexpr bool implicit = IM;  // implicit GEMM on the zero-padded input copy (g.im_kw > 0)

  // ---- column space (gemm_i8_dma_kernel): every image's HWX columns padded to HWP = roundup(HWX, 16); the last
  // 16-byte chunk of an image is END-aligned (source columns HWX-16 .. HWX-1), its leading 16 - HWX%16 columns are
  // duplicates that are never stored
  // Rows shorter than 16 columns (implicit GEMM on 14x14 / 7x7 planes: an "image" is one output row): ONE start-aligned
  // chunk per row, its trailing 16 - HWX columns are garbage (bytes of the next padded row) and are never stored.
  const bool short_rows = g.HWX < 16;  // block-uniform
  const int HWP = (g.HWX + 15) & ~15, full16 = short_rows ? 16 : (g.HWX & ~15), rem16 = short_rows ? 0 : (g.HWX & 15);
  const int CPI = HWP >> 4;  // 16-column chunks per image

  // ---- this lane's scale / bias: ordinary loads FIRST and alone (next to LDS-DMA the compiler can only wait for an
  // ordinary load with vmcnt(0)); they are consumed after the K loop, when everything has drained anyway
  float sc[2] = {1.f, 1.f}, bi[2] = {0.f, 0.f};
  int mrow[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    mrow[u] = mb * BM + wm * 64 + 32 * u + c;
    if (OUT != OUT_I32 && mrow[u] < g.M) {
      sc[u] = g.scale[mrow[u]];
      if (g.bias) bi[u] = g.bias[mrow[u]];
    }
  }

  // ---- DMA pieces of this wave: piece p = q * NW + wave (q = 0 .. PW-1); p < APC: activation piece (group p / 4,
  // kg = p % 4), else weight fragment p - APC.  Branch-free issue: source = min(base + ks * inc, lim) for both kinds
  // (activations: inc = 32 rows, lim = row K-1: rows past K meet zero-padded weights, only the address must stay legal;
  // weights: inc = 1 KiB, no limit); implicit GEMM: the activation offset comes from the (channel, tap) walk instead.
  const uint8_t* src[PW];   // K-step-0 source of this lane's 16 bytes
  const uint8_t* lim[PW];
  uint32_t inc[PW];
  int ldsoff[PW];           // wave-uniform destination offset inside a slot
  bool isact[PW];
  int kc[PW], krs[PW];      // implicit GEMM: (channel, tap) of this lane's k row, advanced by 32 rows per issued K-step
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    int p = q * NW + wave;
    if (p >= PT) p -= NW;  // surplus: the wave's previous piece once more (same bytes to the same place)
    isact[q] = p < APC;
    kc[q] = krs[q] = 0;
    // activation piece
    const int grp = p >> 2, kg = p & 3;
    const int prow = 8 * kg + (lane & 7);
    const int J = nb * (BN / 16) + (grp < WN ? grp : 0) * 8 + (lane >> 3);  // my 16-column chunk of the padded column space
    int pb = J / CPI;
    int pj = (J - pb * CPI) << 4;
    if (pb >= g.NB) { pb = 0; pj = 0; }
    const int pcol = pj < full16 ? pj : g.HWX - 16;  // short rows: full16 = 16, pj = 0
    const uint8_t* asrc;
    if (implicit) {
      const int bi_ = pb / g.im_oh, oh = pb - bi_ * g.im_oh;
      const int nph = g.im_s * g.im_s;  // phase planes per channel (stride 2: 4)
      asrc = reinterpret_cast<const uint8_t*>(g.x) + ((size_t)bi_ * g.im_c * nph * g.im_ph + oh) * g.im_pw + pcol;
      kc[q] = prow / g.im_khkw;
      krs[q] = prow - kc[q] * g.im_khkw;
    } else {
      asrc = reinterpret_cast<const uint8_t*>(g.x) + (size_t)pb * g.x_bstride + pcol;
    }
    const uint8_t* alim = asrc + (size_t)(g.K - 1) * (uint32_t)g.XP;
    if (!implicit) asrc += (size_t)(prow < g.K ? prow : g.K - 1) * (uint32_t)g.XP;
    // weight fragment (32-row m tile) f of this block; tiles past M: any packed tile (their outputs are never stored)
    const int f = p - APC;
    int mt32 = mb * (BM / 32) + (f > 0 ? f : 0);
    mt32 = mt32 < MT32 ? mt32 : MT32 - 1;
    const uint8_t* wsrc = reinterpret_cast<const uint8_t*>(g.wp) + (size_t)mt32 * KS * 1024 + lane * 16;
    src[q] = isact[q] ? asrc : wsrc;
    lim[q] = isact[q] ? alim : reinterpret_cast<const uint8_t*>(~(uintptr_t)0);
    inc[q] = isact[q] ? (implicit ? 0u : 32u * (uint32_t)g.XP) : 1024u;
    ldsoff[q] = isact[q] ? p * 1024 : ACT_BYTES + f * 1024;
  }
  const int kc_step = implicit ? 32 / g.im_khkw : 0, krs_step = implicit ? 32 - kc_step * g.im_khkw : 0;

  auto issue = [&](int ks, int slot) {
    uint8_t* sb = ring + slot * SLOT;
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      const uint8_t* p = src[q] + (size_t)((uint32_t)ks * inc[q]);  // < 2^32: K * XP is one image (checked on the host)
      p = p < lim[q] ? p : lim[q];
      if (implicit) {
        // rows past K meet zero-padded weights: any in-bounds address will do (the last real tap)
        const int cc = kc[q] < g.im_c ? kc[q] : g.im_c - 1, rs = kc[q] < g.im_c ? krs[q] : g.im_khkw - 1;
        const int r = (rs * ((65536 + g.im_kw - 1) / g.im_kw)) >> 16;  // rs / kw, exact for rs < 128, kw <= 11
        const int sx = rs - r * g.im_kw;
        // stride 2: tap (r, sx) lives in phase plane (r & 1, sx & 1) at row / column offset (r >> 1, sx >> 1)
        const size_t off = g.im_s == 1 ? ((size_t)cc * g.im_ph + r) * g.im_pw + sx
                                       : ((size_t)(cc * 4 + (r & 1) * 2 + (sx & 1)) * g.im_ph + (r >> 1)) * g.im_pw + (sx >> 1);
        kc[q] += kc_step;
        krs[q] += krs_step;
        if (krs[q] >= g.im_khkw) {
          krs[q] -= g.im_khkw;
          ++kc[q];
        }
        p = isact[q] ? src[q] + off : p;
      }
      __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(sb + ldsoff[q]), 16, 0, 0);
    }
  };

  // ---- operand reads of one K-step: 8 transposed reads (4 n tiles x 2) + 2 weight fragments.
  // transposed-read address of this lane inside an activation group: k half h -> kg {2h, 2h+1}; 16-lane group parity ->
  // chunk j (even / odd); lane 2q+p of the group -> row q, sub-chunk p.
  // The reads are inline asm ON PURPOSE: hipcc cannot tell which ring slot an in-flight LDS-DMA writes and guards ordinary
  // LDS reads of the ring with s_waitcnt vmcnt(0) (seen in the ISA of the builtin form of this loop), which drains the
  // whole DMA pipeline every K-step.  The counted vmcnt + barrier above each read is the real ordering; the values are
  // waited for (lgkmcnt) at the END of the step that issues the reads, one whole MFMA block later, in a statement that
  // names every destination register, so no use or copy can be scheduled above it.
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
  const uint32_t tr_lane = ring_addr + (h * 2) * 1024 + ((lane >> 4) & 1) * 128 + ((lane & 15) >> 1) * 16 + (lane & 1) * 8 + wn * 4096;
  const uint32_t w_lane = ring_addr + ACT_BYTES + (wm * 2) * 1024 + lane * 16;
  struct Frags {
    v2i lo[4], hi[4];
    v4i w[2];
  };
  auto read_slot = [&](int slot, Frags& f) {
    const uint32_t ta = tr_lane + slot * SLOT, wa = w_lane + slot * SLOT;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(f.lo[t]) : "v"(ta), "n"(t * 256));
      asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(f.hi[t]) : "v"(ta), "n"(t * 256 + 1024));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.w[u]) : "v"(wa), "n"(u * 1024));
  };
  auto wait_frags = [&](Frags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.lo[0]), "+v"(f.lo[1]), "+v"(f.lo[2]), "+v"(f.lo[3]), "+v"(f.hi[0]), "+v"(f.hi[1]), "+v"(f.hi[2]),
                   "+v"(f.hi[3]), "+v"(f.w[0]), "+v"(f.w[1]));
  };

  // ---- pipeline (the launcher guarantees KS >= D) ----
#pragma unroll
  for (int p = 0; p < D; ++p) issue(p, p);
  PLHIP_TR_STAMP(3);
  v16i acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0;  // behind the prologue DMA, while the first bytes travel
  Frags fs[2];  // ping-pong register sets: K-step ks multiplies from fs[par], ks+1 is read into fs[par ^ 1] meanwhile
  int rslot = 1, islot = D % NS;
  // K-step ks: NEXT = ks+1 < KS, ISSUE = ks+D < KS, YOUNGER = K-steps issued behind ks+1 at its wait
  auto step = [&](int ks, Frags& cur, Frags& nxt, auto younger_c, auto issue_c, auto next_c) {
    constexpr int YOUNGER = decltype(younger_c)::value;
    constexpr bool ISSUE = decltype(issue_c)::value;
    constexpr bool NEXT = decltype(next_c)::value;
    if (ks < TR_STAMP_SLOTS - 9) PLHIP_TR_STAMP(5 + ks);
    const bool sub = diag && (g.dbg & 64) && ks == 6;  // sub-stamps of one steady-state K-step (they perturb it)
    if (NEXT) {
      tr_wait_vmcnt<YOUNGER * PW>();   // my pieces of K-step ks+1 have landed
      if (sub && lane == 0) lstamp[22] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();     // ... everyone's have; nobody reads K-step ks-1's slot any more
      if (sub && lane == 0) lstamp[23] = __builtin_amdgcn_s_memtime();
      read_slot(rslot, nxt);
      rslot = rslot + 1 == NS ? 0 : rslot + 1;
      if (sub && lane == 0) lstamp[24] = __builtin_amdgcn_s_memtime();
    }
    if (ISSUE) {
      issue(ks + D, islot);
      islot = islot + 1 == NS ? 0 : islot + 1;
      if (sub && lane == 0) lstamp[25] = __builtin_amdgcn_s_memtime();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const v4i a = {cur.lo[t][0], cur.lo[t][1], cur.hi[t][0], cur.hi[t][1]};
        acc[t][u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, cur.w[u], acc[t][u], 0, 0, 0);
      }
    if (sub && lane == 0) lstamp[21] = __builtin_amdgcn_s_memtime();
    if (NEXT) wait_frags(nxt);
  };
  using std::integral_constant;
  typedef integral_constant<bool, true> T_;
  typedef integral_constant<bool, false> F_;
  typedef integral_constant<int, D - 2> Y_;
  static_assert(D == 3 || D == 4, "the tail below is written out for 3 or 4 K-steps in flight");
  {
    // steps 0 .. S-1 (S = KS - D) issue and read ahead; then ks = KS-D+T, T = 0 .. D-2: (D-2-T, -, next); KS-1: (0, -, -).
    // The sets alternate with ks; an odd S starts in set 1 so that the tail always starts in set 0.
    const int S = KS - D;
    tr_wait_vmcnt<(D - 1) * PW>();
    __builtin_amdgcn_s_barrier();
    PLHIP_TR_STAMP(4);
    int ks = 0;
    if (S & 1) {
      read_slot(0, fs[1]);
      wait_frags(fs[1]);
      step(0, fs[1], fs[0], Y_{}, T_{}, T_{});
      ks = 1;
    } else {
      read_slot(0, fs[0]);
      wait_frags(fs[0]);
    }
    for (; ks < S; ks += 2) {
      step(ks, fs[0], fs[1], Y_{}, T_{}, T_{});
      step(ks + 1, fs[1], fs[0], Y_{}, T_{}, T_{});
    }
    if (D == 4) {
      step(ks, fs[0], fs[1], integral_constant<int, 2>{}, F_{}, T_{});
      step(ks + 1, fs[1], fs[0], integral_constant<int, 1>{}, F_{}, T_{});
      step(ks + 2, fs[0], fs[1], integral_constant<int, 0>{}, F_{}, T_{});
      step(ks + 3, fs[1], fs[0], integral_constant<int, 0>{}, F_{}, F_{});
    } else {
      step(ks, fs[0], fs[1], integral_constant<int, 1>{}, F_{}, T_{});
      step(ks + 1, fs[1], fs[0], integral_constant<int, 0>{}, F_{}, T_{});
      step(ks + 2, fs[0], fs[1], integral_constant<int, 0>{}, F_{}, F_{});
    }
  }

  PLHIP_TR_STAMP(TR_STAMP_SLOTS - 4);
  // ---- epilogue: lane (c, h) owns channel rows mrow[0], mrow[1]; per n tile t, register r <-> n = 32t + 8(r>>2) + 4h + (r&3)
  const int Jb = nb * (BN / 16) + wn * 8;  // first chunk of this wave's 128 columns
  if (OUT == OUT_I8) {
    // int8: the 128 x 64 tile of this wave goes through a wave-private LDS image [64 m][144-byte pitch] so that every
    // global store instruction writes 8 rows x 128 CONTIGUOUS bytes (lane -> row lane>>3, 16-byte chunk lane&7).
    // Storing straight from the MFMA layout (one channel row per lane, 32 bytes contiguous per row and instruction)
    // ran at 7 B/clk/CU: 9.0k of the launch's 25k cycles on the 512->512 14x14 layer.
    __builtin_amdgcn_s_barrier();  // every wave has finished reading the ring: it becomes staging space
    PLHIP_TR_STAMP(TR_STAMP_SLOTS - 6);
    uint8_t* stg = ring + wave * (64 * 144);
    if (g.y2) {  // kernel-uniform: calib-only tail of an fp32-output conv (launch_gemm_tr): the int8 tensor is g.y2
      tr_stage_i8_calib(acc, sc, bi, g.act, g.alpha, g.inv_scale2, stg, c, h);
    } else {
      switch (g.act) {  // wave-uniform: straight-line requantisation per activation
        case ACT_RELU: tr_stage_i8<ACT_RELU>(acc, sc, bi, g.alpha, stg, c, h); break;
        case ACT_RELU6: tr_stage_i8<ACT_RELU6>(acc, sc, bi, g.alpha, stg, c, h); break;
        case ACT_LEAKY: tr_stage_i8<ACT_LEAKY>(acc, sc, bi, g.alpha, stg, c, h); break;
        default: tr_stage_i8<ACT_NONE>(acc, sc, bi, g.alpha, stg, c, h); break;
      }
    }
    PLHIP_TR_STAMP(TR_STAMP_SLOTS - 5);  // requantised + staged
    // this lane's 16-column chunk (the same for all 8 store rounds) and its first row
    const int J = Jb + (lane & 7);
    int b = J / CPI;
    const int pj = (J - b * CPI) << 4;
    const bool cvalid = b < g.NB;
    const int hw0 = pj < full16 ? pj : g.HWX - 16;
    const int skip = pj < full16 ? 0 : 16 - rem16;
    const int room = short_rows ? g.HWX : (implicit ? 16 : g.HWY - hw0);  // output columns left in the row from the chunk's first column
    int hw = hw0;
    if (implicit) {
      const int bi_ = b / g.im_oh;
      hw += (b - bi_ * g.im_oh) * g.HWX;
      b = bi_;
    }
    const int m0 = mb * BM + wm * 64 + (lane >> 3);
    int8_t* yp = (g.y2 ? g.y2 : reinterpret_cast<int8_t*>(g.y)) + (size_t)b * g.y_bstride + (size_t)m0 * (uint32_t)g.HWY + hw;
    const uint8_t* rp = stg + (lane >> 3) * 144 + (lane & 7) * 16;
    const bool fast = skip == 0 && room >= 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const v4i v = *reinterpret_cast<const v4i*>(rp + i * 8 * 144);
      if (cvalid && m0 + 8 * i < g.M) {
        int8_t* q = yp + (size_t)(8 * i) * (uint32_t)g.HWY;
        if (fast) __builtin_memcpy(q, &v, 16);  // possibly unaligned: fine for global memory
        else store_chunk_i8(q, (uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3], skip, room);
      }
    }
  } else {
    // 32-bit outputs: a lane's 4 consecutive n of register group gq are one 16-byte store; chunk 2t + (gq >> 1),
    // column 8 (gq & 1) + 4h inside it
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        const int J = Jb + 2 * t + jc;
        int b = J / CPI;
        const int pj = (J - b * CPI) << 4;
        const bool cvalid = b < g.NB;
        const int hw0 = pj < full16 ? pj : g.HWX - 16;
        const int skip = pj < full16 ? 0 : 16 - rem16;
        const int room = short_rows ? g.HWX : (implicit ? 16 : g.HWY - hw0);
        int hwb = hw0;
        if (implicit) {
          const int bi_ = b / g.im_oh;
          hwb += (b - bi_ * g.im_oh) * g.HWX;
          b = bi_;
        }
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
          const int gq = 2 * jc + gl;
          const int o = 8 * gl + 4 * h;  // first column of the group inside the chunk
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (!(cvalid && mrow[u] < g.M) || o + 3 < skip || o >= room) continue;
            const size_t yoff = (size_t)b * g.y_bstride + (size_t)mrow[u] * (uint32_t)g.HWY + hwb + o;
            if (OUT == OUT_I32) {
              int* yp = reinterpret_cast<int*>(g.y) + yoff;
              if (o >= skip && o + 3 < room) {
                const v4i v = {acc[t][u][4 * gq], acc[t][u][4 * gq + 1], acc[t][u][4 * gq + 2], acc[t][u][4 * gq + 3]};
                __builtin_memcpy(yp, &v, 16);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (o + e >= skip && o + e < room) yp[e] = acc[t][u][4 * gq + e];
              }
            } else {
              float f[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) f[e] = epilogue_f32(acc[t][u][4 * gq + e], sc[u], bi[u], g.act, g.alpha);
              const bool whole = o >= skip && o + 3 < room;
              if (g.res) {  // fused residual add (+ relu): kernel-uniform
                const float* rp = g.res + yoff;
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                if (whole) {
                  v4f rv;
                  __builtin_memcpy(&rv, rp, 16);
                  r[0] = rv[0]; r[1] = rv[1]; r[2] = rv[2]; r[3] = rv[3];
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (o + e >= skip && o + e < room) r[e] = rp[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  f[e] = f[e] + r[e];
                  if (g.res_relu) f[e] = f[e] > 0.f ? f[e] : 0.f;
                }
              }
              if (g.y) {
                float* yp = reinterpret_cast<float*>(g.y) + yoff;
                if (whole) {
                  const v4f v = {f[0], f[1], f[2], f[3]};
                  __builtin_memcpy(yp, &v, 16);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (o + e >= skip && o + e < room) yp[e] = f[e];
                }
              }
              if (g.y2) {  // fused calib fp32 -> int8 of the value just produced (type_trans.cc:45,183-184)
                const uint32_t packed = pack4_i8(round_sat_i8(g.inv_scale2 * f[0]), round_sat_i8(g.inv_scale2 * f[1]),
                                                 round_sat_i8(g.inv_scale2 * f[2]), round_sat_i8(g.inv_scale2 * f[3]));
                int8_t* qp = g.y2 + yoff;
                if (whole) {
                  __builtin_memcpy(qp, &packed, 4);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (o + e >= skip && o + e < room) qp[e] = (int8_t)(packed >> (8 * e));
                }
              }
            }
          }
        }
      }
    }
  }
  if (diag) {  // wave-uniform
    PLHIP_TR_STAMP(TR_STAMP_SLOTS - 3);  // epilogue instructions issued
    tr_wait_vmcnt<0>();
    if (lane == 0) {
      lstamp[TR_STAMP_SLOTS - 2] = __builtin_amdgcn_s_memtime();  // stores acknowledged
      lstamp[TR_STAMP_SLOTS - 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (blockIdx.x < 1024 && lane < TR_STAMP_SLOTS)
      g_tr_stamps[((size_t)blockIdx.x * 8 + wave) * TR_STAMP_SLOTS + lane] = lstamp[lane];
  }
}

int debug_read_tr_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(unsigned long long) * 1024 * 8 * TR_STAMP_SLOTS) bytes = sizeof(unsigned long long) * 1024 * 8 * TR_STAMP_SLOTS;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tr_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

int gemm_tr_enabled() {  // PLHIP_GEMM_TR=0: first-generation kernels only (A/B runs)
  const int v = knob("GEMM_TR", 1);
  return v;
}

template <int WN, int WM, int OUT, bool IM>
static void launch_tr_cfg2(GemmArgs g, hipStream_t s) {
  // 4-wave blocks (M <= 64): 3 K-steps in flight = 4 ring slots = 72 KiB, so that TWO blocks share a CU (with one, its
  // four waves read LDS together and multiply together: 11 B/clk of ingest; two blocks de-phase each other)
  constexpr int D = WN * WM == 4 ? 3 : 4;
  constexpr int BN = WN * 128, BM = WM * 64;
  const int HWP = (g.HWX + 15) & ~15;
  g.NT = (int)(((long)g.NB * HWP + BN - 1) / BN);  // blocks along N
  g.MT = (g.M + BM - 1) / BM;                      // blocks along M
  const unsigned blocks = (unsigned)((long)g.MT * ((g.NT + 7) / 8 * 8));
  const size_t lds = (size_t)(D + 1) * (BN * 32 + BM * 32) + 8 * TR_STAMP_SLOTS * 8;
  auto kfn = gemm_i8_tr_kernel<WN, WM, OUT, D, IM>;
  // per DEVICE and called from several predictor threads: set on every launch (a process-wide flag was wrong on a second GPU)
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(blocks), dim3(64 * WN * WM), lds, s, g);
}

template <int WN, int WM, int OUT>
static void launch_tr_cfg(const GemmArgs& g, hipStream_t s) {
  if (g.im_kw > 0) launch_tr_cfg2<WN, WM, OUT, true>(g, s);
  else launch_tr_cfg2<WN, WM, OUT, false>(g, s);
}

// Returns true when the launch was taken.  g.HWX must already be the TRUE row length (dense slabs) / the padded row
// length of the im2col buffer / OW (implicit GEMM); KS >= 4, HWX >= 16.
bool launch_gemm_tr(const GemmArgs& g_in, int out, hipStream_t s) {
  GemmArgs g = g_in;
  {
    const int delay_env = knob("TR_DELAY", 0);
    g.dbg = (g.dbg & 0xff) | (delay_env << 8);
  }
  // rows shorter than 16 bytes: only on the padded copy of the implicit route (a 16-byte piece may run past the row)
  if (!gemm_tr_enabled() || g.KS < 4 || (g.HWX < 16 && g.im_kw == 0)) return false;
  if ((long)g.NB * ((g.HWX + 15) & ~15) >= ((long)1 << 31) - 1024) return false;
  // fp32-output conv whose fp32 value nobody reads and whose only tail is the calib: the staged int8 epilogue (16-byte
  // row stores) instead of the row-per-lane 32-bit one (ResNet50's stem behind the int8 max pool: 0.54 -> see DESIGN 4)
  if (out == OUT_F32 && !g.y && g.y2 && !g.res) out = OUT_I8;
#define PLHIP_TR_OUT(WN_, WM_)                                          \
  do {                                                                  \
    if (out == OUT_I32) launch_tr_cfg<WN_, WM_, OUT_I32>(g, s);         \
    else if (out == OUT_F32) launch_tr_cfg<WN_, WM_, OUT_F32>(g, s);    \
    else launch_tr_cfg<WN_, WM_, OUT_I8>(g, s);                         \
  } while (0)
      // default 3: 4-wave blocks (128 x 256 / 256 x 128 tiles), two per CU, for every M (ResNet50's 3x3 layers: 5-8 % faster
    // than one 8-wave block per CU, whose waves read LDS together and multiply together); 0 = the 8-wave tiles
  const int cfg_env = knob("TR_CFG", 3);
  if (g.M > 128 && (cfg_env & 1)) PLHIP_TR_OUT(1, 4);
  else if (g.M > 128) PLHIP_TR_OUT(2, 4);
  else if (g.M > 64 && (cfg_env & 2)) PLHIP_TR_OUT(2, 2);
  else if (g.M > 64) PLHIP_TR_OUT(4, 2);
  else PLHIP_TR_OUT(4, 1);
#undef PLHIP_TR_OUT
  return true;
}

}  // namespace plhip
