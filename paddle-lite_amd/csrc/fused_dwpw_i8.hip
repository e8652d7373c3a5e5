// fused_dwpw_i8.hip — depthwise 3x3 [int8_out] fused with the pointwise 1x1 convolution that consumes it, ONE launch.
//
// SURVEY.md §8(f) rank 1.  Replaces the instruction pair
//   DepthwiseConv<kInt8,kInt8>::Run (lite/kernels/arm/conv_depthwise.cc:407-446 -> lite/backends/arm/math/conv3x3s1_depthwise_int8.cc:33-447)
//   GemmLikeConv<kInt8,*>::Run      (lite/kernels/arm/conv_gemmlike.cc:399-462 -> lite/backends/arm/math/gemm_prepacked_int8.cc:2582-2744)
// of the MobileNet programs.  Run as two kernels the int8 tensor between them makes a full HBM round trip; here it never
// leaves the CU.  Results are bit-identical to plhip_depthwise_conv_int8 (int8 out) followed by plhip_conv2d_int8.
//
// Round 4: third form.  The two earlier kernels (git history) were K-step synchronous — one barrier per 32
// channels, every wave walking stage -> produce -> consume in lock step — and lost to the two kernels (1.4x).  This one is
// built on the wide-tile GEMM's structure (gemm_wide_kernel.h): the whole K x 128-pixel activation tile is LDS-resident in
// the image ds_read_b64_tr_b8 transposes from, and the DEPTHWISE STAGE IS ITS PRODUCER instead of the LDS-DMA:
//   * tile = (image, half): 7 output rows of a 14-wide plane = 7 chunks of 16 pixels (14 real) = 4 MFMA n tiles, all M
//     output channels: one 8-wave block per CU, 256 tiles at batch 128 = one wave of blocks;
//   * produce task = 16 channels x 4 column quads x the 7-row strip on one wave: the rows-in-registers dot4 body of
//     depthwise3x3_direct_kernel (9 unaligned 8-byte row windows per lane straight from global memory, v_alignbyte +
//     v_dot4_i32_i8, the reference's requantisation), each requantised dword written straight into the activation image;
//   * a ROUND = 8 tasks = 128 channels = 4 K-steps.  While round r is produced (VALU), the 4 K-steps of round r - 1 are
//     multiplied (MFMA, weights global -> registers one round ahead, a wave owns 32 MTW output channels for all 4 n tiles):
//     the two pipes of a SIMD work side by side, one barrier per ROUND (4 for K = 512), the next round's input rows and
//     weights are fetched under the current round's arithmetic;
//   * epilogue: requantise, two v_permlane32_swap give a lane one 14-pixel output row of one channel, the wave's 64 x 98
//     bytes are assembled in LDS and leave as 16-byte pieces of whole channel rows.
// No inline-asm memory operation: every wait count is the compiler's.
#include <stdlib.h>

#include <type_traits>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"
#include "gemm_tr_common.h"

namespace plhip {

constexpr int FW_TR = 7;       // output rows per tile
constexpr int FW_NT = 4;       // 32-pixel n tiles per tile (2 rows of pitch 16 each; the second half of the last one is empty)
constexpr int FW_SP = 112;     // staging pitch of a channel row (98 bytes used)
constexpr int FW_KSTEP = 4096; // LDS bytes of one K-step of the activation image: [kg 4][k%8 8][chunk slot 8][16 B]

// ---- diagnostic timeline (plhip_debug_set("fused_stamps", 1); never set in production): s_memtime per wave at the phase
// boundaries, read back by plhip_debug_read_fw_stamps (tools/fused_timeline.py).  Slots: 0 realtime start, 1 entry, 2 operands
// of round 0 requested, 3 round 0 produced, 4 + 2 (r - 1) round r done, 5 + 2 (r - 1) behind its barrier (r < 5), 12 last
// K-steps multiplied, 13 requantised + staged, 14 stores issued, 15 stores acknowledged
constexpr int FW_STAMP_SLOTS = 16;
__device__ unsigned long long g_fw_stamps[1024 * 8 * FW_STAMP_SLOTS];
static int g_fw_debug = 0;
void debug_set_fused(int v) { g_fw_debug = v; }  // bit 5: stamps; bits 0-1: timing experiments
int debug_read_fw_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(g_fw_stamps)) bytes = sizeof(g_fw_stamps);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fw_stamps), bytes) == hipSuccess ? 0 : -1;
}
#define PLHIP_FW_STAMP(i)                                                                                   \
  do {                                                                                                      \
    if (diag && lane == 0) g_fw_stamps[((size_t)vb * 8 + wave) * FW_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// 8 bytes of an input row: scalar base + the lane's 32-bit offset (>= 0) + a compile-time row distance (the immediate of
// the load: no address arithmetic per row)
template <int DELTA>
__device__ __forceinline__ void fw_load_row(const int8_t* __restrict__ xs, uint32_t off, uint32_t (&d)[2]) {
  __builtin_memcpy(d, xs + off + DELTA, 8);
}

// MTW: 32-row m tiles per wave (M = 256 MTW).  OUT: output kind.  DWNN / PWNN: the depthwise / pointwise activation is
// relu or relu6 (the packed non-negative requantisation); else none / leaky (leaky with slope 1 for none: exact).
// EXP (timing experiments only, results are wrong): 1 = no MFMAs in the produce + consume rounds, 2 = no depthwise arithmetic there,
// 3 = no input-row fetches there (the arithmetic runs on stale rows), 4 = no weight fetches there
template <int MTW, int OUT, bool DWNN, bool PWNN, int EXP = 0, int ORDER_T = 1, int NEWQ = 1>
__global__ __launch_bounds__(512, 2) void fused_dwpw14_kernel(FusedArgs a) {
  const GemmArgs& g = a.pw;
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.dw_w); PLHIP_PRELOAD(a.dw_scale); PLHIP_PRELOAD(a.dw_bias); PLHIP_PRELOAD(a.dw_act);
  PLHIP_PRELOAD(a.dw_alpha); PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.C); PLHIP_PRELOAD(a.tiles); PLHIP_PRELOAD(a.ones); PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.y);
  PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias); PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.KS); PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha);
  extern __shared__ __attribute__((aligned(16))) uint8_t fw_lds[];  // [activation image KS x 4096][8 waves x 32 MTW rows x FW_SP]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-contiguous tiles: both halves of an image (they share two halo rows) and neighbouring images on one L2
  const unsigned nb = (unsigned)a.tiles, per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;  // block-uniform
  const int b = (int)(vb >> 1), hf = (int)(vb & 1);
  const bool diag = (g.dbg & 32) != 0 && vb < 1024;
#ifdef PLHIP_FW_AGPR
  // accumulators in AGPRs (this file is built WITHOUT -amdgpu-mfma-vgpr-form): naming one AGPR makes the compiler keep the
  // accumulator file for the MFMA results, whose reads / writes then do not compete with the VALU for the VGPR ports
  {
    int agpr_hint;
    asm volatile("; keep AGPRs %0" : "=a"(agpr_hint));
  }
#endif
  if (diag && lane == 0) g_fw_stamps[((size_t)vb * 8 + wave) * FW_STAMP_SLOTS] = __builtin_amdgcn_s_memrealtime();
  PLHIP_FW_STAMP(1);
  const int KS = g.KS, C = a.C, R = KS >> 2;  // rounds of 128 channels (K % 128 == 0: fused_dwpw_plan)
  const int c = lane & 31, h = lane >> 5;

  // ------------------------------------------------------------------ producer state
  const int chl = lane >> 2, q = lane & 3;
  // A lane's 8-byte row window never leaves its row (so no load ever leaves the tensor): quad q fetches from column
  // {0, 3, 6, 6}[q] and two v_perm_b32 with per-lane selectors move the bytes to window positions (byte i = input column
  // 4 q - 1 + i), writing zeros (selector 0x0c) for the pad columns -1 and 14 ..: the same 2 VALU per row the masks cost.
  // Rows outside the image (row -1 of the upper half, row 14 of the lower one) are fetched from the neighbouring row
  // and meet a zeroed filter row.
  const uint32_t sel_lo = q == 0 ? 0x0201000cu : (q == 1 ? 0x03020100u : (q == 2 ? 0x04030201u : 0x0c070605u));
  const uint32_t sel_hi = q == 0 ? 0x06050403u : (q == 1 ? 0x07060504u : (q == 2 ? 0x0c070605u : 0x0c0c0c0cu));
  // byte offset of (image b, channel 16 wave + chl, input row 7 hf, the window's fetch column); a round advances it by 128
  // planes.  Strip row t is input row 7 hf - 1 + t: distance 14 (t - 1), an immediate; the two rows that can lie outside
  // the image get their own scalar base (the neighbouring row's address).
  uint32_t xoff = (uint32_t)(((b * C + 16 * wave + chl) * 14 + 7 * hf) * 14 + (q == 0 ? 0 : (q == 1 ? 3 : 6)));
  const uint32_t top = hf == 0 ? 0u : 0xffffffffu, bot = hf == 1 ? 0u : 0xffffffffu;  // scalar (block-uniform)
  const int8_t* const xs0 = a.x + (hf == 0 ? 0 : -14);   // strip row 0: input row 7 hf - 1, or row 0 again for the upper half
  const int8_t* const xs8 = a.x + (hf == 1 ? 84 : 98);   // strip row 8: input row 7 hf + 7, or row 13 again for the lower half
  // activation image address of (channel, output row o, quad): k = channel: K-step k / 32, kg = (k % 32) / 8, row k % 8,
  // chunk o in slot o ^ (2 ((k % 8) >> 1)) (the swizzle of gemm_wide_kernel.h), byte 4 q:  address = wbase ^ (o << 4)
  const int ch0 = 16 * wave + chl;
  uint32_t wbase = (uint32_t)((ch0 >> 5) * FW_KSTEP + ((ch0 >> 3) & 3) * 1024 + (ch0 & 7) * 128 + ((ch0 & 6) << 4) + 4 * q);
  const float dw_hi2 = a.dw_act == ACT_RELU6 ? fminf(a.dw_alpha + a.dw_alpha, 254.f) : 254.f;
  const float dw_leak = a.dw_act == ACT_LEAKY ? a.dw_alpha : 1.f;

  uint32_t in[9][2];   // the 9 input row windows of the task being produced / fetched
  uint32_t wr[3];      // its packed filter rows (w0, w1, w2, 0)
  float dsc, dbi;      // its doubled scale / bias
  auto load_row = [&](auto t_c, uint32_t off, uint32_t (&d)[2]) __attribute__((always_inline)) {
    constexpr int t = decltype(t_c)::value;
    if constexpr (t == 0) fw_load_row<0>(xs0, off, d);
    else if constexpr (t == 8) fw_load_row<0>(xs8, off, d);
    else fw_load_row<14 * (t - 1)>(a.x, off, d);
  };
  auto fetch_task = [&](int ch, uint32_t off) __attribute__((always_inline)) {
    // ch / off: this lane's channel and window offset of the task
    const int8_t* wp = a.dw_w + (size_t)ch * 9;
    uint32_t w0, w1, w2;
    __builtin_memcpy(&w0, wp, 4);
    __builtin_memcpy(&w1, wp + 3, 4);
    __builtin_memcpy(&w2, wp + 5, 4);
    wr[0] = w0 & 0xffffffu;
    wr[1] = w1 & 0xffffffu;
    wr[2] = w2 >> 8;
    const float s = a.dw_scale[ch], bb = a.dw_bias ? a.dw_bias[ch] : 0.f;
    dsc = s + s;
    dbi = bb + bb;
    load_row(std::integral_constant<int, 0>{}, off, in[0]);
    load_row(std::integral_constant<int, 1>{}, off, in[1]);
    load_row(std::integral_constant<int, 2>{}, off, in[2]);
    load_row(std::integral_constant<int, 3>{}, off, in[3]);
    load_row(std::integral_constant<int, 4>{}, off, in[4]);
    load_row(std::integral_constant<int, 5>{}, off, in[5]);
    load_row(std::integral_constant<int, 6>{}, off, in[6]);
    load_row(std::integral_constant<int, 7>{}, off, in[7]);
    load_row(std::integral_constant<int, 8>{}, off, in[8]);
  };

  // ------------------------------------------------------------------ consumer state
  // transposed-read addresses (gemm_wide_kernel.h): tile t <-> chunk pair (2t, 2t+1); lane 2q'+p of a 16-lane group -> row
  // q', sub-chunk p; 16-lane group parity -> chunk parity; k half h -> kg {2h, 2h+1}
  // n tile tt: slot pair 2 (tt ^ (q' >> 1)) + parity: address of tile tt = tr00 ^ (tt << 5) (bits 5-6 hold nothing else)
  uint32_t tr00;
  {
    const int qr = (lane & 15) >> 1, par = (lane >> 4) & 1;
    tr00 = (uint32_t)((h * 2) * 1024 + qr * 128 + ((2 * (qr >> 1) + par) * 16) + (lane & 1) * 8);
  }
  const int mt0 = wave * MTW;  // my first 32-row m tile
  const uint8_t* const wpk = reinterpret_cast<const uint8_t*>(g.wp) + (size_t)mt0 * KS * 1024;  // [mt][ks][64 lanes][16 B]: wave-uniform
  const uint32_t wlane = (uint32_t)lane * 16;
  v4i W[4][MTW];       // weight fragments of the 4 K-steps being multiplied / fetched
  auto fetch_w = [&](int j, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MTW; ++m) W[j][m] = *reinterpret_cast<const v4i*>(wpk + ((size_t)m * KS + ks) * 1024 + wlane);  // scalar base + lane offset
  };
  v16i acc[FW_NT][MTW];
#pragma unroll
  for (int n = 0; n < FW_NT; ++n)
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][m][r] = 0;

  // ------------------------------------------------------------------ one round
  // PRODUCE: the task in (in, wr, dsc, dbi) -> K-steps 4 rp .. 4 rp + 3 of the image, and the next task's operands fetched
  // into the same registers behind their last use.  CONSUME: K-steps 4 rc .. 4 rc + 3 (W) multiplied, the next four fetched.
  // The MFMAs are dealt over the 9 row chunks of the task so that every chunk carries VALU and matrix work side by side.
  using std::integral_constant;
  // ORDER: 1 (default) = TAPS FIRST: the 9 row chunks' window cuts and v_dot4 taps with no MFMA between them, then the 7
  // requantisation slices with the round's MFMAs dealt over them.  v_dot4_i32_i8 runs on the matrix pipe: beside an MFMA it
  // waits for it (tools/probe_coexec.hip: one wave's MFMA + 8 fma take 51 cycles, MFMA + 8 dot4 90; a dot4-only wave beside an
  // MFMA-only wave runs at 9.5 cycles per dot4), while cvt / fma / min / perm overlap with it.  0 = everything dealt over the
  // row chunks (the first form: rounds took MFMA time + VALU time, profiles/r04_fused_timeline_order0.txt).
  auto round = [&](auto produce_c, auto consume_c, auto order_c, int rp, int rc) __attribute__((always_inline)) {
    constexpr bool PRODUCE = decltype(produce_c)::value, CONSUME = decltype(consume_c)::value;
    constexpr int ORDER = decltype(order_c)::value;
    // next task (clamped: the last round fetches its own operands again, unused)
    const int rn = rp + 1 < R ? rp + 1 : rp;
    const int nch = 128 * rn + 16 * wave + chl;
    const uint32_t noff = xoff + (uint32_t)(rn - rp) * (128 * 196);
    const uint32_t wb = wbase + (uint32_t)rp * (4 * FW_KSTEP);
    const uint32_t rb = (uint32_t)rc * (4 * FW_KSTEP);
    int dacc[FW_TR][4];
    const uint32_t wr0t = wr[0] & top, wr2b = wr[2] & bot;  // input row 0 / 8 of the strip lies outside the image in the upper / lower half
    uint32_t nwr[3];
    float ndsc = 0.f, ndbi = 0.f;
    v2i lo[FW_NT], hi[FW_NT];
    auto read_frag = [&](int j, int n) __attribute__((always_inline)) {
      const uint32_t ta = (tr00 + rb) ^ (uint32_t)(n << 5);
      lo[n] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(fw_lds + ta + j * FW_KSTEP));
      hi[n] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(fw_lds + ta + j * FW_KSTEP + 1024));
    };
    // MFMA i of the round: K-step j = i / (4 MTW), n tile (i / MTW) % 4, m tile i % MTW
    auto mfma = [&](auto i_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      constexpr int j = i / (FW_NT * MTW), n = (i / MTW) % FW_NT, m = i % MTW;
      if constexpr (m == 0) {
        if constexpr (n == 0 && j == 0) {
          read_frag(0, 0);
          read_frag(0, 1);
        }
        // fragments two n tiles ahead
        constexpr int nn = (n + 2) % FW_NT, jn = j + (n + 2) / FW_NT;
        if constexpr (jn < 4) read_frag(jn, nn);
      }
      const v4i av = {lo[n][0], lo[n][1], hi[n][0], hi[n][1]};
      acc[n][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, W[j][m], acc[n][m], 0, 0, 0);
      if constexpr (n == FW_NT - 1 && m == MTW - 1) {  // K-step j done: its registers take K-step j of the next round
        const int ksn = 4 * (rc + 1) + j;
        if constexpr (!(EXP == 4 && PRODUCE)) fetch_w(j, ksn < KS ? ksn : KS - 1);
      }
    };
    auto mfmas = [&](auto self, auto i_c, auto end_c) __attribute__((always_inline)) -> void {
      constexpr int i = decltype(i_c)::value, end = decltype(end_c)::value;
      if constexpr (i < end) {
        mfma(integral_constant<int, i>{});
        self(self, integral_constant<int, i + 1>{}, integral_constant<int, end>{});
      }
    };
    constexpr int NM = 4 * FW_NT * MTW;  // MFMAs per round and wave: 32 / 16
    // first MFMA of row chunk t (t = 9: end): proportional to the chunk's VALU work (rows 0 / 1 carry no requantisation)
    constexpr auto mstart = [](int t) {
      constexpr int cum[10] = {0, 1, 3, 7, 11, 15, 19, 23, 28, 32};
      return cum[t] * NM / 32;
    };
    auto taps = [&](auto t_c) __attribute__((always_inline)) {
      constexpr int t = decltype(t_c)::value;
      const uint32_t e0 = __builtin_amdgcn_perm(in[t][1], in[t][0], sel_lo), e1 = __builtin_amdgcn_perm(in[t][1], in[t][0], sel_hi);
      uint32_t win[4];
      win[0] = e0;
      win[1] = __builtin_amdgcn_alignbyte(e1, e0, 1);
      win[2] = __builtin_amdgcn_alignbyte(e1, e0, 2);
      win[3] = __builtin_amdgcn_alignbyte(e1, e0, 3);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int o = t - r;
        if (o < 0 || o >= FW_TR) continue;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          dacc[o][jj] = r == 0 ? sdot4_first(win[jj], t == 0 ? wr0t : wr[0])
                               : __builtin_amdgcn_sdot4((int)win[jj], (int)(t == 8 ? wr2b : wr[r]), dacc[o][jj], false);
      }
      // the next task's row t into the registers just consumed
      if constexpr (!(EXP == 3 && CONSUME)) load_row(integral_constant<int, t>{}, noff, in[t]);
      if constexpr (t == 2) {
        const int8_t* wp = a.dw_w + (size_t)nch * 9;
        uint32_t w0, w1, w2;
        __builtin_memcpy(&w0, wp, 4);
        __builtin_memcpy(&w1, wp + 3, 4);
        __builtin_memcpy(&w2, wp + 5, 4);
        nwr[0] = w0 & 0xffffffu;
        nwr[1] = w1 & 0xffffffu;
        nwr[2] = w2 >> 8;
        ndsc = a.dw_scale[nch];
        ndbi = (a.dw_bias ? a.dw_bias : a.dw_scale)[nch];  // no branch inside the round
        if (!a.dw_bias) ndbi = 0.f;
      }
    };
    auto finish = [&](auto o_c) __attribute__((always_inline)) {  // output row o is complete: requantise, into the image
      constexpr int o = decltype(o_c)::value;
      const uint32_t pk = DWNN ? (NEWQ ? requant4_nn_rtz(dacc[o], dsc, dbi, dw_hi2, a.ones) : dw_requant4<ACT_RELU6>(dacc[o], dsc, dbi, 0.f, 0.f, dw_hi2, a.ones))
                               : dw_requant4<ACT_LEAKY>(dacc[o], dsc, dbi, dw_leak, -254.f, 254.f);
      *reinterpret_cast<uint32_t*>(fw_lds + (wb ^ (uint32_t)(o << 4))) = pk;
    };
    constexpr bool DO_P = PRODUCE && !(EXP == 2 && CONSUME), DO_C = CONSUME && !(EXP == 1 && PRODUCE);
    auto chunk = [&](auto t_c) __attribute__((always_inline)) {  // ORDER 0: row chunk t with its share of the MFMAs
      constexpr int t = decltype(t_c)::value;
      if constexpr (DO_C) mfmas(mfmas, integral_constant<int, mstart(t)>{}, integral_constant<int, mstart(t + 1)>{});
      if constexpr (DO_P) {
        taps(t_c);
        if constexpr (t >= 2) finish(integral_constant<int, t - 2>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // first MFMA of requantisation slice o (o = 7: end), ORDER 1
    constexpr auto m1start = [](int o) {
      constexpr int cum[8] = {0, 5, 10, 14, 19, 23, 28, 32};
      return cum[o] * NM / 32;
    };
    auto slice = [&](auto o_c) __attribute__((always_inline)) {  // ORDER 1: requantisation slice o with its share of the MFMAs
      constexpr int o = decltype(o_c)::value;
      if constexpr (DO_C) mfmas(mfmas, integral_constant<int, m1start(o)>{}, integral_constant<int, m1start(o + 1)>{});
      if constexpr (DO_P) finish(o_c);
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (ORDER == 0 || !DO_P) {
      chunk(integral_constant<int, 0>{});
      chunk(integral_constant<int, 1>{});
      chunk(integral_constant<int, 2>{});
      chunk(integral_constant<int, 3>{});
      chunk(integral_constant<int, 4>{});
      chunk(integral_constant<int, 5>{});
      chunk(integral_constant<int, 6>{});
      chunk(integral_constant<int, 7>{});
      chunk(integral_constant<int, 8>{});
    } else {
      taps(integral_constant<int, 0>{});
      taps(integral_constant<int, 1>{});
      taps(integral_constant<int, 2>{});
      taps(integral_constant<int, 3>{});
      taps(integral_constant<int, 4>{});
      taps(integral_constant<int, 5>{});
      taps(integral_constant<int, 6>{});
      taps(integral_constant<int, 7>{});
      taps(integral_constant<int, 8>{});
      __builtin_amdgcn_sched_barrier(0);
      slice(integral_constant<int, 0>{});
      slice(integral_constant<int, 1>{});
      slice(integral_constant<int, 2>{});
      slice(integral_constant<int, 3>{});
      slice(integral_constant<int, 4>{});
      slice(integral_constant<int, 5>{});
      slice(integral_constant<int, 6>{});
    }
    if constexpr (PRODUCE && !(EXP == 2 && CONSUME)) {
      wr[0] = nwr[0];
      wr[1] = nwr[1];
      wr[2] = nwr[2];
      dsc = ndsc + ndsc;
      dbi = ndbi + ndbi;
    }
  };

  fetch_task(ch0, xoff);
  fetch_w(0, 0);
  fetch_w(1, 1 < KS ? 1 : KS - 1);
  fetch_w(2, 2 < KS ? 2 : KS - 1);
  fetch_w(3, 3 < KS ? 3 : KS - 1);
  PLHIP_FW_STAMP(2);
  using OD = integral_constant<int, ORDER_T>;
  round(std::true_type{}, std::false_type{}, OD{}, 0, 0);
  xoff += (R > 1 ? 1 : 0) * 128 * 196;
  PLHIP_FW_STAMP(3);
  __syncthreads();
  for (int r = 1; r < R; ++r) {
    round(std::true_type{}, std::true_type{}, OD{}, r, r - 1);
    xoff += (r + 1 < R ? 1 : 0) * 128 * 196;
    if (r < 5) PLHIP_FW_STAMP(4 + 2 * (r - 1));
    __syncthreads();
    if (r < 5) PLHIP_FW_STAMP(5 + 2 * (r - 1));
  }
  float psc[MTW], pbi[MTW];  // pointwise scale / bias of this lane's channels: fetched under the last round's MFMAs
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    psc[m] = 1.f;
    pbi[m] = 0.f;
    if (OUT != OUT_I32) {
      psc[m] = g.scale[(mt0 + m) * 32 + c];
      if (g.bias) pbi[m] = g.bias[(mt0 + m) * 32 + c];
    }
  }
  // ------------------------------------------------------------------ last K-steps + epilogue
  // accumulator register r of n tile n: pixel 32 n + 8 (r >> 2) + 4 h + (r & 3) = output row 2 n + (r >> 3), column
  // 8 ((r >> 2) & 1) + 4 h + (r & 3) of the 16-wide chunk; lane (c, h) owns output channel 32 (mt0 + m) + c.
  const int orow0 = FW_TR * hf;
  if (OUT == OUT_I8) {
    // int8: the last round's 4 K-steps n-tile-major, so that tile n - 1 is requantised (fma / min / cvt: they overlap with the
    // matrix pipe) in the shadow of tile n's MFMAs; only the last tile's epilogue is exposed
    const float hi2 = g.act == ACT_RELU6 ? fminf(g.alpha + g.alpha, 254.f) : 254.f;
    const float leak = g.act == ACT_LEAKY ? g.alpha : 1.f;
    uint8_t* stg = fw_lds + (size_t)KS * FW_KSTEP + wave * (32 * MTW * FW_SP);
    const uint32_t rbf = (uint32_t)(R - 1) * (4 * FW_KSTEP);
    v2i flo[2][4], fhi[2][4];  // fragments of the 4 K-steps of one n tile, two sets
    auto read_tile = [&](auto n_c) __attribute__((always_inline)) {
      constexpr int n = decltype(n_c)::value;
      const uint32_t ta = (tr00 + rbf) ^ (uint32_t)(n << 5);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        flo[n & 1][j] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(fw_lds + ta + j * FW_KSTEP));
        fhi[n & 1][j] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(fw_lds + ta + j * FW_KSTEP + 1024));
      }
    };
    auto mfma_tile = [&](auto n_c) __attribute__((always_inline)) {
      constexpr int n = decltype(n_c)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const v4i av = {flo[n & 1][j][0], flo[n & 1][j][1], fhi[n & 1][j][0], fhi[n & 1][j][1]};
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[n][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, W[j][m], acc[n][m], 0, 0, 0);
      }
    };
    auto epi_tile = [&](auto n_c) __attribute__((always_inline)) {
      constexpr int n = decltype(n_c)::value;
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const float s2 = psc[m] + psc[m], b2 = pbi[m] + pbi[m];
        uint32_t edw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          if (n == FW_NT - 1 && gq >= 2) continue;  // row 7 of the tile does not exist
          int v[4] = {acc[n][m][4 * gq], acc[n][m][4 * gq + 1], acc[n][m][4 * gq + 2], acc[n][m][4 * gq + 3]};
          edw[gq] = PWNN ? (NEWQ ? requant4_nn_rtz(v, s2, b2, hi2, a.ones) : dw_requant4<ACT_RELU6>(v, s2, b2, 0.f, 0.f, hi2, a.ones))
                         : dw_requant4<ACT_LEAKY>(v, s2, b2, leak, -254.f, 254.f);
        }
        // half exchange: every lane gets the 16 pixels of ONE output row of its channel (h = 0: row 2 n, h = 1: row 2 n + 1)
        auto s02 = __builtin_amdgcn_permlane32_swap(edw[0], edw[2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(edw[1], edw[3], false, false);
        const uint32_t d0 = s02[0], d1 = s02[1], d2 = s13[0], d3 = s13[1];
        if (n < FW_NT - 1 || h == 0) {
          // 14 bytes at a 2-byte aligned position: halfword writes
          uint16_t* p = reinterpret_cast<uint16_t*>(stg + (32 * m + c) * FW_SP + (2 * n + h) * 14);
          p[0] = (uint16_t)d0; p[1] = (uint16_t)(d0 >> 16);
          p[2] = (uint16_t)d1; p[3] = (uint16_t)(d1 >> 16);
          p[4] = (uint16_t)d2; p[5] = (uint16_t)(d2 >> 16);
          p[6] = (uint16_t)d3;
        }
      }
    };
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    using I2 = integral_constant<int, 2>;
    using I3 = integral_constant<int, 3>;
    read_tile(I0{});
    read_tile(I1{});
    mfma_tile(I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_tile(I2{});
    mfma_tile(I1{});
    epi_tile(I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_tile(I3{});
    mfma_tile(I2{});
    epi_tile(I1{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_tile(I3{});
    epi_tile(I2{});
    __builtin_amdgcn_sched_barrier(0);
    PLHIP_FW_STAMP(12);
    epi_tile(I3{});
    PLHIP_FW_STAMP(13);
    // copy-out: lane -> (row lane >> 3 of a group of 8, 16-byte piece lane & 7); a channel row is 98 contiguous bytes
    const int piece = lane & 7, rsub = lane >> 3;
    int8_t* ybase = reinterpret_cast<int8_t*>(g.y) + ((size_t)b * g.M + mt0 * 32) * 196 + 98 * hf + piece * 16;
#pragma unroll
    for (int i = 0; i < 4 * MTW; ++i) {
      const int row = 8 * i + rsub;
      const v4i v = *reinterpret_cast<const v4i*>(stg + row * FW_SP + (piece < 7 ? piece : 6) * 16);
      int8_t* yp = ybase + (size_t)row * 196;
      if (piece < 6) __builtin_memcpy(yp, &v, 16);  // possibly unaligned: fine for global memory
      else if (piece == 6) {
        const uint16_t t2 = (uint16_t)v[0];
        __builtin_memcpy(yp, &t2, 2);
      }
    }
  } else {
    round(std::false_type{}, std::true_type{}, OD{}, R, R - 1);
    PLHIP_FW_STAMP(12);
    const float fcap = g.act == ACT_RELU6 ? g.alpha : __builtin_huge_valf();
    const float flo = (g.act == ACT_RELU || g.act == ACT_RELU6) ? 0.f : -__builtin_huge_valf();
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const size_t chan = ((size_t)b * g.M + (mt0 + m) * 32 + c) * 196;
#pragma unroll
      for (int n = 0; n < FW_NT; ++n) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int o = 2 * n + (gq >> 1);
          if (o >= FW_TR) continue;
          const int col = 8 * (gq & 1) + 4 * h;
          const size_t off = chan + (size_t)(orow0 + o) * 14 + col;
          if (OUT == OUT_F32) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float y = __fmaf_rn((float)acc[n][m][4 * gq + e], psc[m], pbi[m]);
              if (g.act == ACT_LEAKY) y = y > 0.f ? y : g.alpha * y;
              f[e] = fminf(fmaxf(y, flo), fcap);
            }
            float* yp = reinterpret_cast<float*>(g.y) + off;
            if (col + 3 < 14) {
              const v4f v = {f[0], f[1], f[2], f[3]};
              __builtin_memcpy(yp, &v, 16);
            } else if (col < 14) {
              yp[0] = f[0];
              yp[1] = f[1];
            }
          } else {
            int* yp = reinterpret_cast<int*>(g.y) + off;
            if (col + 3 < 14) {
              const v4i v = {acc[n][m][4 * gq], acc[n][m][4 * gq + 1], acc[n][m][4 * gq + 2], acc[n][m][4 * gq + 3]};
              __builtin_memcpy(yp, &v, 16);
            } else if (col < 14) {
              yp[0] = acc[n][m][4 * gq];
              yp[1] = acc[n][m][4 * gq + 1];
            }
          }
        }
      }
    }
  }
  PLHIP_FW_STAMP(14);
  if (diag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PLHIP_FW_STAMP(15);
  }
}

// Fills the launch plan and says whether the shape is inside the fused path: 3x3, stride 1, dilation 1, pad 1 on a 14 x 14
// plane, 128 | C <= 512 (whole rounds; the K x 128 activation image + the staging image fit the LDS), M = 256 or 512 (one or
// two m tiles per wave), depthwise activation relu / relu6 / none / leaky, any pointwise activation.
bool fused_dwpw_plan(FusedArgs* a, int kh, int kw, int sh, int sw, int dh, int dw, int out) {
  if (!(kh == 3 && kw == 3 && sh == sw && (sh == 1 || sh == 2) && dh == 1 && dw == 1)) return false;
  a->ones = 0x01010101u;
  a->stream = 0;
  const int fs = knob("FUSED_STREAM", 1);  // 1: every shape of the streaming kernel, 2: stride 1 only, 3: not the 14-wide plane, 0: off
  // (the plane average as output, OUT_GAP, exists on the small-plane kernel only: nothing else may accept it)
  if (out != OUT_GAP && fs && (sh == 1 || fs != 2) && !(fs == 3 && a->ow == 14) && fused_stream_supported(*a)) {  // the large planes: fused_dwpw_stream.hip
    a->stream = 1;
    return true;
  }
  if (knob("FUSED_SMALL", 1) && fused_small_supported(*a)) {  // the 7 x 7 planes: fused_dwpw_small.hip
    a->stream = 2;
    return true;
  }
  if (out == OUT_GAP) return false;  // the plane average: the small-plane kernel only
  if (sh != 1) return false;
  if (!(a->h == 14 && a->w == 14 && a->oh == 14 && a->ow == 14 && a->pt == 1 && a->pl == 1)) return false;
  if (a->C % 128 != 0 || a->C < 128 || a->C > 512) return false;
  if (a->pw.M != 256 && a->pw.M != 512) return false;
  if (a->n < 1 || (long)a->n * a->C * 196 >= ((long)1 << 31) - 65536 || (long)a->n * a->pw.M * 196 >= ((long)1 << 31)) return false;
  (void)out;
  a->tiles = 2 * a->n;
  a->ones = 0x01010101u;
  return true;
}

template <int MTW, int OUT>
static void launch_fused_t(const FusedArgs& a, hipStream_t s) {
  const unsigned blocks = (unsigned)((a.tiles + 7) / 8 * 8);
  const size_t lds = (size_t)a.pw.KS * FW_KSTEP + (OUT == OUT_I8 ? (size_t)8 * 32 * MTW * FW_SP : 0);
  const bool dwnn = a.dw_act == ACT_RELU || a.dw_act == ACT_RELU6;
  const bool pwnn = OUT == OUT_I8 && (a.pw.act == ACT_RELU || a.pw.act == ACT_RELU6);
#define PLHIP_FW_LAUNCH(DN, PN)                                                                                  \
  do {                                                                                                           \
    auto kfn = fused_dwpw14_kernel<MTW, OUT, DN, PN>;                                                            \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);                                                 \
  } while (0)
  if (OUT == OUT_I8 && MTW == 2 && dwnn && pwnn && (a.pw.dbg & 12)) {
    if ((a.pw.dbg & 12) == 4) {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 0, 0, 1>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    } else if ((a.pw.dbg & 12) == 8) {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 0, 1, 0>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    } else {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 0, 0, 0>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    }
    return;
  }
  if (OUT == OUT_I8 && MTW == 2 && dwnn && pwnn && (a.pw.dbg & 3) == 3) {
    if (a.pw.dbg & 16) {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 4>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    } else {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 3>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    }
    return;
  }
  if (OUT == OUT_I8 && MTW == 2 && dwnn && pwnn && (a.pw.dbg & 3)) {  // timing experiments (plhip_debug_set("fused_exp", 1 | 2))
    if (a.pw.dbg & 1) {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 1>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    } else {
      auto kfn = fused_dwpw14_kernel<MTW, OUT, true, true, 2>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
    }
    return;
  }
  if (OUT == OUT_I8) {
    if (dwnn && pwnn) PLHIP_FW_LAUNCH(true, true);
    else if (dwnn) PLHIP_FW_LAUNCH(true, false);
    else if (pwnn) PLHIP_FW_LAUNCH(false, true);
    else PLHIP_FW_LAUNCH(false, false);
  } else {
    if (dwnn) PLHIP_FW_LAUNCH(true, false);
    else PLHIP_FW_LAUNCH(false, false);
  }
#undef PLHIP_FW_LAUNCH
}

// `a` must have passed fused_dwpw_plan with the same `out`.
void launch_fused_dwpw(const FusedArgs& a_in, int out, hipStream_t s) {
  FusedArgs a = a_in;
  a.pw.dbg = g_fw_debug;
  if (a.stream == 2) {
    launch_fused_small(a, out, s);
    return;
  }
  if (a.stream) {
    launch_fused_stream(a, out, s);
    return;
  }
  if (a.pw.M == 512) {
    if (out == OUT_I32) launch_fused_t<2, OUT_I32>(a, s);
    else if (out == OUT_F32) launch_fused_t<2, OUT_F32>(a, s);
    else launch_fused_t<2, OUT_I8>(a, s);
  } else {
    if (out == OUT_I32) launch_fused_t<1, OUT_I32>(a, s);
    else if (out == OUT_F32) launch_fused_t<1, OUT_F32>(a, s);
    else launch_fused_t<1, OUT_I8>(a, s);
  }
}

}  // namespace plhip
