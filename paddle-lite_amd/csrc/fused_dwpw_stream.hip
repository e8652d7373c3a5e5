// fused_dwpw_stream.hip — depthwise 3x3 stride 1 [int8_out] + pointwise 1x1 in ONE launch for the LARGE planes of the
// MobileNet programs (112 x 112, 56 x 56, 28 x 28), where the pair is bound by HBM bytes: run as two kernels the int8 tensor
// between them is written and read back (103 MB of 257 MB for 32 -> 64 @112 at batch 128); here it never leaves the CU.
// Replaces the instruction pair DepthwiseConv<kInt8,kInt8>::Run (lite/kernels/arm/conv_depthwise.cc:407-446 ->
// conv3x3s1_depthwise_int8.cc:33-447) ; GemmLikeConv<kInt8,*>::Run (lite/kernels/arm/conv_gemmlike.cc:399-462 ->
// gemm_prepacked_int8.cc:2582-2744); results bit-identical to the two kernels.
//
// The 14 x 14 pairs (fused_dwpw_i8.hip) are latency / issue bound and keep the whole K x tile image of one big block per CU
// with produce and consume software-pipelined over K rounds.  These pairs are byte bound, their K and M are small, and a
// plane has thousands of tiles: so the structure is the plain one and the overlap comes from SEVERAL SMALL BLOCKS PER CU in
// different phases (one block fetching, one on the VALU, one on the matrix pipe):
//   * tile = TR whole output rows of one image = 224 pixels = 7 MFMA n tiles (TR = 4 / 8 for W = 56 / 28; 448 pixels = 4 rows for W = 112): the output
//     of a tile is 224 CONTIGUOUS bytes per channel, 16-byte aligned; block = 4 waves, grid = images x tiles;
//   * produce: lane = (channel, RS-row strip, column quad), quads of a row on consecutive lanes (coalesced 8-byte row windows,
//     the next iteration's rows fetched under the current one's arithmetic): the strip body of fused_dwpw_i8.hip (in-bounds
//     windows placed by v_perm_b32, taps on v_dot4_i32_i8, the reference's requantisation) writes each requantised dword into
//     the activation image in LDS: image[k][288 B]: pixel-linear rows, pitch 288 = 72 dwords = 8 (mod 64) so that the 8 rows x
//     2 chunks a half-wave's ds_read_b64_tr_b8 touches fall into distinct banks without a swizzle;
//   * one barrier; consume: a wave owns M / 4 output channels (m tiles) for all 7 n tiles (M = 64: 2 m x 2 n splits), its weight
//     fragments straight from L2 (each byte of the weights once per block), K-outer; epilogue: requantise, two
//     v_permlane32_swap give a lane 16 consecutive pixels of one channel = one aligned 16-byte store.
#include <stdlib.h>

#include <type_traits>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"
#include "gemm_tr_common.h"

namespace plhip {

typedef float v2f __attribute__((ext_vector_type(2)));

// TP pixels per tile (224 or 448 = 7 or 14 n tiles); LDS bytes per channel row of the activation image: TP + pad with
// pitch / 4 = 8 (mod 16): 160 / 288 / 544
constexpr int fs_pitch(int tp) { return tp == 128 ? 160 : (tp == 224 ? 288 : 544); }

// diagnostic timeline (plhip_debug_set("fused_stamps", 1)): per wave of the first 2048 tiles: 0 realtime, 1 entry, 2 operands of
// the first PD iterations requested, 3 produced, 4 behind the barrier, 5 multiplied, 6 stores issued, 7 realtime end
constexpr int FS_STAMP_SLOTS = 8;
__device__ unsigned long long g_fs_stamps[2048 * 4 * FS_STAMP_SLOTS];
int debug_read_fs_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(g_fs_stamps)) bytes = sizeof(g_fs_stamps);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fs_stamps), bytes) == hipSuccess ? 0 : -1;
}
#define PLHIP_FS_STAMP(i)                                                                                                  \
  do {                                                                                                                     \
    if (diag && lane == 0) g_fs_stamps[((size_t)vb * 4 + wave) * FS_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memtime();    \
  } while (0)

// W: plane width; K, M: channels in / out; RS: output rows per strip (TR = 224 / W rows per tile, TR % RS == 0); PD: iterations
// of operands in flight
// S: stride of the depthwise stage (1 or 2; W and the tile are those of the OUTPUT plane, the input plane is S W wide)
// MP: passes over the output channels (M / MP per pass: the accumulators of one pass are what the registers hold)
// W % 4 != 0 (the 14-wide plane): a row takes RP = 16 slots of the image (2 of them junk: computed from the neighbouring bytes,
// multiplied, never stored), a tile is 7 rows = 112 of 128 slots
template <int W, int K, int M, int TP, int RS, int PD, int S, int MP, int OUT, bool DWNN, bool PWNN>
__global__ __launch_bounds__(256, M / MP >= 256 ? 2 : 3) void fused_dwpw_stream_kernel(FusedArgs a) {
  constexpr int FS_TP = TP, FS_NT = TP / 32, FS_PITCH = fs_pitch(TP);
  static_assert((FS_PITCH / 4) % 16 == 8 && FS_PITCH >= TP, "image pitch");
  constexpr int RP = (W + 3) / 4 * 4;                       // slots per row
  constexpr int TR = W == 14 ? 7 : FS_TP / RP, NS = TR / RS, QW = RP / 4;   // rows per tile, strips per tile, quads per row
  static_assert(RP == W || (S == 2 && RP == 16), "partial quads: the stride-2 form only");
  constexpr int G = 64 / QW;                                // (channel, strip) groups per wave and iteration
  constexpr int NGRP = K * NS;                              // groups per tile
  constexpr int NIT = (NGRP + 4 * G - 1) / (4 * G);         // iterations
  constexpr int KS = K / 32, MT = M / 32 / MP;              // (m tiles per pass)
  constexpr int MSPLIT = MT >= 4 ? 4 : MT;                  // waves along M
  constexpr int NSPLIT = 4 / MSPLIT;                        // waves along the n tiles
  constexpr int MW = MT / MSPLIT;                           // m tiles per wave and pass
  constexpr int NW = (FS_NT + NSPLIT - 1) / NSPLIT;         // n tiles per wave (the last split may own fewer)
  constexpr int NIN = S == 1 ? RS + 2 : 2 * RS + 1;         // input rows of a strip
  constexpr int ND = S == 1 ? 2 : 3;                        // dwords fetched per input row
  constexpr int WI = S * W;                                 // input plane width
  static_assert(S == 1 || S == 2, "stride");
  static_assert(TR * RP <= FS_TP && TR % RS == 0 && K % 32 == 0 && M % (32 * MP) == 0 && MT % MSPLIT == 0, "geometry");
  const GemmArgs& g = a.pw;
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.dw_w); PLHIP_PRELOAD(a.dw_scale); PLHIP_PRELOAD(a.dw_bias); PLHIP_PRELOAD(a.dw_act);
  PLHIP_PRELOAD(a.dw_alpha); PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.h); PLHIP_PRELOAD(a.tiles); PLHIP_PRELOAD(a.ones); PLHIP_PRELOAD(g.wp);
  PLHIP_PRELOAD(g.y); PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias); PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha); PLHIP_PRELOAD(g.NT);
  extern __shared__ __attribute__((aligned(16))) uint8_t fs_lds[];  // image[K][FS_PITCH], then the depthwise parameters [K][32 B]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-contiguous tiles (neighbouring tiles of an image share their halo rows)
  const unsigned nb = (unsigned)a.tiles, per = (nb + 7) >> 3;
  const unsigned vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nb) return;  // block-uniform
  const int H = a.oh, TPI = g.NT;  // output rows; tiles per image (launcher)
  const int b = (int)(vb / (unsigned)TPI), ti = (int)(vb - (unsigned)b * TPI), tr0 = ti * TR;
  const int c = lane & 31, h = lane >> 5;
  const bool diag = (g.dbg & 32) != 0 && vb < 2048;
  if (diag && lane == 0) g_fs_stamps[((size_t)vb * 4 + wave) * FS_STAMP_SLOTS] = __builtin_amdgcn_s_memrealtime();
  PLHIP_FS_STAMP(1);

  // ------------------------------------------------------------------ produce
  const int gl = lane / QW, q = lane - gl * QW;  // group inside the iteration, column quad
  const bool active = gl < G;
  // window fetch column and byte selectors (fused_dwpw_i8.hip): the 8 bytes never leave the row
  const int colq = q == 0 ? 0 : (q == QW - 1 ? W - 8 : 4 * q - 1);
  const uint32_t sel_lo = q == 0 ? 0x0201000cu : (q == QW - 1 ? 0x06050403u : 0x03020100u);
  const uint32_t sel_hi = q == 0 ? 0x06050403u : (q == QW - 1 ? 0x0c0c0c07u : 0x07060504u);
  const float dw_hi2 = a.dw_act == ACT_RELU6 ? fminf(a.dw_alpha + a.dw_alpha, 254.f) : 254.f;
  const float dw_leak = a.dw_act == ACT_LEAKY ? a.dw_alpha : 1.f;
  const uint32_t plane = (uint32_t)H * W, plane_in = (uint32_t)a.h * WI;
  // stride 2: a quad of 4 outputs reads input columns 8 q - 1 .. 8 q + 7.  The lane fetches the 12 aligned bytes from column
  // 8 q - 4 (d0 d1 d2): output j's window (columns 8 q + 2 j - 1 .. + 1) is bytes 3 + 2 j .. of them: three v_alignbyte and,
  // for j = 3, d2 itself against the filter row moved up one byte.  Column -1 (q = 0) is the left padding: masked.  No window
  // crosses the right or the bottom border (even planes, pad 1).  The only bytes outside the tensor would be the 4 in front of
  // its very first row: that one lane fetches from column 0 instead and moves its dwords up by one.  On the 14-wide plane the
  // last quad's fetch runs 4 bytes into the next row: the tensor's very last row is fetched 4 bytes early and moved down.
  const uint32_t m0 = q == 0 ? 0xffffff00u : 0xffffffffu;
  constexpr bool OVER = RP > W;
  const int x_last = (int)((uint32_t)a.n * K * plane_in) - 12;  // last offset a 12-byte fetch may start at

  // PD iterations of operands in flight (a ring of PD register sets): an iteration is ~100-150 VALU, a fetch from HBM under
  // load ~2 us: with one iteration ahead every wave waited for its rows (first form: 61 / 45 / 33 us for the three pairs)
  uint32_t in[PD][NIN][ND];  // row windows
  // depthwise parameters of all K channels, once per block, into LDS: (w0 w1 w2 0 | w3 w4 w5 0 | w6 w7 w8 0 | 2 scale | 2 bias):
  // per iteration a lane then reads ONE 16-byte and one 4-byte LDS word instead of five global loads (the vector-memory
  // instructions were the startup cost of a block: 44 per wave, 6.7 k cycles to issue with four blocks per CU)
  uint8_t* const prm = fs_lds + (size_t)K * FS_PITCH;
  static_assert(K <= 256, "one channel's parameters per thread");
  uint32_t pw0 = 0, pw1 = 0, pw2 = 0;
  float psc0 = 0.f, pbi0 = 0.f;
  if ((int)threadIdx.x < K) {  // requested first; stored behind the row fetches below (nothing in front of them waits)
    const int8_t* wp = a.dw_w + (size_t)threadIdx.x * 9;
    __builtin_memcpy(&pw0, wp, 4);
    __builtin_memcpy(&pw1, wp + 3, 4);
    __builtin_memcpy(&pw2, wp + 5, 4);
    psc0 = a.dw_scale[threadIdx.x];
    pbi0 = (a.dw_bias ? a.dw_bias : a.dw_scale)[threadIdx.x];
  }
  auto task = [&](int it, int& ch, int& strip, int& r0) {
    int gi = (it * 4 + wave) * G + (active ? gl : 0);
    if (gi >= NGRP) gi = NGRP - 1;  // surplus groups of the last iteration recompute the last one (same values, same place)
    ch = gi / NS;
    strip = gi - ch * NS;
    r0 = tr0 + strip * RS;          // first output row of the strip
    if (r0 > H - RS) r0 = H - RS;   // rows past the image (the last tile of a 28-row plane): any valid strip, never stored
  };
  using std::integral_constant;
  auto fetch = [&](auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value, s = it % PD;
    int ch, strip, r0;
    task(it, ch, strip, r0);
    const uint8_t* xs = reinterpret_cast<const uint8_t*>(a.x);
    if constexpr (S == 1) {
      const uint32_t off0 = (uint32_t)(b * K + ch) * plane + (uint32_t)r0 * W + colq;  // input row r0
      const bool top = r0 == 0, bot = r0 + RS == H;
      __builtin_memcpy(in[s][0], xs + (top ? off0 : off0 - W), 8);
#pragma unroll
      for (int t = 1; t <= RS; ++t) __builtin_memcpy(in[s][t], xs + off0 + (t - 1) * W, 8);
      __builtin_memcpy(in[s][RS + 1], xs + (bot ? off0 + (RS - 1) * W : off0 + RS * W), 8);
    } else {
      const int off0 = (int)((uint32_t)(b * K + ch) * plane_in + (uint32_t)(2 * r0) * WI) + 8 * q - 4;  // input row 2 r0
      const int o1 = off0 < 0 ? 0 : off0;  // (the tensor's first row, first quad)
      __builtin_memcpy(in[s][0], xs + (r0 == 0 ? o1 : off0 - WI), 12);
      __builtin_memcpy(in[s][1], xs + o1, 12);
#pragma unroll
      for (int t = 2; t < NIN; ++t) {
        int o = off0 + (t - 1) * WI;
        if (OVER && t == NIN - 1) o = o > x_last ? x_last : o;
        __builtin_memcpy(in[s][t], xs + o, 12);
      }
    }
  };
  auto compute = [&](auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value, s = it % PD;
    int ch, strip, r0;
    task(it, ch, strip, r0);
    const uint32_t zt = r0 == 0 ? 0u : 0xffffffffu, zb = r0 + RS == H ? 0u : 0xffffffffu;  // the strip's first / last input row outside the image
    // image address of (channel, strip row 0, quad); the lanes beyond the last group write into a sink behind the parameters
    // (a predicated store splits the body into basic blocks: an exec save / restore per store and nothing scheduled across)
    // (not in the 255-register forms: 256 -> 256 @28 went 28.3 -> 33.8 us with it)
    constexpr bool SINK = M / MP < 256;
    const uint32_t ldsa = (uint32_t)ch * FS_PITCH + (uint32_t)(strip * RS) * RP + 4 * q;
    const uint32_t ldsw = SINK && !active ? (uint32_t)(K * FS_PITCH + K * 32) + 4 * (lane & 7) : ldsa;
    int dacc[RS][4];
    const v4i pv = *reinterpret_cast<const v4i*>(prm + ch * 32);
    const uint32_t wr[3] = {(uint32_t)pv[0], (uint32_t)pv[1], (uint32_t)pv[2]};  // packed filter rows (w0, w1, w2, 0)
    const uint32_t w0t = wr[0] & zt, w2b = S == 1 ? wr[2] & zb : wr[2];
    const uint32_t wu[3] = {wr[0] << 8, wr[1] << 8, wr[2] << 8};  // (stride 2: the last output's window sits one byte up)
    const uint32_t w0tu = w0t << 8;
    const bool fix = S == 2 && b == 0 && ch == 0 && q == 0 && r0 == 0;
    const bool fixe = OVER && b == a.n - 1 && ch == K - 1 && q == QW - 1 && r0 + RS == H;
    const float dsc = __uint_as_float((uint32_t)pv[3]), dbi = __uint_as_float(*reinterpret_cast<const uint32_t*>(prm + ch * 32 + 16));
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
      uint32_t win[4];
      if constexpr (S == 1) {
        const uint32_t e0 = __builtin_amdgcn_perm(in[s][t][1], in[s][t][0], sel_lo), e1 = __builtin_amdgcn_perm(in[s][t][1], in[s][t][0], sel_hi);
        win[0] = e0;
        win[1] = __builtin_amdgcn_alignbyte(e1, e0, 1);
        win[2] = __builtin_amdgcn_alignbyte(e1, e0, 2);
        win[3] = __builtin_amdgcn_alignbyte(e1, e0, 3);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int o = t - r;
          if (o < 0 || o >= RS) continue;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            dacc[o][jj] = r == 0 ? sdot4_first(win[jj], t == 0 ? w0t : wr[0])
                                 : __builtin_amdgcn_sdot4((int)win[jj], (int)(t == NIN - 1 ? w2b : wr[r]), dacc[o][jj], false);
        }
      } else {
        uint32_t d0 = in[s][t][0], d1 = in[s][t][1], d2 = in[s][t][2];
        if (t < 2) {  // the lane that fetched from column 0
          d2 = fix ? d1 : d2;
          d1 = fix ? d0 : d1;
        }
        if (OVER && t == NIN - 1) {  // the lane that fetched 4 bytes early
          d0 = fixe ? d1 : d0;
          d1 = fixe ? d2 : d1;
        }
        win[0] = __builtin_amdgcn_alignbyte(d1, d0, 3) & m0;
        win[1] = __builtin_amdgcn_alignbyte(d2, d1, 1);
        win[2] = __builtin_amdgcn_alignbyte(d2, d1, 3);
        win[3] = d2;  // against the filter row moved up one byte
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if ((t - r) & 1) continue;
          const int o = (t - r) / 2;
          if (t - r < 0 || o >= RS) continue;
          const uint32_t f = t == 0 ? w0t : wr[r], fu = t == 0 ? w0tu : wu[r];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            dacc[o][jj] = r == 0 ? sdot4_first(win[jj], jj == 3 ? fu : f)
                                 : __builtin_amdgcn_sdot4((int)win[jj], (int)(jj == 3 ? fu : f), dacc[o][jj], false);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < RS; ++o) {
      const uint32_t pk = DWNN ? requant4_nn_rtz(dacc[o], dsc, dbi, dw_hi2, a.ones)
                               : dw_requant4<ACT_LEAKY>(dacc[o], dsc, dbi, dw_leak, -254.f, 254.f);
      if (SINK || active) *reinterpret_cast<uint32_t*>(fs_lds + ldsw + o * RP) = pk;
    }
  };
  auto prime = [&](auto self, auto it_c) __attribute__((always_inline)) -> void {
    constexpr int it = decltype(it_c)::value;
    if constexpr (it < PD && it < NIT) {
      fetch(it_c);
      self(self, integral_constant<int, it + 1>{});
    }
  };
  prime(prime, integral_constant<int, 0>{});
  PLHIP_FS_STAMP(2);
  // ---- the consumer's first operands, requested here so that they arrive under the depthwise arithmetic
  // wave -> (m split, n split): per pass m tiles [(mp MSPLIT + ms) MW, + MW), n tiles [ns NW, min(FS_NT, ns NW + NW))
  const int ms = wave % MSPLIT, ns = wave / MSPLIT;
  const int n0 = ns * NW;
  const int nmine = n0 + NW <= FS_NT ? NW : FS_NT - n0;  // wave-uniform
  // transposed read: lane 2 q' + p of a 16-lane group -> row q' (k % 8), sub-chunk p; 16-lane group parity -> 16-pixel chunk
  // parity; k half h -> kg {2h, 2h + 1}
  const uint32_t trb = (uint32_t)(((h * 2) * 8 + ((lane & 15) >> 1)) * FS_PITCH + ((lane >> 4) & 1) * 16 + (lane & 1) * 8 + n0 * 32);
  const uint32_t wlane = (uint32_t)lane * 16;
  // weight fragments WD - 1 K-steps ahead (from L2: ~1 us each; two slots where the accumulators leave no registers)
  constexpr int WD = KS < 4 || NW * MW * 16 > 112 ? 2 : 4;
  v4i Wf[WD][MW];
  float psc[2][MW], pbi[2][MW];  // [pass parity]
  auto pass_w = [&](int mp) { return reinterpret_cast<const uint8_t*>(g.wp) + (size_t)((mp * MSPLIT + ms) * MW) * KS * 1024; };  // [mt][ks][64 lanes][16 B]
  auto pass_operands = [&](auto mp_c) __attribute__((always_inline)) {
    constexpr int mp = decltype(mp_c)::value;
    const uint8_t* const wpk = pass_w(mp);
    const int mt0 = (mp * MSPLIT + ms) * MW;
#pragma unroll
    for (int u = 0; u < WD - 1 && u < KS; ++u)
#pragma unroll
      for (int m = 0; m < MW; ++m) Wf[u][m] = *reinterpret_cast<const v4i*>(wpk + ((size_t)m * KS + u) * 1024 + wlane);
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      psc[mp & 1][m] = 1.f;
      pbi[mp & 1][m] = 0.f;
      if (OUT != OUT_I32) {
        psc[mp & 1][m] = g.scale[(mt0 + m) * 32 + c];
        pbi[mp & 1][m] = (g.bias ? g.bias : g.scale)[(mt0 + m) * 32 + c];
        if (!g.bias) pbi[mp & 1][m] = 0.f;
      }
    }
  };
  pass_operands(integral_constant<int, 0>{});
  if ((int)threadIdx.x < K) {
    uint32_t* o = reinterpret_cast<uint32_t*>(prm + threadIdx.x * 32);
    const v4i pv = {(int)(pw0 & 0xffffffu), (int)(pw1 & 0xffffffu), (int)(pw2 >> 8), (int)__float_as_uint(psc0 + psc0)};
    *reinterpret_cast<v4i*>(o) = pv;
    o[4] = a.dw_bias ? __float_as_uint(pbi0 + pbi0) : 0u;
  }
  __syncthreads();  // the parameters are in LDS (the row fetches above are in flight meanwhile)
  auto steps = [&](auto self, auto it_c) __attribute__((always_inline)) -> void {
    constexpr int it = decltype(it_c)::value;
    if constexpr (it < NIT) {
      compute(it_c);
      if constexpr (it + PD < NIT) fetch(integral_constant<int, it + PD>{});
      self(self, integral_constant<int, it + 1>{});
    }
  };
  steps(steps, integral_constant<int, 0>{});
  PLHIP_FS_STAMP(3);
  __syncthreads();
  PLHIP_FS_STAMP(4);

  // ------------------------------------------------------------------ consume, MP passes over the output channels
  const int vrows = H - tr0 < TR ? H - tr0 : TR;  // valid rows of this tile
  const int vpx = vrows * W;                      // (RP == W: valid pixels, a multiple of 16)
  const float hi2 = g.act == ACT_RELU6 ? fminf(g.alpha + g.alpha, 254.f) : 254.f;
  const float leak = g.act == ACT_LEAKY ? g.alpha : 1.f;
  const float fcap = g.act == ACT_RELU6 ? g.alpha : __builtin_huge_valf();
  const float flo = (g.act == ACT_RELU || g.act == ACT_RELU6) ? 0.f : -__builtin_huge_valf();
  auto pass = [&](auto self, auto mp_c) __attribute__((always_inline)) -> void {
    constexpr int mp = decltype(mp_c)::value;
    if constexpr (mp < MP) {
      const uint8_t* const wpk = pass_w(mp);
      const int mt0 = (mp * MSPLIT + ms) * MW;
      v16i acc[NW][MW];
      // (the first K-step multiplies into the constant 0, an inline operand of the MFMA: zeroing the accumulators first was
      // 16 v_mov per tile, 110-130 per wave = 4-6 % of everything these kernels issue)
      const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + WD - 1 < KS) {
#pragma unroll
          for (int m = 0; m < MW; ++m)
            Wf[(ks + WD - 1) % WD][m] = *reinterpret_cast<const v4i*>(wpk + ((size_t)m * KS + ks + WD - 1) * 1024 + wlane);
        }
        const uint32_t ka = trb + (uint32_t)ks * (32 * FS_PITCH);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          if (n < nmine) {
            const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(fs_lds + ka + n * 32));
            const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(fs_lds + ka + n * 32 + 8 * FS_PITCH));
            const v4i av = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int m = 0; m < MW; ++m)
              acc[n][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, Wf[ks % WD][m], ks == 0 ? zero16 : acc[n][m], 0, 0, 0);
          }
        }
      }
      if constexpr (mp == MP - 1) PLHIP_FS_STAMP(5);
      // the next pass's first operands arrive under this pass's epilogue (every slot of the ring is free by now)
      if constexpr (mp + 1 < MP) pass_operands(integral_constant<int, mp + 1>{});
      // ---------------------------------------------------------------- epilogue
      // accumulator register r of n tile n: slot 32 (n0 + n) + 8 (r >> 2) + 4 h + (r & 3); lane (c, h) owns channel 32 (mt0 + m) + c
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const int mch = (mt0 + m) * 32 + c;
        const float sc = psc[mp & 1][m], bi = pbi[mp & 1][m];
        const size_t obase = ((size_t)b * M + mch) * plane + (size_t)tr0 * W;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          if (n >= nmine) continue;
          const int px0 = (n0 + n) * 32;
          if (OUT == OUT_I8) {
            const float s2 = sc + sc, b2 = bi + bi;
            uint32_t edw[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              int v[4] = {acc[n][m][4 * gq], acc[n][m][4 * gq + 1], acc[n][m][4 * gq + 2], acc[n][m][4 * gq + 3]};
              edw[gq] = PWNN ? requant4_nn_rtz(v, s2, b2, hi2, a.ones) : dw_requant4<ACT_LEAKY>(v, s2, b2, leak, -254.f, 254.f);
            }
            // half exchange: every lane gets 16 consecutive slots of its channel (h = 0: px0 + 0..15, h = 1: px0 + 16..31)
            auto s02 = __builtin_amdgcn_permlane32_swap(edw[0], edw[2], false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(edw[1], edw[3], false, false);
            const int px = px0 + 16 * h;
            int8_t* const yo = reinterpret_cast<int8_t*>(g.y) + obase;
            if constexpr (RP == W) {
              const v4i v = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
              if (px < vpx) *reinterpret_cast<v4i*>(yo + px) = v;  // 16-byte aligned
            } else {  // one 14-pixel row: 8 + 4 + 2 bytes at a 2-byte aligned address
              const int row = px >> 4;
              if (row < vrows) {
                int8_t* const yr = yo + row * W;
                const v2i v01 = {(int)s02[0], (int)s02[1]};
                __builtin_memcpy(yr, &v01, 8);
                const uint32_t v2 = s13[0];
                __builtin_memcpy(yr + 8, &v2, 4);
                const uint16_t v3 = (uint16_t)s13[1];
                __builtin_memcpy(yr + 12, &v3, 2);
              }
            }
          } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int px = px0 + 8 * gq + 4 * h;
              int nval = 4;       // values of this quad that exist
              size_t opx = px;    // their offset in the tile's output
              if constexpr (RP == W) {
                if (px >= vpx) continue;
              } else {
                const int row = px >> 4, col = px & 15;
                if (row >= vrows) continue;
                nval = col == 12 ? 2 : 4;
                opx = (size_t)row * W + col;
              }
              if (OUT == OUT_F32) {
                float f[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float y = __fmaf_rn((float)acc[n][m][4 * gq + e], sc, bi);
                  if (g.act == ACT_LEAKY) y = y > 0.f ? y : g.alpha * y;
                  f[e] = fminf(fmaxf(y, flo), fcap);
                }
                float* const yf = reinterpret_cast<float*>(g.y) + obase + opx;
                if (nval == 4) {
                  const v4f v = {f[0], f[1], f[2], f[3]};
                  __builtin_memcpy(yf, &v, 16);
                } else {
                  const v2f v = {f[0], f[1]};
                  __builtin_memcpy(yf, &v, 8);
                }
              } else {
                int* const yi = reinterpret_cast<int*>(g.y) + obase + opx;
                if (nval == 4) {
                  const v4i v = {acc[n][m][4 * gq], acc[n][m][4 * gq + 1], acc[n][m][4 * gq + 2], acc[n][m][4 * gq + 3]};
                  __builtin_memcpy(yi, &v, 16);
                } else {
                  const v2i v = {acc[n][m][4 * gq], acc[n][m][4 * gq + 1]};
                  __builtin_memcpy(yi, &v, 8);
                }
              }
            }
          }
        }
      }
      self(self, integral_constant<int, mp + 1>{});
    }
  };
  pass(pass, integral_constant<int, 0>{});
  PLHIP_FS_STAMP(6);
  if (diag && lane == 0) g_fs_stamps[((size_t)vb * 4 + wave) * FS_STAMP_SLOTS + 7] = __builtin_amdgcn_s_memrealtime();
}

// shapes of the streaming kernel: (W, K, M) = (112, 32, 64), (56, 128, 128), (28, 256, 256): MobileNetV1's stride-1 pairs on
// the large planes; 3x3, stride 1, dilation 1, pad 1, square planes
bool fused_stream_supported(const FusedArgs& a) {
  if (!(a.h == a.w && a.oh == a.ow && a.pt == 1 && a.pl == 1 && (a.stride == 1 || a.stride == 2) && a.h == a.oh * a.stride)) return false;
  if (a.n < 1 || (long)a.n * a.C * a.h * a.w >= ((long)1 << 31) - 65536 || (long)a.n * a.pw.M * a.oh * a.ow >= ((long)1 << 31)) return false;
  if (a.stride == 2)
    return (a.ow == 56 && a.C == 64 && a.pw.M == 128) || (a.ow == 28 && a.C == 128 && a.pw.M == 256) ||
           (a.ow == 14 && a.C == 256 && a.pw.M == 512);
  return (a.w == 112 && a.C == 32 && a.pw.M == 64) || (a.w == 56 && a.C == 128 && a.pw.M == 128) || (a.w == 28 && a.C == 256 && a.pw.M == 256);
}

template <int W, int K, int M, int TP, int RS, int PD, int S, int MP, int OUT>
static void launch_stream_t(FusedArgs a, hipStream_t s) {
  constexpr int TR = W == 14 ? 7 : TP / W;
  a.pw.NT = (a.oh + TR - 1) / TR;  // tiles per image
  a.tiles = a.n * a.pw.NT;
  const unsigned blocks = (unsigned)((a.tiles + 7) / 8 * 8);
  // image, depthwise parameters, sink of the idle lanes (256 -> 256 @28 has none: its 80 KiB are exactly half a CU's LDS, and
  // 512 bytes more made it one block per CU: 28.3 -> 34.0 us)
  const size_t lds = (size_t)K * fs_pitch(TP) + (size_t)K * 32 + (M / MP < 256 ? 512 : 0);
  const bool dwnn = a.dw_act == ACT_RELU || a.dw_act == ACT_RELU6;
  const bool pwnn = OUT == OUT_I8 && (a.pw.act == ACT_RELU || a.pw.act == ACT_RELU6);
#define PLHIP_FS_LAUNCH(DN, PN)                                                                                  \
  do {                                                                                                           \
    auto kfn = fused_dwpw_stream_kernel<W, K, M, TP, RS, PD, S, MP, OUT, DN, PN>;                                               \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, s, a);                                                 \
  } while (0)
  if (dwnn && pwnn) PLHIP_FS_LAUNCH(true, true);
  else if (dwnn) PLHIP_FS_LAUNCH(true, false);
  else if (pwnn) PLHIP_FS_LAUNCH(false, true);
  else PLHIP_FS_LAUNCH(false, false);
#undef PLHIP_FS_LAUNCH
}

template <int W, int K, int M, int TP, int RS, int PD, int S = 1, int MP = 1>
static void launch_stream_o(const FusedArgs& a, int out, hipStream_t s) {
  if (out == OUT_I32) launch_stream_t<W, K, M, TP, RS, PD, S, MP, OUT_I32>(a, s);
  else if (out == OUT_F32) launch_stream_t<W, K, M, TP, RS, PD, S, MP, OUT_F32>(a, s);
  else launch_stream_t<W, K, M, TP, RS, PD, S, MP, OUT_I8>(a, s);
}

void launch_fused_stream(const FusedArgs& a, int out, hipStream_t s) {
  // 112-wide: 4-row tiles of 448 pixels (2-row tiles fetched and cut every input row twice: 61 us, the two kernels 56)
  if (a.stride == 2 && a.ow == 14) launch_stream_o<14, 256, 512, 128, 7, 2, 2, 2>(a, out, s);  // half images, M in two passes
  else if (a.stride == 2 && a.ow == 56) launch_stream_o<56, 64, 128, 224, 4, 2, 2>(a, out, s);
  else if (a.stride == 2) launch_stream_o<28, 128, 256, 224, 4, 2, 2>(a, out, s);
  else if (a.w == 112) launch_stream_o<112, 32, 64, 448, 4, 2>(a, out, s);
  else if (a.w == 56) launch_stream_o<56, 128, 128, 224, 4, 2>(a, out, s);
  else launch_stream_o<28, 256, 256, 224, 4, 2>(a, out, s);
}

}  // namespace plhip
