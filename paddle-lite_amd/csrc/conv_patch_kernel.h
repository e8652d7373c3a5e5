// conv_patch_kernel.h — the dense 3x3 stride-1 convolution as an implicit GEMM on input PATCHES (kernel template +
// launchers; included by conv_patch_i8.hip and conv_patch_stream.hip, described in conv_patch_i8.hip).
#pragma once
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "gemm_tr_common.h"
#include "dw_common.h"

namespace plhip {

constexpr int PATCH_NTW = 7;           // 32-pixel n tiles per wave and tile
constexpr int PATCH_SP = 144;          // staging row pitch of the int8 epilogue: 128 bytes + 16
constexpr int PATCH_STAMP_SLOTS = 32;

template <int I, int N, class F>
__device__ __forceinline__ void patch_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    patch_static_for<I + 1, N>(std::forward<F>(f));
  }
}

#define PLHIP_PATCH_STAMP(i)                                               \
  do {                                                                     \
    if (diag && lane == 0) lstamp[i] = __builtin_amdgcn_s_memtime();       \
  } while (0)

// NH halves of 4 waves per block, WMH x WNH waves per half: a half owns its own stream of pixel tiles.  NH = 1: 4-wave
// blocks, TWO per CU, each with its own ring and its own barriers: the blocks drift apart, so that one's epilogue (VALU,
// stores) runs beside the other's MFMAs (with one 8-wave block the per-slab barrier keeps the SIMD partners in lock step:
// both multiply, then both requantise: profiles/r03_patch_timeline_v1_c2.txt).  NH = 2: one 8-wave block per CU whose slab
// pair shares the weight fragments of its (chunk, shift) through the ring.  NPW DMA pieces per wave and slot; NSLOT ring
// slots; STAT: the whole K of the weights stays in registers (C = 64), else they travel through the ring.
// KT: 3 = stride 1: per 32-channel chunk 3 slabs (column shifts) of 3 tap rows.  2 = a 3x3 STRIDE-2 conv over the 4 phase
// planes of every channel (conv_patch_i8.hip; weights through the ring only): the "chunks" of the padded copy are (group of 32
// channels, phase), and a group is walked in 6 steps of 2 / 1 / 2 / 1 / 2 / 1 tap rows (PATCH_S2_* below): the 9 taps once.
// stride 2, step ls of a group: phase plane (a, b) = (row, column parity) of the padded input, its column shift s' and tap
// rows r' < ROWS <-> taps (2 r' + a, 2 s' + b); first weight fragment of the step among the group's 9
//   ls      0      1      2      3      4      5
//   phase  (0,0)  (1,0)  (0,0)  (1,0)  (0,1)  (1,1)
//   s'      0      0      1      1      0      0
//   rows    2      1      2      1      2      1        (2-row and 1-row steps alternate: every pair of steps is 21 MFMAs)
__host__ __device__ constexpr int patch_s2_phase(int ls) { return (0x312020 >> (4 * ls)) & 15; }  // 2 a + b
__host__ __device__ constexpr int patch_s2_shift(int ls) { return (ls == 2 || ls == 3) ? 1 : 0; }
__host__ __device__ constexpr int patch_s2_rows(int ls) { return (ls & 1) ? 1 : 2; }
__host__ __device__ constexpr int patch_s2_wfrag(int ls) { return (0x865320 >> (4 * ls)) & 15; }

// DMA pieces per wave of the slot of step ls, and their sum over steps [ls0, ls1)
template <int KT, int NPW>
__host__ __device__ constexpr int patch_npws(int ls) { return KT == 3 ? NPW : NPW - ((ls % 6) & 1); }
template <int KT, int NPW>
__host__ __device__ constexpr int patch_npws_sum(int ls0, int ls1) {
  int n = 0;
  for (int l = ls0; l < ls1; ++l) n += patch_npws<KT, NPW>(l);
  return n;
}

template <int NH, int WMH, int WNH, int NPW, int NSLOT, bool STAT, int OUT, bool NONNEG, int KT = 3>
__global__ __launch_bounds__(256 * NH, 2) void conv_patch_i8_kernel(PatchArgs a) {
  static_assert(KT == 3 || (KT == 2 && !STAT && NH == 2 && WMH == 4 && (NSLOT - 1) % 2 == 0), "taps");
  constexpr int SPG = KT == 3 ? 3 : 6;  // steps per chunk (stride 2: per group of 4 phase chunks)
  constexpr int NTW = PATCH_NTW, NTH = WNH * NTW * 32, D = NSLOT - 1, NW = 4 * NH;
  constexpr int WPIECES = STAT ? 0 : KT * WMH;  // 1-KiB weight pieces of a slot
  static_assert(WMH * WNH == 4 && NSLOT >= 3 && NSLOT <= 4 && (NH == 1 || NH == 2), "layout");
  PLHIP_PRELOAD(a.xp); PLHIP_PRELOAD(a.wp); PLHIP_PRELOAD(a.y); PLHIP_PRELOAD(a.scale); PLHIP_PRELOAD(a.bias);
  PLHIP_PRELOAD(a.C); PLHIP_PRELOAD(a.M); PLHIP_PRELOAD(a.OH); PLHIP_PRELOAD(a.OW); PLHIP_PRELOAD(a.PWp); PLHIP_PRELOAD(a.PLANE);
  PLHIP_PRELOAD(a.NCH); PLHIP_PRELOAD(a.pitch); PLHIP_PRELOAD(a.pps); PLHIP_PRELOAD(a.TPI); PLHIP_PRELOAD(a.T); PLHIP_PRELOAD(a.T8);
  PLHIP_PRELOAD(a.MB); PLHIP_PRELOAD(a.NQ); PLHIP_PRELOAD(a.rounds); PLHIP_PRELOAD(a.HWY); PLHIP_PRELOAD(a.y_bstride);
  PLHIP_PRELOAD(a.act); PLHIP_PRELOAD(a.alpha); PLHIP_PRELOAD(a.pw_m); PLHIP_PRELOAD(a.pw_s); PLHIP_PRELOAD(a.tpi_m);
  PLHIP_PRELOAD(a.tpi_s); PLHIP_PRELOAD(a.pitch_m); PLHIP_PRELOAD(a.pitch_s); PLHIP_PRELOAD(a.dbg); PLHIP_PRELOAD(a.res);
  PLHIP_PRELOAD(a.glob); PLHIP_PRELOAD(a.nimg); PLHIP_PRELOAD(a.IMGP); PLHIP_PRELOAD(a.imgp_m); PLHIP_PRELOAD(a.imgp_s); PLHIP_PRELOAD(a.hwy_m); PLHIP_PRELOAD(a.hwy_s);
  PLHIP_PRELOAD(a.y2); PLHIP_PRELOAD(a.inv_scale2); PLHIP_PRELOAD(a.res_relu); PLHIP_PRELOAD(a.stamps); PLHIP_PRELOAD(a.delay);
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];        // NSLOT x [half 0: 32 x pitch][half 1][weights]
  __shared__ __attribute__((aligned(16))) uint8_t stg_all[NW * 32 * PATCH_SP];  // int8 epilogue staging, one image per wave
  __shared__ unsigned long long stamp_all[NW * PATCH_STAMP_SLOTS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = NH == 2 ? wave >> 2 : 0, wq = wave & 3;
  const int wm = wq % WMH, wn = wq / WMH;  // wave-uniform
  const int c = lane & 31, h = lane >> 5;
  const bool diag = (a.dbg & 32) != 0;
  unsigned long long* lstamp = stamp_all + wave * PATCH_STAMP_SLOTS;
  if (diag && lane == 0) {
    lstamp[0] = __builtin_amdgcn_s_memrealtime();
    lstamp[1] = __builtin_amdgcn_s_memtime();
    lstamp[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }

  // ---- block -> (XCD, M block, n-block slot); stream of a half = 2 nq + half inside the XCD's contiguous tile range
  const int bx = blockIdx.x & 7, bq = blockIdx.x >> 3;
  const int mb = bq % a.MB, nq = bq / a.MB;
  // XCD x owns the CONTIGUOUS tiles [x T8, (x + 1) T8) for the whole kernel (its images' padded copy was written by the same
  // XCD, pad_rows8_i8_kernel: the reads hit the local L2 instead of crossing the fabric; neighbouring tiles share their halo
  // rows there too).  Inside it: stream li of SX = NQ NH takes the tiles x T8 + k SX + li, k = 0, 1, ..
  const int SX = a.NQ * NH, li0 = nq * NH;
  auto tile_of = [&](int k, int li, bool& live) __attribute__((always_inline)) -> int {
    const int j = k * SX + li;
    const int t = bx * a.T8 + j;
    live = j < a.T8 && t < a.T;
    return live ? t : a.T - 1;  // idle streams re-read the last tile (every step issues the same number of DMA pieces)
  };
  const int MT32 = (a.M + 31) >> 5;
  const int mt = mb * WMH + wm;                        // my 32-row m tile
  const int mtc = mt < MT32 ? mt : MT32 - 1;           // tiles past M: any packed tile (their rows are never stored)
  const int mrow = mt * 32 + c;
  const int SLAB = 32 * a.pitch;
  const int SLOTB = NH * SLAB + WPIECES * 1024;
  const int NCH = KT == 3 ? a.NCH : a.NCH >> 2;  // K loop: chunks (stride 2: groups of 32 channels = 4 phase chunks)

  // ---- this lane's scale / bias: the two oldest loads of the wave, inline asm like every load here (the compiler would
  // guard an ordinary load with vmcnt(0) at its first use, the epilogue, and drain the DMA pipeline there); every counted
  // wait below covers them.  Always two loads (no bias: the scale once more) so that the counts are constants.
  float sc = 1.f, bi = 0.f;
  if constexpr (OUT != OUT_I32) {
    const uint32_t mo = (uint32_t)(mrow < a.M ? mrow : a.M - 1) * 4u;
    const float* sp = a.scale;
    const float* bp = a.bias ? a.bias : a.scale;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(sc) : "v"(mo), "s"(sp) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=v"(bi) : "v"(mo), "s"(bp) : "memory");
  }

  // ---- my DMA pieces of every slot.  Activation pieces first: piece i = wave + NW j (j < NPA) is 1 KiB i of the slabs
  // [half 0: pps KiB][half 1: pps KiB]: lane -> (channel row, 16 bytes of the row); i >= NH pps: piece i - NH pps once
  // more (same bytes to the same place: the per-step issue count stays a constant).  Then the weight pieces (ring mode):
  // i = wave + NW (j - NPA) -> m tile i / 3 of the block, tap row i % 3.
  constexpr int NPA = STAT ? NPW : NPW - (WPIECES + NW - 1) / NW;
  // (the 2 x 2 layout with its 72 weight registers has no room to keep the 5 per-lane offsets: it recomputes them per issue)
  constexpr bool PVO_KEPT = !(STAT && WNH == 2);
  uint32_t pvo[PVO_KEPT ? NPA : 1];   // per-lane source offset from the (tile, chunk, shift) base of the piece's half
  int pai[NPA];                       // piece index i (wave-uniform)
  auto piece_offset = [&](int i, uint32_t ln) __attribute__((always_inline)) -> uint32_t {
    const uint32_t pos = (uint32_t)(i >= a.pps ? i - a.pps : i) * 1024u + ln * 16u;
    const uint32_t row = fastdiv_u31(pos, a.pitch_m, a.pitch_s);
    return row * (uint32_t)a.PLANE + (pos - row * (uint32_t)a.pitch);
  };
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    int i = wave + NW * j;
    if (i >= NH * a.pps) i -= NH * a.pps;
    if (i >= NH * a.pps) i = 0;
    pai[j] = i;
    if constexpr (PVO_KEPT) pvo[j] = piece_offset(i, (uint32_t)lane);
  }
  uint32_t pwoff[STAT ? 1 : NPW - NPA];  // weight piece: byte offset of its (m tile, r) fragment at (chunk 0, s 0); wave-uniform
  int pwi[STAT ? 1 : NPW - NPA];
  if constexpr (!STAT) {
#pragma unroll
    for (int j = 0; j < NPW - NPA; ++j) {
      int iw = wave + NW * j;
      if (iw >= WPIECES) iw -= NW;
      if (iw >= WPIECES) iw = 0;
      const int mtl = iw / KT, r = iw - mtl * KT;
      int mtw = mb * WMH + mtl;
      mtw = mtw < MT32 ? mtw : MT32 - 1;
      pwi[j] = iw;
      pwoff[j] = (uint32_t)((mtw * NCH * 9 + r) * 1024);
    }
  }
  // stride 2, the 1-row steps: 2 pps activation pieces + WMH weight pieces fit the NPA activation slots of the 8 waves: the slot
  // i = wave + NW (NPA - 1) >= 2 pps is weight piece i - 2 pps (m tile i - 2 pps, the step's one tap row) there
  int w1i = -1;          // wave-uniform
  uint32_t pw1off = 0;
  if constexpr (KT == 2) {
    const int i1 = wave + NW * (NPA - 1) - NH * a.pps;
    if (i1 >= 0 && i1 < WMH) {
      w1i = i1;
      int mtw = mb * WMH + i1;
      mtw = mtw < MT32 ? mtw : MT32 - 1;
      pw1off = (uint32_t)(mtw * NCH * 9 * 1024);
    }
  }
  const uint32_t lane16 = (uint32_t)lane * 16u;

  // ---- weights in registers (STAT): [chunk][s][r] fragments of my m tile, loaded once (inline asm: invisible to the
  // compiler's wait-count pass; the counted waits of the first round cover them, see the prologue)
  v4i w[STAT ? 18 : 1];
  const uint8_t* wbase = reinterpret_cast<const uint8_t*>(a.wp) + (size_t)mtc * (18 * 1024);  // wave-uniform
  auto load_w = [&](auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    const uint32_t vo = (uint32_t)lane * 16u + (uint32_t)(i >> 2) * 4096u;
    if constexpr ((i & 3) == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[i]) : "v"(vo), "s"(wbase) : "memory");
    else if constexpr ((i & 3) == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[i]) : "v"(vo), "s"(wbase) : "memory");
    else if constexpr ((i & 3) == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(w[i]) : "v"(vo), "s"(wbase) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(w[i]) : "v"(vo), "s"(wbase) : "memory");
  };

  // ---- DMA issue of the slab pair (cursor tile bases cb[0 / 1], chunk ic, shift is) into ring slot `slot`
  const uint8_t* cb[NH];  // base of the cursor's tile of half 0 / 1 in the padded copy (channel 0, shift 0); wave-uniform
  auto cursor_tiles = [&](int ik) __attribute__((always_inline)) {
    ik = ik < a.rounds ? ik : a.rounds - 1;  // past the end: a harmless re-fetch (keeps the per-step issue count constant)
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      bool lv;
      const int t = tile_of(ik, li0 + hf, lv);
      const uint32_t b = fastdiv_u31((uint32_t)t, a.tpi_m, a.tpi_s);
      const int p0 = (t - (int)b * a.TPI) * NTH;
      cb[hf] = reinterpret_cast<const uint8_t*>(a.xp) + (size_t)b * a.C * (uint32_t)a.PLANE + p0;
    }
  };
  // one DMA piece j of the slot (chunk ic, shift is) -> ring slot `slot`; j < NPA: activations, then weights
  // (stride 2: `is` is the step index inside the group, a compile-time constant at every call site)
  auto issue_piece = [&](auto j_c, const uint8_t* const (&tb)[NH], int ic, int is, int slot) __attribute__((always_inline)) {
    constexpr int j = decltype(j_c)::value;
    uint8_t* sb = ring + slot * SLOTB;
    const int chunk = KT == 3 ? ic : ic * 4 + patch_s2_phase(is);
    const int shift = KT == 3 ? is : patch_s2_shift(is);
    const int wfrag = KT == 3 ? ic * 9 + is * 3 : ic * 9 + patch_s2_wfrag(is);
    const uint8_t* wsrc = reinterpret_cast<const uint8_t*>(a.wp) + (size_t)wfrag * 1024;
    if constexpr (j < NPA) {
      const size_t coff = (size_t)(chunk * 32) * (uint32_t)a.PLANE + shift;
      const uint8_t* sbase = (NH == 2 && pai[j] >= a.pps ? tb[NH - 1] : tb[0]) + coff;  // wave-uniform
      uint32_t vo;
      if constexpr (PVO_KEPT) {
        vo = pvo[j];
      } else {
        uint32_t ln = (uint32_t)lane;
        asm volatile("" : "+v"(ln));  // (opaque: not hoisted back out of the loop)
        vo = piece_offset(pai[j], ln);
      }
      const uint8_t* src = sbase + vo;
      uint8_t* dst = sb + pai[j] * 1024;
      if constexpr (KT == 2 && j == NPA - 1) {
        if (patch_s2_rows(is) == 1 && w1i >= 0) {  // wave-uniform; selects, not a branch
          src = wsrc + pw1off + lane16;
          dst = sb + NH * SLAB + w1i * (KT * 1024);
        }
      }
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc + pwoff[j - NPA] + lane16), (lds_ptr_t)(sb + NH * SLAB + pwi[j - NPA] * 1024), 16, 0, 0);
    }
  };
  // pieces per wave of the slot of step ls (stride 2: the 1-row steps have no piece NPW - 1)
  auto issue = [&](int ic, auto is_c, int slot) __attribute__((always_inline)) {
    constexpr int is = decltype(is_c)::value;
    const uint8_t* tb[NH];
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) tb[hf] = cb[hf];
    patch_static_for<0, patch_npws<KT, NPW>(is)>([&](auto j_c) __attribute__((always_inline)) { issue_piece(j_c, tb, ic, is, slot); });
  };

  // ---- fragment addresses: lane 2q'+p of a 16-lane group -> channel row q', 8-byte sub-chunk p; group parity -> 16-pixel
  // chunk; k half h -> channels 16h .. 16h+15 (lo: +0..7, hi: +8..15)
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
  const uint32_t fa = ring_addr + half * SLAB + (16 * h + ((lane & 15) >> 1)) * a.pitch + wn * (NTW * 32) + ((lane >> 4) & 1) * 16 + (lane & 1) * 8;
  const uint32_t wa = ring_addr + NH * SLAB + (wm * KT) * 1024 + lane * 16;
  const uint32_t pitch8 = 8u * (uint32_t)a.pitch;

  v16i acc[NTW];
  v2i flo[4], fhi[4];  // fragment ring: 3 reads ahead
  v4i wr[KT];          // weight fragments of the step (ring mode)

  // ---- epilogue constants
  const float hi2 = a.act == ACT_RELU6 ? fminf(a.alpha + a.alpha, 254.f) : 254.f;
  const float lo2 = NONNEG ? 0.f : -254.f;
  const float leak = a.act == ACT_LEAKY ? a.alpha : 1.f;  // int8, !NONNEG: none = leaky with slope 1
  uint8_t* stg = stg_all + wave * (32 * PATCH_SP);

  int nepi = 0;  // epilogues done (diagnostic stamps of the first)
  // destination offset (inside a channel plane) of p-space pixel p: rows are OW of PWp wide; p past the image -> its end.
  // Branch free (it runs on the scalar unit for wave-uniform p: a branch there is a pipeline bubble per piece): PWp >= 16,
  // so the host's magic is never the power-of-two marker 0 (launch_conv_patch uses the general form for powers of two too)
  // Global mode (a.glob: 7-wide planes, whose 56..72 pixels are less than a tile): the padded copy is CHANNEL-major
  // ([c][image][PH][PWp]), so for one channel the images follow each other and p runs over all of them, IMGP pixels each;
  // the value returned is then a VIRTUAL compact index img * HWY + oh * OW + ow (still monotone in p); virt_to_y maps it
  // to the output element (images are M * HWY apart for one channel).
  auto dst_of = [&](auto glob_c, uint32_t p, int& valid, int span) __attribute__((always_inline)) -> int {
    int vbase = 0, img_ok = -1;
    if constexpr (decltype(glob_c)::value) {
      const uint32_t img = __umulhi(p, a.imgp_m) >> a.imgp_s;
      p -= img * (uint32_t)a.IMGP;
      img_ok = ((int)img - a.nimg) >> 31;  // all ones: the image exists (the last tile runs past the last one)
      vbase = ((int)img < a.nimg ? (int)img : a.nimg) * a.HWY;
    }
    const uint32_t oh = __umulhi(p, a.pw_m) >> a.pw_s;
    const int ow0 = (int)(p - oh * (uint32_t)a.PWp);
    const int inside = ((int)oh - a.OH) >> 31;  // all ones: the row exists
    int v = a.OW - ow0;
    v = v < 0 ? 0 : v;
    v = v > span ? span : v;
    const int owc = ow0 < a.OW ? ow0 : a.OW;
    const int ohc = (int)oh < a.OH ? (int)oh : a.OH;
    valid = v & inside & img_ok;
    return vbase + ((ohc * a.OW + (owc & inside)) & img_ok);
  };
  // element offset of virtual compact index v of channel row m inside y (global mode: per lane, v / HWY by magic)
  auto virt_to_y = [&](auto glob_c, int v, int m, int& rem) __attribute__((always_inline)) -> size_t {
    if constexpr (!decltype(glob_c)::value) {
      rem = 0;
      return (size_t)m * (uint32_t)a.HWY + v;
    }
    const uint32_t img = __umulhi((uint32_t)v, a.hwy_m) >> a.hwy_s;
    rem = v - (int)img * a.HWY;
    return ((size_t)img * a.M + m) * (uint32_t)a.HWY + rem;
  };

  // GLOB (global mode) is a compile-time property of the epilogue: a kernel-uniform test per geometry call cost the other
  // layers 5-10 % (config #2 20.3 -> 22.5 us, measured on one box against the build before)
  auto epilogue = [&](auto glob_c, int b, int p0, int mt) __attribute__((always_inline)) {  // (mt: an opaque copy, see the call)
    constexpr bool GLOB = decltype(glob_c)::value;
    const int pw0 = p0 + wn * (NTW * 32);  // first pixel of my n tiles (wave-uniform: so is every dst_of below -> scalar unit)
    const int mrow = mt * 32 + c;
    const float bi_ = a.bias ? bi : 0.f;
    const float s2 = sc + sc, b2 = bi_ + bi_;
    if constexpr (OUT == OUT_I8) {
      // A lane holds 16 consecutive pixels of its channel per n tile (after the half swap) = two 8-pixel pieces, each inside
      // one padded row; piece k (k = 2h + e) keeps its first valid_k bytes, which belong at d_k in the dense output row.  The
      // wave writes the pieces into its staging image IN PIXEL ORDER, all 8 bytes each, at byte d_k - d_a (unaligned
      // ds_write_b64): the dropped tail of a piece is overwritten by the next one, so the image comes out compacted.  Then
      // it is read back as 16-byte pieces of whole channel rows: a store instruction writes 8 rows x up to 128 contiguous bytes.
      int8_t* yb = reinterpret_cast<int8_t*>(a.y) + (size_t)b * a.y_bstride;
      const uint32_t stg_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)stg;
      const uint32_t wrow = stg_addr + (uint32_t)c * PATCH_SP;                               // my channel row (staging writes)
      const uint32_t rrow = stg_addr + (uint32_t)(lane >> 3) * PATCH_SP + (lane & 7) * 16;   // read-back: row lane >> 3 (+ 8 it)
      patch_static_for<0, 2>([&](auto g_c) __attribute__((always_inline)) {
        constexpr int gi = decltype(g_c)::value;
        constexpr int nt0 = gi * 4, cnt = gi == 0 ? 4 : NTW - 4;
        int dummy;
        const int d_a = dst_of(glob_c, (uint32_t)(pw0 + 32 * nt0), dummy, 0);
        const int d_b = dst_of(glob_c, (uint32_t)(pw0 + 32 * (nt0 + cnt)), dummy, 0);
        patch_static_for<0, cnt>([&](auto t_c) __attribute__((always_inline)) {
          constexpr int t = decltype(t_c)::value;
          const v4i ch = NONNEG ? tr_requant_chunk<ACT_RELU>(acc[nt0 + t], s2, b2, leak, lo2, hi2)
                                : tr_requant_chunk<ACT_LEAKY>(acc[nt0 + t], s2, b2, leak, lo2, hi2);
          const v2i pc0 = {ch[0], ch[1]}, pc1 = {ch[2], ch[3]};
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // pixel order; lanes of half k >> 1 hold piece k
            int valid;
            const int d = dst_of(glob_c, (uint32_t)(pw0 + 32 * (nt0 + t) + 8 * k), valid, 8);  // scalar
            // a dropped piece goes to the spare 8 bytes at the end of the row (no branch: a pipeline bubble per piece)
            const uint32_t wa_ = wrow + (uint32_t)(valid > 0 ? d - d_a : PATCH_SP - 8);
            const unsigned long long m = (k >> 1) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
            if (k & 1) asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(wa_), "v"(pc1), "s"(m) : "memory");
            else asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(wa_), "v"(pc0), "s"(m) : "memory");
          }
        });
        const int len = d_b - d_a;  // <= 128
        if (diag && nepi == 0 && lane == 0) lstamp[18 + 2 * gi] = __builtin_amdgcn_s_memtime();  // group staged
        v4i rv[4];
        asm volatile("ds_read_b128 %0, %1" : "=v"(rv[0]) : "v"(rrow) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rv[1]) : "v"(rrow), "n"(8 * PATCH_SP) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rv[2]) : "v"(rrow), "n"(16 * PATCH_SP) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rv[3]) : "v"(rrow), "n"(24 * PATCH_SP) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3])::"memory");
        // whole 16-byte pieces, then the row's tail (len % 16 bytes, the same for every row: wave-uniform decisions)
        const int off = (lane & 7) * 16, tail = len & 15, toff = len & ~15;
        if constexpr (GLOB) {  // a piece may end in the next image (HWY = 49 bytes per image and channel): byte-wise there
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int m = mt * 32 + it * 8 + (lane >> 3);
            const v4i v = rv[it];
            const int cnt_ = len - off < 16 ? len - off : 16;
            if (cnt_ > 0 && m < a.M) {
              int rem;
              int8_t* yp = reinterpret_cast<int8_t*>(a.y) + virt_to_y(glob_c, d_a + off, m, rem);
              const int n1 = a.HWY - rem < cnt_ ? a.HWY - rem : cnt_;  // bytes that stay in this image
              if (n1 == 16) {
                __builtin_memcpy(yp, &v, 16);
              } else {
                int8_t* yq = yp - rem + (size_t)a.M * (uint32_t)a.HWY;  // the same channel row of the next image
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                  const int8_t bv = (int8_t)((uint32_t)v[k >> 2] >> (8 * (k & 3)));
                  if (k < n1) yp[k] = bv;
                  else if (k < cnt_) yq[k - n1] = bv;
                }
              }
            }
          }
        } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = mt * 32 + it * 8 + (lane >> 3);
          int8_t* yp = yb + (size_t)m * (uint32_t)a.HWY + d_a + off;
          const v4i v = rv[it];
          if (off + 16 <= len && m < a.M) __builtin_memcpy(yp, &v, 16);  // possibly unaligned: fine for global memory
          if (tail) {                                                     // wave-uniform
            const bool mine = off == toff && m < a.M;
            int tb = 0;  // bytes of the tail already stored
            if (tail & 8) {
              const v2i v8 = {v[0], v[1]};
              if (mine) __builtin_memcpy(yp, &v8, 8);
              tb = 8;
            }
            if (tail & 4) {
              const int dw = tb ? v[2] : v[0];
              if (mine) __builtin_memcpy(yp + tb, &dw, 4);
              tb += 4;
            }
            if (tail & 3) {
              const uint32_t dw = (uint32_t)(tb == 0 ? v[0] : (tb == 4 ? v[1] : (tb == 8 ? v[2] : v[3])));
              if (tail & 2) {
                const uint16_t hw16 = (uint16_t)dw;
                if (mine) __builtin_memcpy(yp + tb, &hw16, 2);
              }
              if (tail & 1) {
                const uint8_t b8 = (uint8_t)((tail & 2) ? dw >> 16 : dw);
                if (mine) yp[tb + (tail & 2)] = (int8_t)b8;
              }
            }
          }
        }
        }
        if (diag && nepi == 0 && lane == 0) lstamp[19 + 2 * gi] = __builtin_amdgcn_s_memtime();  // group's stores issued
      });
    } else {
      // 32-bit outputs: register group gq of n tile t = 4 consecutive pixels 32t + 8gq + 4h (inside one row: PWp % 4 == 0);
      // the geometry of both halves' groups is computed on the scalar unit, the lane picks its half's
      const float fcap = a.act == ACT_RELU6 ? a.alpha : __builtin_huge_valf();
      const float flo_ = (a.act == ACT_RELU || a.act == ACT_RELU6) ? 0.f : -__builtin_huge_valf();
      patch_static_for<0, NTW>([&](auto t_c) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          int v0_, v1_;
          const int d0_ = dst_of(glob_c, (uint32_t)(pw0 + 32 * t + 8 * gq), v0_, 4);
          const int d1_ = dst_of(glob_c, (uint32_t)(pw0 + 32 * t + 8 * gq + 4), v1_, 4);
          if (v0_ == 0 && v1_ == 0) continue;  // wave-uniform
          const int valid = h ? v1_ : v0_, d = h ? d1_ : d0_;
          if (valid == 0 || mrow >= a.M) continue;
          int rem_;
          const size_t yoff = (size_t)b * a.y_bstride + virt_to_y(glob_c, d, mrow, rem_);  // (a 4-pixel group never leaves its image)
          if constexpr (OUT == OUT_I32) {
            int* yp = reinterpret_cast<int*>(a.y) + yoff;
            if (valid == 4) {
              const v4i v = {acc[t][4 * gq], acc[t][4 * gq + 1], acc[t][4 * gq + 2], acc[t][4 * gq + 3]};
              __builtin_memcpy(yp, &v, 16);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (e < valid) yp[e] = acc[t][4 * gq + e];
            }
          } else {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float yv = __fmaf_rn((float)acc[t][4 * gq + e], sc, bi_);
              if (a.act == ACT_LEAKY) yv = yv > 0.f ? yv : a.alpha * yv;  // kernel-uniform
              f[e] = fminf(fmaxf(yv, flo_), fcap);
            }
            if (a.res) {  // fused residual add (+ relu): kernel-uniform
              const float* rp = a.res + yoff;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (e < valid) f[e] = f[e] + rp[e];
                if (a.res_relu) f[e] = f[e] > 0.f ? f[e] : 0.f;
              }
            }
            if (a.y) {
              float* yp = reinterpret_cast<float*>(a.y) + yoff;
              if (valid == 4) {
                const v4f v = {f[0], f[1], f[2], f[3]};
                __builtin_memcpy(yp, &v, 16);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (e < valid) yp[e] = f[e];
              }
            }
            if (a.y2) {  // fused calib fp32 -> int8 of the value just produced (type_trans.cc:45,183-184)
              int8_t* qp = a.y2 + yoff;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (e < valid) qp[e] = (int8_t)round_sat_i8(a.inv_scale2 * f[e]);
            }
          }
        }
      });
    }
  };

  // ---- one step = one slab pair: 3 tap rows x NTW n tiles of MFMAs from my half's slab.
  // CH >= 0: chunk known at compile time (STAT); SS: column shift s; FIRST / LAST: first / last slab of a tile.
  int slot = 0;                  // ring slot of the current step
  int nstep = 0;                 // steps done (diagnostic stamps of the first six)
  int iqk = 0, iqc = 0, iqs = 0; // issue cursor: the slab pair D steps ahead
  auto advance_cursor = [&]() __attribute__((always_inline)) {
    if (++iqs == SPG) {
      iqs = 0;
      if (++iqc == NCH) {
        iqc = 0;
        ++iqk;
        cursor_tiles(iqk);
      }
    }
  };
  // FIRST: 0 = not the first slab of a tile, 1 = the first (the accumulators start from the constant 0 operand), 2 = run
  // time (`first`: the accumulators are zeroed)
  // R0: the first round of the register-resident weights: the prologue interleaves their loads with the first D slabs, the
  // wait of step q < D leaves everything behind fragment 3q + 2 in flight (see the prologue).
  auto step = [&](auto ch_c, auto ss_c, auto first_c, bool r0, bool first) __attribute__((always_inline)) {
    using std::integral_constant;
    constexpr int CH = decltype(ch_c)::value, SS = decltype(ss_c)::value, FIRST = decltype(first_c)::value;
    // my pieces of this slot have landed (counted: everything issued after them may still fly) ...
    constexpr int Q = CH * 3 + SS;
    constexpr int ROWS = KT == 3 ? 3 : patch_s2_rows(SS);  // tap rows of this step
    constexpr int LS = (SS + D) % SPG;                     // the step whose slot this one fills; stride 2: as many rows as this one
    constexpr int NPI = patch_npws<KT, NPW>(LS);                          // DMA pieces this step issues
    static_assert(KT == 3 || patch_s2_rows(LS) == ROWS, "lookahead");
    constexpr int REG = patch_npws_sum<KT, NPW>(SS + 1, SS + D);
    constexpr int WAIT0 = (STAT && Q < D) ? (D - 1 - Q) * (NPW + 3) + (18 - 3 * D) + Q * NPW : REG;
    static_assert(WAIT0 >= REG && WAIT0 < 64, "vmcnt");
    if constexpr (STAT && Q < D) {
      if (r0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT0) : "memory");  // block-uniform
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(REG) : "memory");
      asm volatile("" : "+v"(w[3 * Q]), "+v"(w[3 * Q + 1]), "+v"(w[3 * Q + 2]));  // no use of these fragments above the wait
      if constexpr (Q == 0 && OUT != OUT_I32) asm volatile("" : "+v"(sc), "+v"(bi));  // (the two oldest loads)
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(REG) : "memory");
      if constexpr (STAT) asm volatile("" : "+v"(w[3 * Q]), "+v"(w[3 * Q + 1]), "+v"(w[3 * Q + 2]));
    }
    __builtin_amdgcn_s_barrier();  // ... everyone's have, and nobody reads the previous step's slot any more
    if (diag && nstep < 6 && lane == 0) lstamp[5 + nstep] = __builtin_amdgcn_s_memtime();
    // The slot of the step before is free now: the DMA of the slot D steps ahead goes into it, ONE PIECE BEHIND EVERY FOURTH
    // MFMA.  (All pieces at the top of the step, from all waves at once, is a burst of 24-40 KiB into an address path that
    // takes ~58 B/clk: every wave sat ~700 cycles in the issue with the matrix pipe idle, and the time of the data movement
    // ADDED to the MFMA time instead of hiding under it: profiles/r03_patch_decompose.txt.)
    int islot = slot + D;
    islot = islot >= NSLOT ? islot - NSLOT : islot;
    const int qc = iqc, qs = KT == 3 ? iqs : LS;
    const uint8_t* qb[NH];  // (the cursor's tile bases BEFORE it advances: the last slot of a round still belongs to the old tile)
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) qb[hf] = cb[hf];
    advance_cursor();
    const uint32_t sb = (uint32_t)(slot * SLOTB);
    {  // (an idle stream, past the last tile, multiplies the clamped tile's bytes again: no branch around the loop)
      if (FIRST == 2 && first) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0;
      }
      uint32_t alo[ROWS], ahi[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        alo[r] = fa + sb + (uint32_t)(r * a.PWp);
        ahi[r] = alo[r] + pitch8;
      }
      if constexpr (!STAT) {
        const uint32_t wab = wa + sb;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wr[0]) : "v"(wab) : "memory");
        if constexpr (ROWS >= 2) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(wr[1]) : "v"(wab) : "memory");
        if constexpr (ROWS == 3) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wr[2]) : "v"(wab) : "memory");
      }
      constexpr int NM = ROWS * NTW;  // MFMA i <-> (tap row i / NTW, n tile i % NTW)
#define PLHIP_PATCH_READ(I_)                                                                                              \
  do {                                                                                                                    \
    constexpr int r_ = (I_) / NTW, t_ = (I_) % NTW, f_ = (I_) & 3;                                                        \
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(flo[f_]) : "v"(alo[r_]), "n"(t_ * 32) : "memory");           \
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(fhi[f_]) : "v"(ahi[r_]), "n"(t_ * 32) : "memory");           \
  } while (0)
      PLHIP_PATCH_READ(0);
      PLHIP_PATCH_READ(1);
      PLHIP_PATCH_READ(2);
      if constexpr (!STAT && ROWS == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2])::"memory");
      if constexpr (!STAT && ROWS == 2) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wr[0]), "+v"(wr[1])::"memory");
      if constexpr (!STAT && ROWS == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wr[0])::"memory");
      constexpr int IEV = NM >= 4 * NPI - 2 ? 4 : 2;  // a DMA piece behind every IEV-th MFMA
      patch_static_for<0, NM>([&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        constexpr int r = i / NTW, t = i % NTW, f = i & 3;
        constexpr int younger = (NM - 1 - i) < 2 ? 2 * (NM - 1 - i) : 4;
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(flo[f]), "+v"(fhi[f]) : "n"(younger) : "memory");
        const v4i av = {flo[f][0], flo[f][1], fhi[f][0], fhi[f][1]};
        if constexpr (STAT) {
          if constexpr (FIRST == 1 && r == 0) {
            const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, w[(CH * 3 + SS) * 3 + r], zero, 0, 0, 0);
          } else {
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, w[(CH * 3 + SS) * 3 + r], acc[t], 0, 0, 0);
          }
        } else {
          acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, wr[r], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + 3 < NM) PLHIP_PATCH_READ(i + 3);
        if constexpr (i % IEV == 1 && i / IEV < NPI) issue_piece(integral_constant<int, i / IEV>{}, qb, qc, qs, islot);
        __builtin_amdgcn_sched_barrier(0);
      });
#undef PLHIP_PATCH_READ
      static_assert(IEV * (NPI - 1) + 1 < NM, "every piece has its MFMA");
    }
    if (diag && nstep < 6 && lane == 0) lstamp[11 + nstep] = __builtin_amdgcn_s_memtime();
    ++nstep;
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  };

  // ---- prologue: the slots of the first D steps; register-resident weights: the three fragments of step p right behind
  // slab p, the rest behind slab D - 1: step 0 starts when a quarter of the prologue's bytes has arrived
  if (NH == 1 && a.delay > 0 && 2 * blockIdx.x >= gridDim.x) {  // the CU's second block starts late (see the template comment)
    for (int i = 0; i < a.delay; i += 100) __builtin_amdgcn_s_sleep(100);
  }
  cursor_tiles(0);
  patch_static_for<0, D>([&](auto p_c) __attribute__((always_inline)) {
    constexpr int pp = decltype(p_c)::value;
    issue(iqc, std::integral_constant<int, pp % SPG>{}, pp);
    advance_cursor();
    if constexpr (STAT) {
      load_w(std::integral_constant<int, 3 * pp>{});
      load_w(std::integral_constant<int, 3 * pp + 1>{});
      load_w(std::integral_constant<int, 3 * pp + 2>{});
    }
  });
  if constexpr (STAT) patch_static_for<3 * D, 18>([&](auto i_c) __attribute__((always_inline)) { load_w(i_c); });
  PLHIP_PATCH_STAMP(3);

  using std::integral_constant;
  int nstamp = 22;  // 5-10: barrier of step i passed, 11-16: its MFMAs issued, 18-21: first epilogue, 22..: end of round k
  for (int k = 0; k < a.rounds; ++k) {
    bool live;                                                     // wave-uniform
    const int tc = tile_of(k, li0 + half, live);
    const int b = (int)fastdiv_u31((uint32_t)tc, a.tpi_m, a.tpi_s);
    const int p0 = (tc - b * a.TPI) * NTH;
    if constexpr (STAT) {
      typedef integral_constant<int, 0> I0;
      typedef integral_constant<int, 1> I1;
      typedef integral_constant<int, 2> I2;
      const bool r0 = k == 0;
      step(I0{}, I0{}, I1{}, r0, false);
      step(I0{}, I1{}, I0{}, r0, false);
      step(I0{}, I2{}, I0{}, r0, false);
      step(I1{}, I0{}, I0{}, r0, false);
      step(I1{}, I1{}, I0{}, r0, false);
      step(I1{}, I2{}, I0{}, r0, false);
    } else {
      if (k == 0 && OUT != OUT_I32) {  // scale / bias: older than every DMA piece, so long landed at the first epilogue
        constexpr int NPRO = patch_npws_sum<KT, NPW>(0, D);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPRO) : "memory");
        asm volatile("" : "+v"(sc), "+v"(bi));
      }
      for (int ic = 0; ic < NCH; ++ic) {
        typedef integral_constant<int, -1> IR;
        step(IR{}, integral_constant<int, 0>{}, integral_constant<int, 2>{}, false, ic == 0);
        step(IR{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, false, false);
        step(IR{}, integral_constant<int, 2>{}, integral_constant<int, 0>{}, false, false);
        if constexpr (KT == 2) {
          step(IR{}, integral_constant<int, 3>{}, integral_constant<int, 0>{}, false, false);
          step(IR{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, false, false);
          step(IR{}, integral_constant<int, 5>{}, integral_constant<int, 0>{}, false, false);
        }
      }
    }
    if (live && !(a.dbg & 1)) {  // (PLHIP_PATCH_DEBUG & 1: no epilogue; timing experiments)
      // (opaque copies: the address arithmetic of the epilogue must not be hoisted above the K loop, where its lane masks
      // and offsets would sit in registers for the whole tile)
      int be = b, pe = p0, me = mt;
      asm volatile("" : "+s"(be), "+s"(pe), "+s"(me));
      // (the 2 x 2 layout has no registers to spare for a second epilogue body: conv_patch_supported keeps 7-wide planes with
      // M <= 64 on the ring kernel)
      if (WNH == 1 && a.glob) epilogue(std::true_type{}, be, pe, me);  // kernel-uniform
      else epilogue(std::false_type{}, be, pe, me);
      ++nepi;
    }
    if (nstamp < PATCH_STAMP_SLOTS - 3) {
      PLHIP_PATCH_STAMP(nstamp);
      ++nstamp;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMA pieces must land before the LDS is released
  if (diag) {  // wave-uniform
    if (lane == 0) {
      lstamp[PATCH_STAMP_SLOTS - 2] = __builtin_amdgcn_s_memtime();
      lstamp[PATCH_STAMP_SLOTS - 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.stamps && blockIdx.x < 512 && lane < PATCH_STAMP_SLOTS)
      a.stamps[((size_t)blockIdx.x * 8 + wave) * PATCH_STAMP_SLOTS + lane] = lstamp[lane];
  }
}

template <int NH, int WMH, int WNH, int NPW, int NSLOT, bool STAT, int OUT, int KT = 3>
static inline void launch_patch_t(const PatchArgs& a, hipStream_t s) {
  const size_t lds = (size_t)NSLOT * (NH * 32 * a.pitch + (STAT ? 0 : KT * WMH * 1024));
  const unsigned blocks = (unsigned)(8 * a.MB * a.NQ);
  const bool nonneg = a.act == ACT_RELU || a.act == ACT_RELU6;
  if (OUT == OUT_I8 && !nonneg) {
    auto kfn = conv_patch_i8_kernel<NH, WMH, WNH, NPW, NSLOT, STAT, OUT, false, KT>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256 * NH), lds, s, a);
  } else {
    auto kfn = conv_patch_i8_kernel<NH, WMH, WNH, NPW, NSLOT, STAT, OUT, true, KT>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256 * NH), lds, s, a);
  }
}

template <int NH, int WMH, int WNH, int NPW, int NSLOT, bool STAT, int KT = 3>
static inline void launch_patch_o(const PatchArgs& a, int out, hipStream_t s) {
  if (out == OUT_I32) launch_patch_t<NH, WMH, WNH, NPW, NSLOT, STAT, OUT_I32, KT>(a, s);
  else if (out == OUT_F32) launch_patch_t<NH, WMH, WNH, NPW, NSLOT, STAT, OUT_F32, KT>(a, s);
  else launch_patch_t<NH, WMH, WNH, NPW, NSLOT, STAT, OUT_I8, KT>(a, s);
}

}  // namespace plhip
