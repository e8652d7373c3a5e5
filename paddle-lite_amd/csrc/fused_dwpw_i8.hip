// fused_dwpw_i8.hip — depthwise 3x3 (int8 out) fused with the pointwise 1x1 convolution that consumes it.
//
// SURVEY.md §8(f) rank 1.  In the MobileNet programs every depthwise_conv2d [int8_out] feeds exactly one conv2d 1x1; run
// as two kernels the int8 intermediate makes a full HBM round trip (~40 % of the network's traffic).  Here it never
// leaves the CU: a workgroup owns BN = WN * 128 columns (output pixels) and BM = WM * 64 output channels, and walks the
// channels (= the GEMM's K) in steps of 32.  Three stages run concurrently, one raw s_barrier per K-step:
//   STAGE   (LDS-DMA, no registers)  the input rows the tile's depthwise windows touch, 32 channels per K-step, go
//           global -> LDS as 16-byte pieces (global_load_lds_dwordx4), L + 2 = 4 K-steps ahead of their use;
//   PRODUCE (VALU)  every wave computes its share of the depthwise outputs of the NEXT K-step — the reference's
//           arithmetic (conv_depthwise_3x3_int8_int8, lite/backends/arm/math/conv_impl.cc:909-1018: int32 taps, fmla
//           requantisation, round half away, clamp +-127): a lane = 4 consecutive output pixels of one channel,
//           3 UNALIGNED ds_read_b64 / b96 row windows (gfx950 serves unaligned LDS accesses: tools/probe_lds_unaligned.hip),
//           v_alignbyte + v_dot4_i32_i8 — and writes the dword straight into the K-step's activation image in LDS
//           ([kg = k/8][16-column chunk][k%8][16 B], the layout ds_read_b64_tr_b8 transposes from);
//   CONSUME (MFMA)  8 transposed LDS reads + 2 weight fragments (global -> registers, 3 K-steps ahead, packed MFMA order of
//           pack_weights_kernel) feed 8 v_mfma_i32_32x32x32_i8 of the wave's 128 (n) x 64 (m) tile, activations as the A
//           operand as in gemm_tr_i8.hip (a lane owns one output channel).
// All vector-memory traffic of the K loop is issued by hand (DMA builtin + inline-asm loads) and waited for with counted
// s_waitcnt vmcnt: the compiler cannot count LDS-DMA and would drain the queue.  The LDS-DMA ring is the ONLY dynamic
// LDS object; everything the compiler reads / writes with ordinary instructions lives in static LDS arrays, which it
// proves distinct from the DMA target (one shared object costs an s_waitcnt vmcnt(0) before every LDS access).
//
// Column space: QUADS of 4 consecutive output columns of one output row, enumerated over (image, row, quad);
// ceil(OW/4)*4 - OW trailing columns of a row's last quad are computed (from masked input) and never stored: the int8
// epilogue compacts them away in its LDS staging image and stores 16 contiguous bytes per lane.
// The result is bit-identical to depthwise [int8_out] followed by the 1x1 conv (tests/test_gpu_fused.py).
#ifdef PLHIP_EXPERIMENTS  // make EXPERIMENTS=1: measured slower than the two kernels on every MobileNet pair: not
                          // part of the default library; without it the entry points report "unsupported" and the
                          // depthwise kernel class runs the two kernels inside its one instruction
#include <stdlib.h>

#include <type_traits>

#include "plhip_device.h"
#include "plhip_kernels.h"
#include "dw_common.h"
#include "gemm_tr_common.h"

namespace plhip {

typedef int v3i __attribute__((ext_vector_type(3)));

constexpr int FZ_L = 2;            // K-steps of weights / staged input in flight behind the one being consumed
constexpr int FZ_D = FZ_L + 2;     // ring slots of staged input
constexpr int FZ_MAXSEG = 4;       // images a column tile may touch
constexpr int FZ_PAD = 16;         // LDS bytes in front of the ring (a window may start 3 bytes before its band)
constexpr int FZ_MAXC = 1024;
// DMA instructions per wave and K-step the kernel keeps source offsets for: wide tiles (few channels, big planes) need more
constexpr int fz_maxpwd(int wn) { return wn >= 8 ? 8 : (wn >= 2 ? 4 : 2); }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (clamped to 40: a smaller count only waits longer)
__device__ __forceinline__ void fz_wait_vmcnt(int n) {
#define FZ_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
  // the steady-state counts of the common shapes first (1 or 2 DMA instructions per wave and K-step): the general switch
  // below is a 6-deep tree of scalar compares and taken branches, ~300 cycles per K-step in the timeline
  if (n == 6) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    return;
  }
  if (n == 8) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    return;
  }
  switch (n < 40 ? n : 40) {
    FZ_W(0) FZ_W(1) FZ_W(2) FZ_W(3) FZ_W(4) FZ_W(5) FZ_W(6) FZ_W(7) FZ_W(8) FZ_W(9) FZ_W(10) FZ_W(11) FZ_W(12) FZ_W(13)
    FZ_W(14) FZ_W(15) FZ_W(16) FZ_W(17) FZ_W(18) FZ_W(19) FZ_W(20) FZ_W(21) FZ_W(22) FZ_W(23) FZ_W(24) FZ_W(25) FZ_W(26)
    FZ_W(27) FZ_W(28) FZ_W(29) FZ_W(30) FZ_W(31) FZ_W(32) FZ_W(33) FZ_W(34) FZ_W(35) FZ_W(36) FZ_W(37) FZ_W(38) FZ_W(39)
    default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
  }
#undef FZ_W
}

// 4 depthwise outputs (one quad) of one channel from its 3 row windows (byte 0 = input column 4*xq*S - pl) -> one dword.
// p4: (w row0 | w row1 | w row2 | 2*scale), pb: 2*bias.  UNS: relu / relu6 (results 0 .. 127: the packed (+1, >>1)
// finish of dw_requant4), bounds [0, hi2]; else none / leaky as leaky with slope `alpha` (1 for none: exact).
template <int S, bool UNS>
__device__ __forceinline__ uint32_t fz_compute(const uint32_t (&in)[3][S == 1 ? 2 : 3], const uint32_t (&cmask)[S == 1 ? 2 : 3],
                                               const bool (&rowv)[3], const v4i p4, const uint32_t pb, float alpha, float hi2) {
  constexpr int ND = S == 1 ? 2 : 3;
  int acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    uint32_t e[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) e[i] = in[r][i] & cmask[i];  // columns left / right of the image: zero padding
    uint32_t win[4];
    if (S == 1) {
      win[0] = e[0];
      win[1] = __builtin_amdgcn_alignbyte(e[1], e[0], 1);
      win[2] = __builtin_amdgcn_alignbyte(e[1], e[0], 2);
      win[3] = __builtin_amdgcn_alignbyte(e[1], e[0], 3);
    } else {
      win[0] = e[0];
      win[1] = __builtin_amdgcn_alignbyte(e[1], e[0], 2);
      win[2] = e[1];
      win[3] = __builtin_amdgcn_alignbyte(e[ND - 1], e[1], 2);
    }
    const int wr = rowv[r] ? p4[r] : 0;  // a row above / below the image: zero weights
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_sdot4((int)win[j], wr, acc[j], false);
  }
  if (UNS) return dw_requant4<ACT_RELU6>(acc, __uint_as_float((uint32_t)p4[3]), __uint_as_float(pb), alpha, 0.f, hi2);
  return dw_requant4<ACT_LEAKY>(acc, __uint_as_float((uint32_t)p4[3]), __uint_as_float(pb), alpha, -254.f, 254.f);
}

// flat pixel index (over all images) of the first column of quad Q (both < 2^31: fused_dwpw_plan)
__device__ __forceinline__ int fz_pixel(int Q, int owq, int ow) {
  const int gr = Q / owq;
  return gr * ow + 4 * (Q - gr * owq);
}

// output address pieces of one quad Q of the column space (32-bit outputs)
struct FzOut {
  size_t off;  // element offset of the quad's first column inside a channel plane of y, image offset included
  int room;    // valid columns (0: quad outside the tensor)
};
__device__ __forceinline__ FzOut fz_quad_out(long Q, long NQ, int owq, int oh, int ow, size_t y_bstride) {
  FzOut o;
  o.off = 0;
  o.room = 0;
  if (Q < NQ) {
    const int gr = (int)(Q / owq), xq = (int)(Q - (long)gr * owq);
    const int b = gr / oh, oy = gr - b * oh;
    o.off = (size_t)b * y_bstride + (size_t)oy * ow + 4 * xq;
    o.room = ow - 4 * xq < 4 ? ow - 4 * xq : 4;
  }
  return o;
}

// ---- diagnostic timeline (PLHIP_FUSED_DEBUG & 32; never set in production): per-wave s_memtime stamps kept in LDS and
// flushed at the end (plhip_debug_read_fz_stamps; tools/fused_timeline.py).  Slots: 0 realtime start, 1 entry, 2 dw
// parameters visible, 3 first K-step produced, 4+ks top of K-step ks (ks < 16), 20-24 sub-stamps of K-step 6 (& 64), 26 loop end, 27 staged, 28 stores issued,
// 29 stores acknowledged, 31 realtime end
constexpr int FZ_STAMP_SLOTS = 32;
__device__ unsigned long long g_fz_stamps[1024 * 8 * FZ_STAMP_SLOTS];
#define PLHIP_FZ_STAMP(i)                                             \
  do {                                                                \
    if (diag && lane == 0) lstamp[i] = __builtin_amdgcn_s_memtime();  \
  } while (0)

template <int WN, int WM, int OUT, int S>
__global__ __launch_bounds__(512, 2) void fused_dwpw_kernel(FusedArgs a) {
  static_assert(WN * WM == 8, "8 waves: 2 per SIMD, 256 registers each");
  constexpr int ND = S == 1 ? 2 : 3;
  constexpr int L = FZ_L, D = FZ_D;
  constexpr int MAXPWD = fz_maxpwd(WN);
  constexpr int BM = WM * 64;
  constexpr int IPW = 16 / WM;         // wave-items (2 channels x 32 quads) per K-step and wave
  constexpr int ACT_SLOT = WN * 128 * 32;
  constexpr int STG = 8 * 64 * 144;    // int8 epilogue staging, overlays the activation slots
  constexpr int REGION0 = (OUT == OUT_I8 && STG > 2 * ACT_SLOT) ? STG : 2 * ACT_SLOT;
  const GemmArgs& g = a.pw;
  PLHIP_PRELOAD(a.x); PLHIP_PRELOAD(a.dw_w); PLHIP_PRELOAD(a.dw_scale); PLHIP_PRELOAD(a.dw_bias); PLHIP_PRELOAD(a.dw_act);
  PLHIP_PRELOAD(a.dw_alpha); PLHIP_PRELOAD(a.n); PLHIP_PRELOAD(a.C); PLHIP_PRELOAD(a.h); PLHIP_PRELOAD(a.w); PLHIP_PRELOAD(a.oh);
  PLHIP_PRELOAD(a.ow); PLHIP_PRELOAD(a.pt); PLHIP_PRELOAD(a.pl); PLHIP_PRELOAD(a.owq); PLHIP_PRELOAD(a.NQ);
  PLHIP_PRELOAD(a.slot_bytes); PLHIP_PRELOAD(a.ni); PLHIP_PRELOAD(a.pwd); PLHIP_PRELOAD(g.wp); PLHIP_PRELOAD(g.y);
  PLHIP_PRELOAD(g.scale); PLHIP_PRELOAD(g.bias); PLHIP_PRELOAD(g.M); PLHIP_PRELOAD(g.KS); PLHIP_PRELOAD(g.HWY);
  PLHIP_PRELOAD(g.y_bstride); PLHIP_PRELOAD(g.MT); PLHIP_PRELOAD(g.NT); PLHIP_PRELOAD(g.act); PLHIP_PRELOAD(g.alpha);
  PLHIP_PRELOAD(g.dbg);
  // dynamic LDS = the LDS-DMA target + what is only ever read by inline asm beside it:
  // [PAD][min(D, KS) slots][1 KiB dummy][PAD][dw parameters: C x 32 B (w0w1w2 | w3w4w5 | w6w7w8 | 2*scale | 2*bias | ...)]
  extern __shared__ __attribute__((aligned(16))) uint8_t raw[];
  __shared__ __attribute__((aligned(16))) uint8_t fsm[REGION0];  // 2 activation slots | int8 staging
  __shared__ unsigned long long lstamp_all[8 * FZ_STAMP_SLOTS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave / WM, wm = wave - wn * WM;  // wave-uniform: column group, channel slice (producer) / m slice (consumer)
  const int c = lane & 31, h = lane >> 5;
  int mb, nb;
  tr_xcd_tile_map(blockIdx.x, g.MT, g.NT, mb, nb);
  if (nb >= g.NT) return;  // block-uniform (grid padded to 8 N blocks)
  const int dbg = g.dbg;   // PLHIP_FUSED_DEBUG (timing experiments only): 4 = no depthwise arithmetic, 32 = stamps
  const bool diag = (dbg & 32) != 0;
  unsigned long long* lstamp = lstamp_all + wave * FZ_STAMP_SLOTS;
  if (diag && lane == 0) {
    lstamp[0] = __builtin_amdgcn_s_memrealtime();
    lstamp[1] = __builtin_amdgcn_s_memtime();
  }
  const int KS = g.KS;
  const int MT32 = (g.M + 31) >> 5;
  const int owq = a.owq;
  const long NQ = a.NQ;
  const int hw = a.h * a.w;
  const int TOTAL = a.n * a.C * hw;  // < 2^31 (checked by the launcher)

  // ---- the tile's images ("segments") and, per segment, the band of input rows its depthwise windows touch ----
  const long Qb = (long)nb * (32 * WN);
  const long Qe = (Qb + 32 * WN < NQ ? Qb + 32 * WN : NQ) - 1;
  const int grA = (int)(Qb / owq), grB = (int)(Qe / owq);
  const int bA = grA / a.oh, bB = grB / a.oh;
  // Staged layout of one K-step: [segment][channel][input row][PPR pieces]: every input row of the band is fetched on
  // its own, from `pl` bytes before its first column, into a 16-byte aligned LDS row: the window of quad xq then starts at
  // LDS row offset 4 * xq * S — dword aligned (unaligned LDS reads work on gfx950 but run several times slower).
  const int PPR = (a.w + a.pl + 15) >> 4;  // pieces per row
  int seg_ppc[FZ_MAXSEG], seg_po[FZ_MAXSEG + 1], seg_iylo[FZ_MAXSEG], seg_iyhi[FZ_MAXSEG];
  seg_po[0] = 0;
#pragma unroll
  for (int s = 0; s < FZ_MAXSEG; ++s) {
    const int b = bA + s;
    const bool live = b <= bB;
    const int oy_lo = s == 0 ? grA - bA * a.oh : 0;
    const int oy_hi = b == bB ? grB - bB * a.oh : a.oh - 1;
    int iy_lo = oy_lo * S - a.pt, iy_hi = oy_hi * S - a.pt + 2;
    iy_lo = iy_lo < 0 ? 0 : iy_lo;
    iy_hi = iy_hi > a.h - 1 ? a.h - 1 : iy_hi;
    seg_iylo[s] = iy_lo;
    seg_iyhi[s] = iy_hi;
    seg_ppc[s] = live ? (iy_hi - iy_lo + 1) * PPR : 0;  // pieces per channel
    seg_po[s + 1] = seg_po[s] + 32 * seg_ppc[s];
  }
  const int TP = seg_po[FZ_MAXSEG];  // 16-byte pieces of one K-step of this tile

  // ---- DMA pieces of this wave: instruction j = q * 8 + wave of the slot (q < pwd), piece gi = 64 j + lane, laid out
  // [segment][channel][piece] so that the LDS destination is lane-linear.  doff: source byte offset for K-step 0.
  int doff[MAXPWD];
#pragma unroll
  for (int q = 0; q < MAXPWD; ++q) {
    doff[q] = TOTAL;
    if (q < a.pwd) {
      const int gi = (q * 8 + wave) * 64 + lane;
      if (gi < TP) {
        const int s = (gi >= seg_po[1]) + (gi >= seg_po[2]) + (gi >= seg_po[3]);
        const int po = s == 0 ? seg_po[0] : (s == 1 ? seg_po[1] : (s == 2 ? seg_po[2] : seg_po[3]));
        const int ppc = s == 0 ? seg_ppc[0] : (s == 1 ? seg_ppc[1] : (s == 2 ? seg_ppc[2] : seg_ppc[3]));
        const int ylo = s == 0 ? seg_iylo[0] : (s == 1 ? seg_iylo[1] : (s == 2 ? seg_iylo[2] : seg_iylo[3]));
        const int r = gi - po;
        const int ch = r / ppc, rem = r - ch * ppc;
        const int row = rem / PPR, pc = rem - row * PPR;
        doff[q] = ((bA + s) * a.C + ch) * hw + (ylo + row) * a.w - a.pl + 16 * pc;  // -pl .. : only the tensor's first row
      } else {
        doff[q] = TOTAL;  // nothing to fetch
      }
    }
  }
  // K-step kf of the input -> ring slot kf % D.  A source range leaving the tensor (the first row of the first plane
  // starts `pl` bytes early; the last piece of the last plane's last row; channels past C in the last K-step) is clamped
  // into it: memory safe; the pieces whose bytes matter are repaired by fix_edges below.  K-steps past KS (the pipeline's
  // run-out) and surplus instructions fetch one line for the whole wave into the dummy KiB.
  const int dummy_off = FZ_PAD + (KS < D ? KS : D) * a.slot_bytes;
  uint32_t* prm = reinterpret_cast<uint32_t*>(raw + dummy_off + 1024 + FZ_PAD);
  auto issue_raw = [&](int kf) __attribute__((always_inline)) {
    const bool past = kf >= KS;
    const int kc = past ? KS - 1 : kf;
    const int slot = FZ_PAD + (kc & (D - 1)) * a.slot_bytes;
    const int koff = kc * 32 * hw;  // wave-uniform
#pragma unroll
    for (int q = 0; q < MAXPWD; ++q) {
      if (q < a.pwd) {  // kernel-uniform
        const int j = q * 8 + wave;
        const bool idle = past || j >= a.ni;
        int so = idle ? 0 : doff[q] + koff;
        so = so < 0 ? 0 : (so > TOTAL - 16 ? TOTAL - 16 : so);
        const int dst = idle ? dummy_off : slot + j * 1024;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(a.x + so), (lds_ptr_t)(raw + dst), 16, 0, 0);
      }
    }
  };
  // repair of the (at most two) pieces of the whole launch whose true source range crosses the tensor's bounds: the
  // lane that fetched it rewrites it with bounds-checked byte loads (block-uniform test first: K-step 0 of image 0, the
  // K-step of channel C-1 of image n-1)
  const int KL = (a.C - 1) >> 5;
  auto fix_edges = [&](int kf) __attribute__((always_inline)) {
    if (!((kf == 0 && bA == 0) || (kf == KL && bB == a.n - 1))) return;  // block-uniform
    asm volatile("" : "+s"(kf));  // opaque: or the loop-invariant byte addresses below are hoisted out of the K loop and spilled
    const int slot = FZ_PAD + (kf & (D - 1)) * a.slot_bytes;
#pragma unroll
    for (int q = 0; q < MAXPWD; ++q) {
      if (q < a.pwd) {
        const int j = q * 8 + wave;
        const int so = doff[q] + kf * 32 * hw;
        // my piece: starts inside the tensor's last plane and crosses its end, or starts before the tensor
        if (j < a.ni && doff[q] < TOTAL && (so < 0 || (so > TOTAL - 16 && so >= TOTAL - hw && so < TOTAL))) {
          uint32_t v[4] = {0, 0, 0, 0};
          for (int i = 0; i < 16; ++i)
            if (so + i >= 0 && so + i < TOTAL) v[i >> 2] |= (uint32_t)(uint8_t)a.x[so + i] << (8 * (i & 3));
          const v4i vv = {(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
          *reinterpret_cast<v4i*>(raw + slot + j * 1024 + lane * 16) = vv;
        }
      }
    }
  };

  // ---- weight fragments of this wave's two 32-row m tiles: global -> registers by hand, L + 1 K-steps ahead ----
  int wt[2];  // tiles past M: any packed tile (their outputs are never stored)
#pragma unroll
  for (int u = 0; u < 2; ++u) wt[u] = min(mb * (BM / 32) + wm * 2 + u, MT32 - 1) * KS;  // wave-uniform
  const v4i* wlane = reinterpret_cast<const v4i*>(g.wp) + lane;
  v4i wset[L + 2][2];
  auto issue_w = [&](int kf, v4i (&w)[2]) __attribute__((always_inline)) {
    const int kc = kf < KS ? kf : KS - 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const v4i* p = wlane + (size_t)(wt[u] + kc) * 64;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[u]) : "v"(p) : "memory");
    }
  };

  // ---- pipeline prologue: raw(0); {raw(1), w(0)}; ... {raw(L+1), w(L)} ----
  issue_raw(0);
#pragma unroll
  for (int p = 0; p <= L; ++p) {
    issue_raw(p + 1);
    issue_w(p, wset[p]);
  }

  // ---- depthwise parameters -> LDS ----
  for (int k = threadIdx.x; k < a.C; k += 512) {
    const int8_t* wp = a.dw_w + (size_t)k * 9;
    uint32_t w0, w1, w2;
    __builtin_memcpy(&w0, wp, 4);
    __builtin_memcpy(&w1, wp + 3, 4);
    __builtin_memcpy(&w2, wp + 5, 4);  // one byte early + shift: never reads past the filter tensor
    const float s = a.dw_scale[k], b = a.dw_bias ? a.dw_bias[k] : 0.f;
    const v4i p4 = {(int)(w0 & 0xffffffu), (int)(w1 & 0xffffffu), (int)(w2 >> 8), (int)__float_as_uint(s + s)};
    *reinterpret_cast<v4i*>(prm + 8 * k) = p4;
    prm[8 * k + 4] = __float_as_uint(b + b);
  }
  // ---- this lane's quad (producer role): quad c of group wn; channel parity h ----
  int wa[3];           // LDS byte offset (inside a slot) of the row windows of channel parity h of the K-step's first pair
  int band;            // LDS bytes per channel of this lane's segment
  bool rowv[3];        // input row inside the image
  uint32_t cmask[ND];  // byte-validity masks of a window's dwords
  {
    long Q = Qb + wn * 32 + c;
    if (Q > Qe) Q = Qb;  // dead quads compute the tile's first quad again: results never stored
    const int gr = (int)(Q / owq), xq = (int)(Q - (long)gr * owq);
    const int b = gr / a.oh, oy = gr - b * a.oh;
    const int s = b - bA;
    const int po = s == 0 ? seg_po[0] : (s == 1 ? seg_po[1] : (s == 2 ? seg_po[2] : seg_po[3]));
    const int ppc = s == 0 ? seg_ppc[0] : (s == 1 ? seg_ppc[1] : (s == 2 ? seg_ppc[2] : seg_ppc[3]));
    const int ylo = s == 0 ? seg_iylo[0] : (s == 1 ? seg_iylo[1] : (s == 2 ? seg_iylo[2] : seg_iylo[3]));
    const int yhi = s == 0 ? seg_iyhi[0] : (s == 1 ? seg_iyhi[1] : (s == 2 ? seg_iyhi[2] : seg_iyhi[3]));
    band = 16 * ppc;
    const int iy0 = oy * S - a.pt;
    const int start = 4 * xq * S - a.pl;  // input column of the window's byte 0 (>= -3: masked)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ih = iy0 + r;
      rowv[r] = ih >= 0 && ih < a.h;
      const int ihc = ih < ylo ? ylo : (ih > yhi ? yhi : ih);  // rows outside the image: any row of the band
      wa[r] = 16 * po + h * band + (ihc - ylo) * (16 * PPR) + 4 * xq * S;  // dword aligned
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      uint32_t m = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = start + 4 * d + i;
        if (col >= 0 && col < a.w) m |= 0xffu << (8 * i);
      }
      cmask[d] = m;
    }
  }
  // LDS destination of this lane's dword inside an activation slot, channel-independent part
  const int wr_lane = wn * 4096 + (c >> 2) * 128 + (c & 3) * 4 + h * 16;
  const int ch0 = wm * (2 * IPW);  // first channel (inside a K-step) of this wave's slice
  const bool dw_uns = a.dw_act == ACT_RELU || a.dw_act == ACT_RELU6;
  const float dw_hi2 = a.dw_act == ACT_RELU6 ? fminf(a.dw_alpha + a.dw_alpha, 254.f) : 254.f;
  const float dw_slope = a.dw_act == ACT_LEAKY ? a.dw_alpha : 1.f;
  const uint32_t raw_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)raw;
  const uint32_t prm_addr = raw_addr + dummy_off + 1024 + FZ_PAD;

  // PRODUCE one K-step: IPW items, two at a time (6 window reads in flight, then their arithmetic)
  auto produce_step = [&](int ks, uint8_t* slot_act, auto uns_c) __attribute__((always_inline)) {
    constexpr bool UNS = decltype(uns_c)::value;
    const uint32_t sbase = raw_addr + FZ_PAD + (ks & (D - 1)) * a.slot_bytes;
#pragma unroll 1  // a real loop: unrolled, the scheduler hoists every pair's parameter reads and spills
    for (int it = 0; it < IPW; it += 2) {
      v2i dd[2][3];  // asm outputs: nothing may touch them before the wait below names them
      uint32_t d2[2][3];  // third dword of a stride-2 window
      v4i p4[2];
      uint32_t pb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = ch0 + 2 * (it + i);  // even channel of the pair inside the K-step (wave-uniform)
        {
          const int k = ks * 32 + kk + h;
          const int kc = k < a.C ? k : a.C - 1;  // channels past C meet zero-padded weights: any parameters will do
          const uint32_t pa = prm_addr + 32 * kc;
          asm volatile("ds_read_b128 %0, %1" : "=v"(p4[i]) : "v"(pa));
          asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(pb[i]) : "v"(pa));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const uint32_t ad = sbase + (uint32_t)(wa[r] + kk * band);
          // 4-byte aligned (S = 1) / 8-byte aligned (S = 2)
          if (ND == 2) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(dd[i][r]) : "v"(ad));
          else asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %2 offset:8" : "=&v"(dd[i][r]), "=&v"(d2[i][r]) : "v"(ad));
        }
      }
      if (ND == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dd[0][0]), "+v"(dd[0][1]), "+v"(dd[0][2]), "+v"(dd[1][0]), "+v"(dd[1][1]), "+v"(dd[1][2]), "+v"(p4[0]),
                       "+v"(p4[1]), "+v"(pb[0]), "+v"(pb[1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dd[0][0]), "+v"(dd[0][1]), "+v"(dd[0][2]), "+v"(dd[1][0]), "+v"(dd[1][1]), "+v"(dd[1][2]), "+v"(p4[0]),
                       "+v"(p4[1]), "+v"(pb[0]), "+v"(pb[1]), "+v"(d2[0][0]), "+v"(d2[0][1]), "+v"(d2[0][2]), "+v"(d2[1][0]),
                       "+v"(d2[1][1]), "+v"(d2[1][2]));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = ch0 + 2 * (it + i);
        uint32_t in[3][ND];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          in[r][0] = (uint32_t)dd[i][r][0];
          in[r][1] = (uint32_t)dd[i][r][1];
          if (ND == 3) in[r][ND - 1] = d2[i][r];
        }
        const uint32_t pk = (dbg & 4) ? (in[0][0] ^ in[1][1] ^ in[2][0] ^ (uint32_t)p4[i][0] ^ pb[i])
                                      : fz_compute<S, UNS>(in, cmask, rowv, p4[i], pb[i], dw_slope, dw_hi2);
        *reinterpret_cast<uint32_t*>(slot_act + wr_lane + (kk >> 3) * 1024 + (kk & 7) * 16) = pk;
      }
    }
  };
  auto produce = [&](int ks, uint8_t* slot_act) __attribute__((always_inline)) {
    if (dw_uns) produce_step(ks, slot_act, std::integral_constant<bool, true>{});
    else produce_step(ks, slot_act, std::integral_constant<bool, false>{});
  };

  // transposed-read address of this lane inside its activation group (gemm_tr_i8.hip): k half h -> kg {2h, 2h+1};
  // 16-lane group parity -> chunk j (even / odd); lane 2q+p of the group -> row q, sub-chunk p
  const int tr_lane = wn * 4096 + (h * 2) * 1024 + ((lane >> 4) & 1) * 128 + ((lane & 15) >> 1) * 16 + (lane & 1) * 8;

  v16i acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0;
  const int step_ops = a.pwd + 2;  // vector-memory instructions a wave issues per K-step
  // once, everything: the parameter / scale loads above are the compiler's; from here on only hand-counted traffic flies
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  fix_edges(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // dw parameters + everyone's raw(0) pieces visible
  PLHIP_FZ_STAMP(2);
  produce(0, fsm);
  PLHIP_FZ_STAMP(3);

  // K-step ks: weights(ks) live in wset[CS]; weights(ks + L + 1) are requested into wset[(CS + L + 1) % (L + 2)]
  auto step = [&](int ks, auto cs_c) __attribute__((always_inline)) {
    constexpr int CS = decltype(cs_c)::value;
    if (ks < 16) PLHIP_FZ_STAMP(4 + ks);
    const bool sub = diag && (dbg & 64) && ks == 6;  // sub-stamps of one steady-state K-step: slots 20 .. 24
    // everything this wave issued up to K-step ks - L - 1 has landed: raw(ks + 1) and weights(ks)
    fz_wait_vmcnt(L * step_ops);
    asm volatile("" : "+v"(wset[CS][0]), "+v"(wset[CS][1]));
    fix_edges(ks + 1);
    // my writes of activation K-step ks are done; after the barrier everyone's raw(ks + 1) pieces and activation
    // K-step ks are visible, and nobody reads raw(ks) / activation K-step ks - 1 any more
    if (sub && lane == 0) lstamp[20] = __builtin_amdgcn_s_memtime();  // counted wait done
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (sub && lane == 0) lstamp[21] = __builtin_amdgcn_s_memtime();  // barrier passed
    uint8_t* cur = fsm + (ks & 1) * ACT_SLOT;
    uint8_t* nxt = fsm + ((ks + 1) & 1) * ACT_SLOT;
    v2i lo[4], hi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      lo[t] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(cur + tr_lane + t * 256));
      hi[t] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr_t)(cur + tr_lane + t * 256 + 1024));
    }
    issue_raw(ks + L + 2);
    issue_w(ks + L + 1, wset[(CS + L + 1) % (L + 2)]);
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const v4i av = {lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
          acc[t][u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, wset[CS][u], acc[t][u], 0, 0, 0);
        }
    };
    // MFMAs FIRST: a wave sits in its 8 MFMAs for ~256 cycles while the matrix pipe is busy; the other wave of the SIMD
    // queues behind it, so the two skew by one MFMA block and each one's depthwise VALU work then overlaps the other's
    // MFMAs (produce-first made both waves compete for the VALU and then both queue for the matrix pipe)
    if (sub && lane == 0) lstamp[22] = __builtin_amdgcn_s_memtime();  // transposed reads + DMA + weight loads issued
    multiply();
    if (sub && lane == 0) lstamp[23] = __builtin_amdgcn_s_memtime();  // MFMAs issued
    if (ks + 1 < KS) produce(ks + 1, nxt);
    if (sub && lane == 0) lstamp[24] = __builtin_amdgcn_s_memtime();  // next K-step's activations written
  };
  static_assert(L + 2 == 4 && D == 4, "the loop below is unrolled for 4 weight sets / ring slots");
  for (int ks = 0; ks < KS; ks += 4) {
    step(ks, std::integral_constant<int, 0>{});
    if (ks + 1 < KS) step(ks + 1, std::integral_constant<int, 1>{});
    if (ks + 2 < KS) step(ks + 2, std::integral_constant<int, 2>{});
    if (ks + 3 < KS) step(ks + 3, std::integral_constant<int, 3>{});
  }
  // run-out requests (dummy KiB, repeated weights) are still in flight: drain them before the registers are reused
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wset[0][0]), "+v"(wset[0][1]), "+v"(wset[1][0]), "+v"(wset[1][1]), "+v"(wset[2][0]),
               "+v"(wset[2][1]), "+v"(wset[3][0]), "+v"(wset[3][1]));
  PLHIP_FZ_STAMP(26);

  // ---- epilogue: lane (c, h) owns channel rows mrow[0], mrow[1]; per n tile t, register r <-> n = 32t + 8(r>>2) + 4h + (r&3)
  const long Qw = Qb + wn * 32;  // first quad of this wave's 128 columns
  // this lane's output channels: scale / bias (loaded here: the K loop has no registers to spare)
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

  if (OUT == OUT_I8) {
    __builtin_amdgcn_s_barrier();  // every wave has finished reading the activation slots: they become staging space
    uint8_t* stg = fsm + wave * (64 * 144);
    switch (g.act) {  // wave-uniform: straight-line requantisation per activation; PADDED image: quad qi at byte 4 * qi
      case ACT_RELU: tr_stage_i8<ACT_RELU>(acc, sc, bi, g.alpha, stg, c, h); break;
      case ACT_RELU6: tr_stage_i8<ACT_RELU6>(acc, sc, bi, g.alpha, stg, c, h); break;
      case ACT_LEAKY: tr_stage_i8<ACT_LEAKY>(acc, sc, bi, g.alpha, stg, c, h); break;
      default: tr_stage_i8<ACT_NONE>(acc, sc, bi, g.alpha, stg, c, h); break;
    }
    PLHIP_FZ_STAMP(27);
    // lane -> row lane>>3 of each 8-row round, COMPACT bytes [16 j, 16 j + 16), j = lane & 7, of the wave's pixels (the
    // padding columns of a row's last quad dropped)
    const int HW = g.HWY;
    const int P0 = fz_pixel((int)Qw, owq, a.ow);
    const int Pend = fz_pixel((int)(Qw + 32 < NQ ? Qw + 32 : NQ), owq, a.ow);  // fz_pixel(NQ) = n * HW
    const int P = P0 + 16 * (lane & 7);
    int nvalid = Pend - P;
    nvalid = nvalid < 0 ? 0 : (nvalid > 16 ? 16 : nvalid);
    const int b = P / HW;
    const int p = P - b * HW;
    const int cnt1 = HW - p < nvalid ? HW - p : nvalid;  // bytes in image b; the rest opens image b + 1 (HW >= 16)
    const int m0 = mb * BM + wm * 64 + (lane >> 3);
    int8_t* y1 = reinterpret_cast<int8_t*>(g.y) + (size_t)b * g.y_bstride + (size_t)m0 * (uint32_t)HW + p;
    int8_t* y2 = reinterpret_cast<int8_t*>(g.y) + (size_t)(b + 1) * g.y_bstride + (size_t)m0 * (uint32_t)HW - cnt1;
    const uint8_t* rrow = stg + (lane >> 3) * 144;
    // OW % 4 != 0: compact dword d of this lane = the last roomA - sA bytes of quad A followed by the first bytes of quad
    // A + 1: two aligned LDS dwords, one shift, one v_alignbyte (OW % 4 == 1, where a dword can span three quads, is kept
    // off this path by fused_dwpw_plan)
    const bool packed = (a.ow & 3) != 0;  // kernel-uniform
    int offA[4] = {0, 0, 0, 0}, shl[4] = {0, 0, 0, 0}, sel[4] = {0, 0, 0, 0};
    if (packed) {
      int gr = P / a.ow, x = P - gr * a.ow;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int Q = gr * owq + (x >> 2);
        const int roomA = a.ow - (x & ~3) < 4 ? a.ow - (x & ~3) : 4;
        int oa = 4 * (Q - (int)Qw);
        offA[d] = oa < 0 ? 0 : (oa > 120 ? 120 : oa);  // lanes past the wave's pixels: any legal offset (nothing stored)
        shl[d] = 8 * (4 - roomA);
        sel[d] = 4 - roomA + (x & 3);
        x += 4;
        if (x >= a.ow) {
          x -= a.ow;
          ++gr;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v4i v;
      if (!packed) {
        v = *reinterpret_cast<const v4i*>(rrow + i * 8 * 144 + (lane & 7) * 16);
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t qa = *reinterpret_cast<const uint32_t*>(rrow + i * 8 * 144 + offA[d]);
          const uint32_t qb = *reinterpret_cast<const uint32_t*>(rrow + i * 8 * 144 + offA[d] + 4);
          v[d] = (int)__builtin_amdgcn_alignbyte(qb, qa << shl[d], (uint32_t)sel[d]);
        }
      }
      if (m0 + 8 * i >= g.M || nvalid == 0) continue;
      const size_t ro = (size_t)(8 * i) * (uint32_t)HW;
      if (cnt1 == 16) {
        __builtin_memcpy(y1 + ro, &v, 16);  // possibly unaligned: fine for global memory
      } else if (!(a.ow & 1)) {
        // the tile's last bytes / a run crossing into the next image (rare lanes); even OW: everything is 2-byte aligned
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (2 * e < nvalid) {
            const uint16_t hv = (uint16_t)((uint32_t)v[e >> 1] >> (16 * (e & 1)));
            *reinterpret_cast<uint16_t*>((2 * e < cnt1 ? y1 : y2) + ro + 2 * e) = hv;
          }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (e < nvalid) ((e < cnt1 ? y1 : y2))[ro + e] = (int8_t)((uint32_t)v[e >> 2] >> (8 * (e & 3)));
      }
    }
  } else {
    // 32-bit outputs: a lane's 4 consecutive n of register group gq are one quad: quad 8t + 2gq + h of the wave's 32
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const FzOut qo = fz_quad_out(Qw + 8 * t + 2 * gq + h, NQ, owq, a.oh, a.ow, g.y_bstride);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (qo.room == 0 || mrow[u] >= g.M) continue;
          const size_t yoff = (size_t)mrow[u] * (uint32_t)g.HWY + qo.off;
          if (OUT == OUT_I32) {
            int* yp = reinterpret_cast<int*>(g.y) + yoff;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < qo.room) yp[e] = acc[t][u][4 * gq + e];
          } else {
            float* yp = reinterpret_cast<float*>(g.y) + yoff;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < qo.room) yp[e] = epilogue_f32(acc[t][u][4 * gq + e], sc[u], bi[u], g.act, g.alpha);
          }
        }
      }
    }
  }
  if (diag) {  // wave-uniform
    PLHIP_FZ_STAMP(28);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      lstamp[29] = __builtin_amdgcn_s_memtime();
      lstamp[31] = __builtin_amdgcn_s_memrealtime();
    }
    if (blockIdx.x < 1024 && lane < FZ_STAMP_SLOTS)
      g_fz_stamps[((size_t)blockIdx.x * 8 + wave) * FZ_STAMP_SLOTS + lane] = lstamp[lane];
  }
}

int debug_read_fz_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(g_fz_stamps)) bytes = sizeof(g_fz_stamps);
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fz_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

// Tile shape.  Preferred: BM = the smallest of 64 / 128 / 256 / 512 covering M (the depthwise stage is computed once per
// column tile and the column tile is as wide as 8 waves allow); M > 512 runs as ceil(M / 512) workgroups per column tile,
// each recomputing the depthwise outputs it consumes.  When 4 K-steps of the wide tile's staged input do not fit the
// LDS, the tile is narrowed (WN halves, BM doubles past M: the surplus m slices multiply tiles that are never stored, which
// costs nothing where the depthwise VALU work is the bound).
static bool fused_plan_tile(FusedArgs* a, int wn, int out) {
  const long qt = 32L * wn;  // quads per column tile
  const long nt = (a->NQ + qt - 1) / qt;
  // the largest tile: every tile when there are few, else the first 4096 (tiles repeat with the image period) + the last
  long tp_max = 0;
  const long probe = nt < 4096 ? nt : 4096;
  for (long i = 0; i <= probe; ++i) {
    const long nb = i < probe ? i : nt - 1;
    const long Qb = nb * qt, Qe = (Qb + qt < a->NQ ? Qb + qt : a->NQ) - 1;
    const long grA = Qb / a->owq, grB = Qe / a->owq;
    const long bA = grA / a->oh, bB = grB / a->oh;
    if (bB - bA + 1 > FZ_MAXSEG) return false;
    long tp = 0;
    for (long b = bA; b <= bB; ++b) {
      const long oy_lo = b == bA ? grA - bA * a->oh : 0, oy_hi = b == bB ? grB - bB * a->oh : a->oh - 1;
      long iy_lo = oy_lo * a->stride - a->pt, iy_hi = oy_hi * a->stride - a->pt + 2;
      iy_lo = iy_lo < 0 ? 0 : iy_lo;
      iy_hi = iy_hi > a->h - 1 ? a->h - 1 : iy_hi;
      if (iy_hi < iy_lo) return false;
      tp += 32 * (iy_hi - iy_lo + 1) * ((a->w + a->pl + 15) / 16);
    }
    tp_max = tp > tp_max ? tp : tp_max;
  }
  a->wn = wn;
  a->ni = (int)((tp_max + 63) / 64);
  a->pwd = (a->ni + 7) / 8;
  a->slot_bytes = a->ni * 1024;
  if (a->pwd > fz_maxpwd(wn)) return false;
  const int ks = (a->C + 31) / 32;
  const size_t stat = (size_t)((out == OUT_I8 && 8 * 64 * 144 > 2 * wn * 4096) ? 8 * 64 * 144 : 2 * wn * 4096) + 8 * FZ_STAMP_SLOTS * 8;
  a->raw_bytes = 2 * FZ_PAD + (size_t)(ks < FZ_D ? ks : FZ_D) * a->slot_bytes + 1024 + (size_t)a->C * 32;
  return stat + a->raw_bytes <= 160 * 1024;
}

// Fills the launch plan (tile, owq, NQ, the staged-input slot) and says whether the shape is inside the fused path: 3x3,
// stride 1 | 2, dilation 1, C <= 1024, a column tile touching <= 4 images, and min(4, KS) slots of staged input
// (32 channels x the tile's input rows) fitting the LDS left beside the static arrays.
bool fused_dwpw_plan(FusedArgs* a, int kh, int kw, int sh, int sw, int dh, int dw, int out) {
  if (!(kh == 3 && kw == 3 && sh == sw && (sw == 1 || sw == 2) && dh == 1 && dw == 1)) return false;
  if (a->pl > 3 || a->pt > 2 || a->C > FZ_MAXC || a->oh < 1 || a->ow < 1) return false;
  const long total = (long)a->n * a->C * a->h * a->w;
  if (total >= ((long)1 << 31) - 64 || total < 16 || (long)a->oh * a->ow < 16) return false;
  if ((long)a->n * a->oh * ((a->ow + 3) / 4 * 4) >= ((long)1 << 31) - 4096) return false;  // quad / pixel indices stay 32-bit
  if (out == OUT_I8 && (a->ow & 3) == 1) return false;  // int8 epilogue: a compact dword would span three quads (room 1)
  a->owq = (a->ow + 3) / 4;
  a->NQ = (long)a->n * a->oh * a->owq;
  const int M = a->pw.M;
  for (int wn = M <= 64 ? 8 : (M <= 128 ? 4 : (M <= 256 ? 2 : 1)); wn >= 1; wn >>= 1)
    if (fused_plan_tile(a, wn, out)) return true;
  return false;
}

template <int WN, int WM, int OUT>
static void launch_fused_t(FusedArgs a, hipStream_t s) {
  a.pw.NT = (int)((a.NQ + WN * 32 - 1) / (WN * 32));   // blocks along the columns
  a.pw.MT = (a.pw.M + WM * 64 - 1) / (WM * 64);        // blocks along the output channels
  const unsigned blocks = (unsigned)((long)a.pw.MT * ((a.pw.NT + 7) / 8 * 8));
  const size_t lds = a.raw_bytes;
  // the attribute is per DEVICE and launches come from several predictor threads: set it on every launch (a host-side
  // table update; a process-wide "already granted" cache was wrong on a second GPU and racy between threads)
  if (a.stride == 1) {
    auto kfn = fused_dwpw_kernel<WN, WM, OUT, 1>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
  } else {
    auto kfn = fused_dwpw_kernel<WN, WM, OUT, 2>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, s, a);
  }
}

template <int WN, int WM>
static void launch_fused_o(const FusedArgs& a, int out, hipStream_t s) {
  if (out == OUT_I32) launch_fused_t<WN, WM, OUT_I32>(a, s);
  else if (out == OUT_F32) launch_fused_t<WN, WM, OUT_F32>(a, s);
  else launch_fused_t<WN, WM, OUT_I8>(a, s);
}

// `a` must have passed fused_dwpw_plan with the same `out`.
void launch_fused_dwpw(const FusedArgs& a_in, int out, hipStream_t s) {
  FusedArgs a = a_in;
  static int dbg_env = -1;
  if (dbg_env < 0) {
    const char* e = getenv("PLHIP_FUSED_DEBUG");
    dbg_env = e ? atoi(e) : 0;
  }
  a.pw.dbg = dbg_env;
  if (a.wn == 8) launch_fused_o<8, 1>(a, out, s);
  else if (a.wn == 4) launch_fused_o<4, 2>(a, out, s);
  else if (a.wn == 2) launch_fused_o<2, 4>(a, out, s);
  else launch_fused_o<1, 8>(a, out, s);
}

}  // namespace plhip
#else
#include "plhip_kernels.h"
namespace plhip {
bool fused_dwpw_plan(FusedArgs*, int, int, int, int, int, int, int) { return false; }
void launch_fused_dwpw(const FusedArgs&, int, hipStream_t) {}
int debug_read_fz_stamps(void*, size_t) { return -1; }
}  // namespace plhip
#endif
