// conv_patch_i8.hip — dense 3x3 stride-1 int8 convolution as an implicit GEMM on input PATCHES: BASELINE config #2 and
// ResNet50's 3x3 layers.
//
// Replaces the reference's im2col + GEMM route for these layers (lite/kernels/arm/conv_gemmlike.cc:125 ->
// lite/backends/arm/math/conv_impl.cc:490-598 conv_im2col_gemm_int8: im2col_int8 :103-153 + gemm_prepack_int8,
// gemm_prepacked_int8.cc:2582-2744) and its 3x3 specialisation (conv3x3s1_direct_int8.cc); numerics = the direct int32
// accumulator + the fused epilogue of conv_block_utils.h:3185-3225.
//
// Why a new kernel (round-3 evidence, profiles/r03_final_c2bench.txt, r03_final_layer_table_c4.txt): the ring kernel of
// gemm_tr_i8.hip runs these layers at 0.08-0.15 of the int8 MFMA peak.  Its K rows are the (channel, tap) pairs, every one
// fetched from L2 as its own 16-byte pieces: 9x the input per tile through the CU's address path (~21-36 B/clk in that
// pattern) for 4.6 k cycles of MFMA per 256 x 128 tile, plus the weight panel again for every tile.
// Here the K loop is re-ordered as (32-channel chunk, column shift s, tap row r):
//   * on a zero-padded copy whose row pitch PWp is a multiple of 8, output pixel p = oh * PWp + ow (the ow >= OW columns are
//     computed and dropped) finds tap (r, s) of channel c at padded[c][p + r * PWp + s]: a SHIFT of the flattened plane.
//     A tile of NT pixels therefore needs, per channel, ONE contiguous run of NT + 2 PWp bytes for all three tap rows, and
//     one copy of it per column shift s (each starts s bytes later, so that every 8-byte group the transposing LDS read
//     fetches is aligned): 3 copies instead of 9 K rows, each ~1.5x the tile's pixels: ~4.5 bytes of L2 -> LDS traffic per
//     pixel and channel instead of 9, in runs of 350-600 contiguous bytes (the fast LDS-DMA pattern, tools/ingest_bench.hip);
//   * a SLAB = (32 channels) x (that run) of one shift s; per slab a wave issues 3 tap rows x 7 n tiles = 21 MFMAs
//     (v_mfma_i32_32x32x32_i8, activations = A operand through ds_read_b64_tr_b8, no VALU in the loop): the tap row is an
//     LDS address offset r * PWp, the n tile an immediate;
//   * slabs move through a ring of 3-4 slots by LDS-DMA, one barrier per slab (672 cycles of MFMA per wave);
//   * the channel-row pitch of a slab is an ODD multiple of 32 bytes: the 8 rows x 2 chunks a half-wave's transposed read
//     touches fall into 16 distinct 16-byte bank slots;
//   * C = 64 (config #2, ResNet50's res2): the whole weight panel of a wave (32 rows x 576 = 72 VGPRs) stays in
//     registers for all the tiles the block works through; deeper layers: the slab pair carries the 3 weight fragments
//     per m tile of its (chunk, s) through the ring (both halves of the block use the same ones);
//   * the two halves of a block (waves 0-3 / 4-7, SIMD partners) each own a stream of pixel tiles;
//   * int8 epilogue: requantise (doubled-value trick, gemm_epilogue.h), two v_permlane32_swap -> 16 consecutive pixels
//     per lane, written into a wave-private LDS image at their COMPACTED position (the dropped columns disappear there),
//     read back as 16-byte pieces of whole channel rows: a store instruction writes 8 rows x up to 128 contiguous bytes.
// Weights: [m tile][chunk][s][r][64 lanes][16 B] (launch_pack_conv_patch), lane (m % 32, h) holds channels 16h .. 16h+15.
#include "conv_patch_kernel.h"

namespace plhip {

__device__ unsigned long long g_patch_stamps[512 * 8 * PATCH_STAMP_SLOTS];
static unsigned long long* patch_stamps_ptr() {
  static unsigned long long* p = nullptr;
  if (!p) (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_patch_stamps));
  return p;
}
int debug_read_patch_stamps(void* dst, size_t bytes) {
  const size_t cap = sizeof(unsigned long long) * 512 * 8 * PATCH_STAMP_SLOTS;
  if (bytes > cap) bytes = cap;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_patch_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

static int patch_env() {  // PLHIP_CONV_PATCH=0: the ring-kernel implicit GEMM instead (A/B runs)
  const int v = knob("CONV_PATCH", 1);
  return v;
}

// Row pitch of the padded copy: a multiple of 8 >= w + pl + pr.  The right padding may also be the NEXT row's left padding
// (pr <= pl: column w + pl + j of a row is column j of the next one, zero for j < pl) — a 7-wide plane with pad 1 fits 8.
int conv_patch_row_pitch(int w, int pl, int pr) {
  const int need = pr <= pl ? w + pl : w + pl + pr;
  return (need + 7) & ~7;
}

bool conv_patch_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int w, int pl, int pr) {
  if (!patch_env()) return false;
  if (kh != 3 || kw != 3 || sh != 1 || sw != 1 || dh != 1 || dw != 1 || groups != 1) return false;
  if (cin % 32 != 0 || cin < 64 || cout < 32) return false;
  const int pwp = conv_patch_row_pitch(w, pl, pr);
  if (pwp > 64 || pwp < 8) return false;
  if (cout <= 64 && cin != 64) return false;  // the 2 x 2 wave layout exists for the register-resident weights only
  if (cout <= 64 && conv_patch_global(pwp)) return false;  // ... and not in global mode (no registers for its second epilogue body)
  return true;
}

// planes smaller than a tile (7-wide: 72 padded pixels): global mode, the padded copy is channel-major
bool conv_patch_global(int pwp) { return pwp < 16; }

size_t conv_patch_packed_bytes(int cin, int cout) { return (size_t)((cout + 31) / 32) * (cin / 32) * 9 * 1024; }

__global__ void pack_conv_patch_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, int cin, int cout, size_t total) {
  const int NCH = cin / 32;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 15;
    const int lane = (idx >> 4) & 63;
    size_t t = idx >> 10;
    const int r = t % 3;
    t /= 3;
    const int s = t % 3;
    t /= 3;
    const int ch = t % NCH;
    const int mt = (int)(t / NCH);
    const int m = mt * 32 + (lane & 31);
    const int c = ch * 32 + 16 * (lane >> 5) + j;
    wp[idx] = m < cout ? w[(((size_t)m * cin + c) * 3 + r) * 3 + s] : (int8_t)0;
  }
}

void launch_pack_conv_patch(const int8_t* w_oihw, int8_t* wp, int cin, int cout, hipStream_t s) {
  const size_t total = conv_patch_packed_bytes(cin, cout);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_patch_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, wp, cin, cout, total);
}

// Zero-padded copy for this route: xp[plane][ph][pw] = x[plane][ph - pt][pw - pl] or 0, rows of pw (a multiple of 8) bytes.
// One thread = 16 aligned output bytes = two 8-byte halves, each inside one padded row: ONE unaligned 8-byte load from a
// start clamped into the source row, shifted into place (the shift brings the zeros of the left / right border in), one
// 16-byte store.  (The dword-per-thread copy of the ring route, pad_input_i8_kernel, ran this shape at 0.6 TB/s: 12.8 us
// for config #2's 7.6 MB — 4-byte loads and stores; profiles/r03_patch_v1_kernel_stats.csv.)
__global__ __launch_bounds__(256) void pad_rows8_i8_kernel(PadArgs a) {
  const long nq = a.total >> 4;
  const uint32_t plane_sz = (uint32_t)a.ph * (uint32_t)a.pw;
  // workgroups are dealt round-robin over the 8 XCDs: block b writes the (b % 8)-th eighth of the buffer, i.e. XCD x the
  // images whose tiles conv_patch_i8_kernel gives to XCD x (its L2 then holds what that kernel's blocks read first)
  const long nbx = (gridDim.x + 7) >> 3;
  const long vb = (long)(blockIdx.x & 7) * nbx + (blockIdx.x >> 3);
  for (long q = vb * blockDim.x + threadIdx.x; q < nq; q += 8 * nbx * blockDim.x) {
    unsigned long long out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t o = ((uint32_t)q << 4) + 8u * e;
      uint32_t plane = fastdiv_u31(o, a.div_plane_m, a.div_plane_s);
      const uint32_t rem = o - plane * plane_sz;
      if (a.tb > 0 && (int)plane < a.planes) {  // channel-major copy: output plane (c, b) <- input plane (b, c)
        const uint32_t c_ = plane / (uint32_t)a.tb, b_ = plane - c_ * (uint32_t)a.tb;
        plane = b_ * (uint32_t)a.tc + c_;
      }
      const int ph = (int)fastdiv_u31(rem, a.div_pw_m, a.div_pw_s), pc = (int)rem - ph * a.pw;
      const int ih = ph - a.pt, iw0 = pc - a.pl;
      unsigned long long v = 0;
      if ((int)plane < a.planes && ih >= 0 && ih < a.h && iw0 < a.w && iw0 + 8 > 0) {
        const int8_t* row = a.x + ((size_t)plane * a.h + ih) * a.w;
        if (a.w >= 8) {
          const int st = iw0 < 0 ? 0 : (iw0 + 8 > a.w ? a.w - 8 : iw0);
          __builtin_memcpy(&v, row + st, 8);
          const int sh = st - iw0;  // > 0: the data starts `sh` columns late (left border); < 0: early (right border)
          v = sh >= 0 ? v << (8 * sh) : v >> (8 * -sh);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (iw0 + k >= 0 && iw0 + k < a.w) v |= (unsigned long long)(uint8_t)row[iw0 + k] << (8 * k);
        }
      }
      out[e] = v;
    }
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 o2 = {out[0], out[1]};
    reinterpret_cast<v2u64*>(a.xp)[q] = o2;
  }
}

void launch_pad_rows8(PadArgs a, hipStream_t s) {  // a.pw % 8 == 0, a.total % 16 == 0, a.xp 16-byte aligned
  unsigned m;
  int sh;
  auto magic = [&](long d) {
    int l = 0;
    while ((1L << l) < d) ++l;
    if ((1L << l) == d) {
      m = 0;
      sh = l;
    } else {
      m = (unsigned)(((1ULL << (31 + l)) / (unsigned long long)d) + 1ULL);
      sh = l - 1;
    }
  };
  magic((long)a.ph * a.pw);
  a.div_plane_m = m; a.div_plane_s = sh;
  magic(a.pw);
  a.div_pw_m = m; a.div_pw_s = sh;
  long blocks = ((a.total >> 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  blocks = (blocks + 7) & ~7L;  // whole rounds over the 8 XCDs (the kernel's block map)
  hipLaunchKernelGGL(pad_rows8_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// ---- 3x3 STRIDE 2 over phase planes (ResNet50's three downsampling 3x3 convs).
// With xpad = the zero-padded input, phase plane P_ab[i][j] = xpad[2i + a][2j + b]:
//   out[oh][ow] = sum w[r][s] xpad[2 oh + r][2 ow + s] = sum w[r][s] P_{r&1, s&1}[oh + (r >> 1)][ow + (s >> 1)],
// i.e. on phase plane (0,0) a 2x2 stride-1 conv, on (0,1) a 2x1, on (1,0) a 1x2 and on (1,1) a 1x1 one: the patch kernel's slab
// walk with 6 slabs per 32 channels (plane, column shift) of 2 / 1 tap rows (conv_patch_kernel.h, patch_s2_*) instead of 3
// slabs of 3.  (The dense 2x2 form over 4 Cin "channels" (c, a, b) multiplies 7 zero taps in 16 and measured no faster than
// the ring kernel.)  Row pitch of a phase plane: a multiple of 8 >= ceil((W + pads) / 2); the copy's planes are ordered
// [image][group of 32 channels][phase 2a + b][channel % 32]: a "chunk" of the kernel = the 32 channels of one phase.
int conv_patch_s2_row_pitch(int w, int pl, int pr) { return (((w + pl + pr + 1) >> 1) + 7) & ~7; }
bool conv_patch_s2_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int w, int pl, int pr) {
  if (!patch_env()) return false;
  const int s2_env = knob("CONV_PATCH_S2", 1);  // 0 = the ring kernel's stride-2 implicit GEMM (A/B runs)
  if (!s2_env) return false;
  if (kh != 3 || kw != 3 || sh != 2 || sw != 2 || dh != 1 || dw != 1 || groups != 1) return false;
  if (cin % 32 != 0 || cout <= 64) return false;  // whole 32-channel groups; the 4 x 1 wave layout
  const int pwp = conv_patch_s2_row_pitch(w, pl, pr);
  return pwp <= 64 && pwp >= 8;
}
size_t conv_patch_s2_packed_bytes(int cin, int cout) { return (size_t)((cout + 31) / 32) * (cin / 32) * 9 * 1024; }

// weights [m tile][group of 32 channels][9 fragments in step order][64 lanes][16 B]: lane (m % 32, h) holds channels
// 16h .. 16h + 15 of the group; fragment f = patch_s2_wfrag(ls) + r' <-> tap (2 r' + a, 2 s' + b) of step ls
__global__ void pack_conv_patch_s2_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, int cin, int cout, size_t total) {
  const int NG = cin / 32;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 15;
    const int lane = (idx >> 4) & 63;
    size_t t = idx >> 10;
    const int f = t % 9;
    t /= 9;
    const int g = t % NG;
    const int mt = (int)(t / NG);
    int ls = 0;
    while (ls < 5 && patch_s2_wfrag(ls + 1) <= f) ++ls;
    const int r1 = f - patch_s2_wfrag(ls), ph = patch_s2_phase(ls);
    const int r = 2 * r1 + (ph >> 1), sx = 2 * patch_s2_shift(ls) + (ph & 1);
    const int m = mt * 32 + (lane & 31);
    const int c = g * 32 + 16 * (lane >> 5) + j;
    wp[idx] = m < cout ? w[(((size_t)m * cin + c) * 3 + r) * 3 + sx] : (int8_t)0;
  }
}
void launch_pack_conv_patch_s2(const int8_t* w_oihw, int8_t* wp, int cin, int cout, hipStream_t s) {
  const size_t total = conv_patch_s2_packed_bytes(cin, cout);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_patch_s2_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, wp, cin, cout, total);
}

// phase-split zero-padded copy: output plane (b, g, phase 2a + b', c32) [ph][pw] = x[b][32 g + c32][2 ph + a - pt][2 pw + b' - pl]
// or 0; a.planes = B * 4 C output planes, a.tc = 4 C, a.ph / a.pw the phase plane's dims (pw % 8 == 0).  One thread = 16 output
// bytes = two 8-byte halves, each inside one output row.  A half = every other byte of a 15-byte source span: ONE unaligned
// 16-byte fetch at the span's address wherever that lies inside the input buffer — also when the span starts left of the row
// or ends right of it (the 14-wide planes: every span does): the bytes of the neighbouring rows it brings along are masked
// off — and byte loads only at the two ends of the buffer.  (First version: byte loads for every span that crossed a row
// border, half of all spans at 56 x 56 and all of them at 14 x 14: 0.5-1.5 TB/s.)
// a.tb > 0: channel-major output (global mode), as pad_rows8_i8_kernel; div_pwq: the divisor a.tb.
__global__ __launch_bounds__(256) void pad_phase8_i8_kernel(PadArgs a) {
  const long nq = a.total >> 4;
  const uint32_t plane_sz = (uint32_t)a.ph * (uint32_t)a.pw;
  const long xbytes = (long)(a.planes >> 2) * a.h * a.w;
  const long nbx = (gridDim.x + 7) >> 3;
  const long vb = (long)(blockIdx.x & 7) * nbx + (blockIdx.x >> 3);
  for (long q = vb * blockDim.x + threadIdx.x; q < nq; q += 8 * nbx * blockDim.x) {
    unsigned long long out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t o = ((uint32_t)q << 4) + 8u * e;
      const uint32_t plane = fastdiv_u31(o, a.div_plane_m, a.div_plane_s);
      const uint32_t rem = o - plane * plane_sz;
      const bool inside = (int)plane < a.planes;
      uint32_t b_, ce;  // image, (group, phase, channel % 32)
      if (a.tb > 0) {
        ce = fastdiv_u31(plane, a.div_pwq_m, a.div_pwq_s);
        b_ = plane - ce * (uint32_t)a.tb;
      } else {
        b_ = fastdiv_u31(plane, a.div_ph_m, a.div_ph_s);
        ce = plane - b_ * (uint32_t)a.tc;
      }
      const uint32_t src_plane = b_ * ((uint32_t)a.tc >> 2) + (ce >> 7) * 32u + (ce & 31u);
      const int pa = (ce >> 6) & 1, pb = (ce >> 5) & 1;
      const int ph = (int)fastdiv_u31(rem, a.div_pw_m, a.div_pw_s), pc = (int)rem - ph * a.pw;
      const int ih = 2 * ph + pa - a.pt, iw0 = 2 * pc + pb - a.pl;  // source row; source column of output byte 0 (then + 2 per byte)
      unsigned long long v = 0;
      if (inside && ih >= 0 && ih < a.h && iw0 < a.w && iw0 + 15 > 0) {
        const long off = ((long)src_plane * a.h + ih) * a.w + iw0;
        if (off >= 0 && off + 16 <= xbytes) {
          uint32_t dd[4];
          __builtin_memcpy(dd, a.x + off, 16);
          v = (unsigned long long)__builtin_amdgcn_perm(dd[1], dd[0], 0x06040200u) |
              ((unsigned long long)__builtin_amdgcn_perm(dd[3], dd[2], 0x06040200u) << 32);
          // output bytes [k0, k1) come from columns inside the row
          const int k0 = iw0 < 0 ? (1 - iw0) >> 1 : 0;
          const int k1 = (a.w - iw0 + 1) >> 1;  // >= 1
          if (k0 > 0) v &= ~0ull << (8 * k0);
          if (k1 < 8) v &= ~(~0ull << (8 * k1));
        } else {
          const int8_t* row = a.x + (off - iw0);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int iw = iw0 + 2 * k;
            if (iw >= 0 && iw < a.w) v |= (unsigned long long)(uint8_t)row[iw] << (8 * k);
          }
        }
      }
      out[e] = v;
    }
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 o2 = {out[0], out[1]};
    reinterpret_cast<v2u64*>(a.xp)[q] = o2;
  }
}
void launch_pad_phase8(PadArgs a, hipStream_t s) {  // a.pw % 8 == 0, a.total % 16 == 0, a.xp 16-byte aligned, a.tc = 4 C
  unsigned m;
  int sh;
  auto magic = [&](long d) {
    int l = 0;
    while ((1L << l) < d) ++l;
    if ((1L << l) == d) {
      m = 0;
      sh = l;
    } else {
      m = (unsigned)(((1ULL << (31 + l)) / (unsigned long long)d) + 1ULL);
      sh = l - 1;
    }
  };
  magic((long)a.ph * a.pw);
  a.div_plane_m = m; a.div_plane_s = sh;
  magic(a.pw);
  a.div_pw_m = m; a.div_pw_s = sh;
  magic(a.tc);
  a.div_ph_m = m; a.div_ph_s = sh;
  magic(a.tb > 0 ? a.tb : 1);
  a.div_pwq_m = m; a.div_pwq_s = sh;
  long blocks = ((a.total >> 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  blocks = (blocks + 7) & ~7L;
  hipLaunchKernelGGL(pad_phase8_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// fastdiv_u31's (magic, shift) for divisor d (dw_common.h); general_pow2: the multiply form for powers of two as well
// (d >= 2: magic 2^31 + 1, shift l - 1), for device code that must not branch on the marker 0
static inline void magic_u31(long d, unsigned& m, int& sh, bool general_pow2 = false) {
  int l = 0;
  while ((1L << l) < d) ++l;
  if ((1L << l) == d && !(general_pow2 && d >= 2)) {
    m = 0;
    sh = l;
    return;
  }
  m = (unsigned)(((1ULL << (31 + l)) / (unsigned long long)d) + 1ULL);
  sh = l - 1;
}

void launch_conv_patch(PatchArgs a, int out, hipStream_t s) {
      // its OWN variable (diagnostics only): 1 = no epilogue (timing experiments), 32 = timeline stamps.  (PLHIP_GEMM_DEBUG
    // also re-routes the GEMM kernels of the other layers, e.g. 7-wide implicit-GEMM rows onto a kernel that cannot run them.)
  const int dbg_env = knob("PATCH_DEBUG", 0);
  a.dbg = dbg_env;
  a.stamps = (a.dbg & 32) ? patch_stamps_ptr() : nullptr;
  a.NCH = a.C / 32;
  a.HWY = a.OH * a.OW;
  a.y_bstride = (size_t)a.M * a.HWY;
  const bool stat = a.NCH == 2 && !a.s2;
  const bool layout_b = a.M <= 64 && !a.s2;       // 2 m tiles x 2 pixel groups per half (else 4 x 1)
  const int NTH = (layout_b ? 2 : 1) * PATCH_NTW * 32;
  const int WMH = layout_b ? 2 : 4;
  a.glob = conv_patch_global(a.PWp) ? 1 : 0;
  a.IMGP = a.PLANE;
  a.nimg = a.B;
  if (a.glob) {  // p runs over all images of a channel: one "image" of B * IMGP pixels, PLANE = the channel stride
    const long PT = (long)a.B * a.IMGP;
    a.PLANE = (int)PT;
    a.B = 1;
    a.TPI = (int)((PT + NTH - 1) / NTH);
  } else {
    const long P = (long)a.OH * a.PWp;
    a.TPI = (int)((P + NTH - 1) / NTH);
  }
  a.T = a.B * a.TPI;
  magic_u31(a.IMGP, a.imgp_m, a.imgp_s, true);
  magic_u31(a.HWY > 1 ? a.HWY : 2, a.hwy_m, a.hwy_s, true);
  const int KT = a.s2 ? 2 : 3;  // tap rows a slab is read at (stride 2: phase planes, at most 2)
  int pitch = (NTH + (KT - 1) * a.PWp + 31) & ~31;
  if (((pitch >> 5) & 1) == 0) pitch += 32;
  a.pitch = pitch;
  a.pps = pitch >> 5;
  a.MB = (a.M + 32 * WMH - 1) / (32 * WMH);
  // blocks: 8 XCDs x MB x NQ, each with NH tile streams; the register-resident variants run 4-wave blocks (NH = 1), two per
  // CU: 512 blocks fill the chip, the ring variant one 8-wave block (NH = 2) per CU
  const int NH = stat ? 1 : 2;
  a.T8 = (a.T + 7) / 8;                                    // XCD x works through the tiles [x T8, (x + 1) T8)
  const int nq_max = (64 / NH) / a.MB > 0 ? (64 / NH) / a.MB : 1;
  int nq = (a.T8 + NH - 1) / NH;
  nq = nq < 1 ? 1 : (nq > nq_max ? nq_max : nq);
  a.rounds = (a.T8 + NH * nq - 1) / (NH * nq);
  nq = (a.T8 + NH * a.rounds - 1) / (NH * a.rounds);  // the fewest blocks that need no more rounds
  a.NQ = nq;
  const int delay_env = knob("PATCH_DELAY", 0);  // s_sleep units (64 clocks) the second block of a CU starts late
  a.delay = delay_env;
  magic_u31(a.PWp, a.pw_m, a.pw_s, true);
  magic_u31(a.TPI, a.tpi_m, a.tpi_s);
  magic_u31(a.pitch, a.pitch_m, a.pitch_s);
  if (a.s2) launch_patch_s2(a, out, s);
  else if (layout_b) launch_patch_stat_b(a, out, s);
  else if (stat) launch_patch_stat_a(a, out, s);
  else launch_patch_stream_a(a, out, s);
}

void launch_patch_stat_a(const PatchArgs& a, int out, hipStream_t s) { launch_patch_o<1, 4, 1, 3, 4, true>(a, out, s); }

}  // namespace plhip
