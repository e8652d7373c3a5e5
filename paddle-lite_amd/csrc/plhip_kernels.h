// plhip_kernels.h — argument blocks and host launchers of the gfx950 kernels (internal to libplhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace plhip {

// A/B and timing knobs of the launchers: name -> value set through plhip_debug_set (include/plhip.h), else the
// default.  The library never reads the environment.
int knob(const char* name, int dflt);

struct GemmArgs {
  const int8_t* wp;    // packed weights of this group: [MT32][KS][64][16]
  const int8_t* x;     // B operand base for this group: row k of image b at x + b*x_bstride + k*HWX
  void* y;             // output base for this group: row m of image b at y + (b*y_bstride + m*HWY) elements
  const float* scale;  // [M] folded per-channel scale (unused for I32)
  const float* bias;   // [M] folded bias or nullptr
  int M, K, KS;        // rows, reduction length, K-steps of 32
  int HWX, HWY;        // columns per image in n-space (multiple of 4) / valid columns per image in y (= y row pitch)
  int XP;              // x row pitch in bytes (HWX for the im2col buffer, HW for a dense NCHW slab)
  long x_bytes;        // bytes readable from x (guards the last partial dword when XP % 4 != 0)
  int NB;              // images
  size_t x_bstride, y_bstride;
  int MT, NT;          // wave tiles along M (32*MA rows) and N (128 columns)
  int act;
  float alpha;
  int dbg;             // PLHIP_GEMM_DEBUG (timing experiments only): 1 = skip the epilogue, 2 = skip the K loop
  // fused graph tail of an fp32-output conv (OUT_F32 only; all optional, zero = plain conv):
  //   v = act(fma(acc, s, b));  if (res) v = v + res[same offset];  if (res_relu) v = max(v, 0);
  //   if (y) y = v;  if (y2) y2 = round_sat_i8(v * inv_scale2)          (calib, type_trans.cc:45,183-184)
  // i.e. conv2d[fp32_out] -> elementwise_add / fusion_elementwise_add_activation -> calib of the reference program
  // in one launch, every value rounded exactly as the three instructions round it.  y may be nullptr when y2 is set.
  const float* res;
  int res_relu;
  int8_t* y2;
  float inv_scale2;
  // implicit GEMM (dense kh x kw, stride 1, dilation 1) on a zero-PADDED copy of the input [b][c][PH][PW]: im_kw > 0.
  // Then an "image" of the column space is one output row (NB = batch * OH, HWX = OW) and K-row k = (c, r, s) of it
  // starts at  x + ((b*C + c)*PH + oh + r)*PW + s : contiguous bytes, so the B tile still moves as 16-byte pieces.
  int im_kw, im_khkw, im_c, im_ph, im_pw, im_oh;
  // stride of the implicit conv (1 or 2; transposed-read kernel only).  Stride 2: the padded copy is PHASE-SPLIT, plane
  // (c, p, q) holds padded[c][2y + p][2x + q] (im_ph x im_pw each), so that tap (r, s) of output (oh, ow) is byte
  // ((c*4 + (r&1)*2 + (s&1)) * im_ph + oh + (r>>1)) * im_pw + ow + (s>>1): contiguous in ow again.
  int im_s;
  // wide-tile kernel (gemm_wide_i8.hip): fastdiv_u31's (magic, shift) for the chunks per image, set by its launcher
  unsigned cpi_m;
  int cpi_s;
  unsigned long long* stamps;  // its diagnostic timeline buffer (PLHIP_GEMM_DEBUG & 32) or nullptr
};

struct PadArgs {
  const int8_t* x;  // [planes][h][w]
  int8_t* xp;       // [planes][ph][pw] + slack, zero border; stride 2: [planes][2][2][ph][pw] phase planes
  int planes, h, w, ph, pw, pt, pl;
  int stride;       // 1, or 2 = phase-split copy (ph, pw are then the phase plane's dims)
  int tb, tc;       // pad_rows8 only, tb > 0: CHANNEL-major output: plane c * tb + b of xp <- plane b * tc + c of x
  long total;       // bytes of xp to write (multiple of 4: planes*ph*pw rounded up + slack)
  // exact division of a 31-bit index by ph*pw and by pw without a divide sequence (launch_pad_input fills them; same
  // (magic, shift) form as DwArgs: magic == 0 -> power of two)
  unsigned div_plane_m, div_pw_m, div_pwq_m, div_ph_m;  // pwq = pw / 4 and ph: the phase-split copy's divisors
  int div_plane_s, div_pw_s, div_pwq_s, div_ph_s;
};
void launch_pad_input(const PadArgs& a, hipStream_t s);
void launch_pad_rows8(PadArgs a, hipStream_t s);  // conv_patch_i8.hip: rows of pw % 8 == 0 bytes, 16 bytes per thread

// dense 3x3 stride-1 convolution on input patches (conv_patch_i8.hip)
struct PatchArgs {
  const int8_t* xp;    // zero-padded copy [B][C][PH][PWp] (+ slack): padded[ih + pt][iw + pl] = x[ih][iw]
  const int8_t* wp;    // packed weights [MT32][NCH][3 s][3 r][64 lanes][16 B] (launch_pack_conv_patch)
  void* y;             // [B][M][OH][OW] int8 / fp32 / int32 (may be nullptr with y2)
  const float* scale;  // [M] folded per-channel scale (unused for I32)
  const float* bias;   // [M] or nullptr
  int B, C, M, OH, OW;
  int PWp, PLANE;      // row pitch (multiple of 8) and plane size PH * PWp of the padded copy
  int NCH;             // 32-channel chunks: C / 32
  int pitch, pps;      // LDS bytes per channel row of a slab (odd multiple of 32, >= tile + 2 PWp), pitch / 32
  int TPI, T, T8;      // tiles per image, tiles in all, tiles per XCD (ceil(T / 8))
  int MB, NQ, rounds;  // M blocks, blocks per XCD and M block, tiles per stream
  int HWY;             // OH * OW
  size_t y_bstride;    // M * OH * OW
  int act;
  float alpha;
  unsigned pw_m, tpi_m, pitch_m;  // fastdiv_u31 (magic, shift) for PWp, TPI, pitch
  int pw_s, tpi_s, pitch_s;
  // fused tail of an fp32-output conv (OUT_F32 only), as GemmArgs
  const float* res;
  int res_relu;
  int8_t* y2;
  float inv_scale2;
  int dbg;
  // global mode (planes smaller than a tile: 7-wide): the padded copy is [c][image][PH][PWp], p runs over all images
  int s2;              // 3x3 stride 2 as a 2x2 conv over phase planes: C = 4 Cin, the slabs are walked with 2 taps per side
  int glob, nimg, IMGP;  // nimg = the real image count; IMGP = PH * PWp pixels per image (PLANE is then the CHANNEL stride B * IMGP, B = 1, TPI = T)
  unsigned imgp_m, hwy_m;
  int imgp_s, hwy_s;
  int delay;           // NH = 1: s_sleep units the second block of a CU starts late (experiments)
  unsigned long long* stamps;
};
// row pitch of the padded copy for (w, pl, pr), 0 = outside the route
int conv_patch_row_pitch(int w, int pl, int pr);
bool conv_patch_global(int pwp);  // planes smaller than a tile: channel-major padded copy, p across images
bool conv_patch_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int w, int pl, int pr);
size_t conv_patch_packed_bytes(int cin, int cout);
void launch_pack_conv_patch(const int8_t* w_oihw, int8_t* wp, int cin, int cout, hipStream_t s);
// fills the launch plan of `a` (B, C, M, OH, OW, PWp, PLANE set by the caller) and launches
void launch_conv_patch(PatchArgs a, int out, hipStream_t s);
void launch_patch_stat_a(const PatchArgs& a, int out, hipStream_t s);    // per-variant translation units
void launch_patch_stat_b(const PatchArgs& a, int out, hipStream_t s);
void launch_patch_stream_a(const PatchArgs& a, int out, hipStream_t s);
void launch_patch_s2(const PatchArgs& a, int out, hipStream_t s);
int conv_patch_s2_row_pitch(int w, int pl, int pr);
bool conv_patch_s2_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int w, int pl, int pr);
size_t conv_patch_s2_packed_bytes(int cin, int cout);
void launch_pack_conv_patch_s2(const int8_t* w_oihw, int8_t* wp, int cin, int cout, hipStream_t s);
void launch_pad_phase8(PadArgs a, hipStream_t s);
int debug_read_patch_stamps(void* dst, size_t bytes);

struct Im2colArgs {
  const int8_t* x;
  int8_t* col;
  int cin, cin_g, h, w, kh, kw, pt, pl, sh, sw, dh, dw, oh, ow;
  int G, Kg, N, Np;
  size_t rows;  // B*G*Kg
};

struct DwArgs {
  const int8_t* x;
  const int8_t* wt;  // [C][kh*kw]
  void* y;
  const float* scale;
  const float* bias;
  int planes, C, h, w, oh, ow, kh, kw, pt, pl, sh, sw, dh, dw;
  int PB;       // planes per block
  int OB;       // output rows per block band
  int bands;    // bands per plane
  int in_rows;  // staged input rows per band
  int pitch;    // LDS row pitch in bytes (multiple of 4)
  long total_lanes;               // direct kernel: planes * strips * quads
  int owq_log2, spp_log2, fast_div;  // direct kernel: power-of-two index split
  // exact division of a 31-bit index by a run-time constant without a divide: q = d is a power of two ? n >> sh
  // : mulhi(n, magic) >> sh  (magic = floor(2^(31+s)/d) + 1, s = ceil(log2 d), sh = s - 1; exact for n < 2^31)
  unsigned div_owq_m, div_spp_m, div_c_m;
  int div_owq_s, div_spp_s, div_c_s;
  int stage_bytes;                // direct kernel: LDS bytes per wave for output staging (0 = off)
  int lw, nblocks;                // direct kernel: lanes of a wave that own work (staging: whole strips), workgroups of work
  int act;
  float alpha;
  // direct kernels, set by their launcher: the int8 clamp's upper bound on doubled values (relu6: min(2 alpha, 254)) and the
  // byte-wise +1 of the packed rounding, as kernel arguments = scalar operands (the compiler re-made both per output row)
  float hi2 = 254.f;
  unsigned ones = 0x01010101u;
};

// fused depthwise 3x3 (int8 out) -> pointwise 1x1 (fused_dwpw_i8.hip)
struct FusedArgs {
  const int8_t* x;        // [n, C, h, w]
  const int8_t* dw_w;     // [C, 1, 3, 3]
  const float* dw_scale;  // folded depthwise scale / bias (int8-out folding), per channel
  const float* dw_bias;   // or nullptr
  int dw_act;
  float dw_alpha;
  int n, C, h, w, oh, ow, pt, pl, stride;
  int tiles;              // launch plan (fused_dwpw_plan): (image, half-plane) tiles = 2 n
  unsigned ones;          // 0x01010101 as a scalar operand (the byte-wise +1 of the packed rounding)
  int stream;             // 1 = the streaming kernel of the large planes (fused_dwpw_stream.hip), 0 = the 14 x 14 kernel
  GemmArgs pw;            // wp, y, scale, bias, M, KS, HWY (= oh*ow), y_bstride, act, alpha
};
// fills the plan from (n, C, h, w, oh, ow, pt, pl, stride, pw.M); false = shape outside the fused path
bool fused_dwpw_plan(FusedArgs* a, int kh, int kw, int sh, int sw, int dh, int dw, int out);
void launch_fused_dwpw(const FusedArgs& a, int out, hipStream_t s);
bool fused_stream_supported(const FusedArgs& a);   // fused_dwpw_stream.hip: the 112 / 56 / 28-wide stride-1 pairs
void launch_fused_stream(const FusedArgs& a, int out, hipStream_t s);
bool fused_small_supported(const FusedArgs& a);    // fused_dwpw_small.hip: the 7 x 7 planes (512 -> 1024 stride 2, 1024 -> 1024)
void launch_fused_small(const FusedArgs& a, int out, hipStream_t s);
void debug_set_fused(int v);                        // bit 5 (32): timeline stamps
int debug_read_fw_stamps(void* dst, size_t bytes);
int debug_read_fs_stamps(void* dst, size_t bytes);  // the streaming kernel's
int debug_read_f7_stamps(void* dst, size_t bytes);  // the small-plane kernel's

int launch_gemm_i8(const GemmArgs& g, int ma, int out, bool vec_store, bool aligned_loads, hipStream_t s);  // 0 or -3
// second-generation ring kernel (gemm_tr_i8.hip); false = shape outside it, the caller falls back
bool launch_gemm_tr(const GemmArgs& g, int out, hipStream_t s);
int gemm_tr_enabled();
// third-generation kernel: one wide tile per CU (gemm_wide_i8.hip); false = shape outside it, the caller falls back
bool launch_gemm_wide(const GemmArgs& g, int out, hipStream_t s);
int gemm_wide_ntt(const GemmArgs& g);  // n tiles per block it would use, 0 = not taken
int debug_read_wide_stamps(void* dst, size_t bytes);
void debug_set_wide_ntt(int v);
void launch_wide_n4(const GemmArgs& g, int out, hipStream_t s);  // per-tile translation units (gemm_wide_n*.hip)
void launch_wide_n7(const GemmArgs& g, int out, hipStream_t s);
void launch_wide_n8(const GemmArgs& g, int out, hipStream_t s);  // 4 / 7 / 8 force that tile, 0 = automatic, -1 = back to the environment's choice
void launch_pack_weights(const int8_t* w, int8_t* wp, int G, int Mg, int Kg, int MT32, int KS, hipStream_t s);
void launch_im2col(const Im2colArgs& a, hipStream_t s);
int launch_depthwise(const DwArgs& a, int out, hipStream_t s);  // returns 0 or -3 (unsupported LDS size)

// direct 3x3 stride-2 convolution for small Cin (network stems)
struct DirectS2Args {
  const int8_t* x;
  const uint32_t* wp;  // packed [cin*3 + r][coutp] dwords (w0, w1, w2, 0)
  void* y;
  const float* scale;
  const float* bias;
  int n, cin, h, w, cout, coutp, oh, ow, pt, pl;
  int act;
  float alpha;
  // fused tail of an fp32-output conv (the 7x7 stem only), as GemmArgs
  const float* res = nullptr;
  int res_relu = 0;
  int8_t* y2 = nullptr;
  float inv_scale2 = 0.f;
  // fused calib[fp32_to_int8] in front (conv_stem_f32in.hip): the fp32 image and 1 / its quantisation scale
  const float* xf = nullptr;
  float x_inv_scale = 0.f;
};
// conv_stem7_i8.hip: 7x7 stride 2, Cin <= 3 (ResNet50's stem); wp = its A fragments
bool conv7x7s2_stem_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int n, int h, int w,
                              int oh, int ow, int pl);
size_t conv7x7s2_stem_packed_bytes(int cout);
void launch_pack_conv7x7s2_stem(const int8_t* w_oihw, int8_t* wp, int cin, int cout, hipStream_t s);
void launch_conv7x7s2_stem(const DirectS2Args& a, int out, bool vec_store, hipStream_t s);
bool conv3x3s2_direct_supported(int cin, int cout, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int pl);
size_t conv3x3s2_direct_packed_bytes(int cin, int cout);
void launch_pack_conv3x3s2_direct(const int8_t* w_oihw, uint32_t* wp, int cin, int cout, hipStream_t s);
void launch_conv3x3s2_direct(const DirectS2Args& a, int out, hipStream_t s);
size_t conv3x3s2_dot4_bytes(int cin, int cout);              // offset of the MFMA A fragments inside the packed block
bool conv3x3s2_f32in_supported(const DirectS2Args& a);      // conv_stem_f32in.hip: calib[fp32_to_int8] + this conv in one launch
void launch_conv3x3s2_f32in(const DirectS2Args& a, const int8_t* afrag, int out, hipStream_t s);

size_t fc_packed_bytes(int k, int n);
void launch_pack_fc(const int8_t* w_kn, int8_t* wp, int k, int n, hipStream_t s);
void launch_fc(const int8_t* x, const int8_t* wp, const float* scale, const float* bias, void* y, int m, int k, int n,
               int relu, int out, hipStream_t s);
void launch_calib_f32_to_i8(const float* x, int8_t* y, float scale, int64_t count, hipStream_t s);
void launch_calib_i8_to_f32(const int8_t* x, float* y, float scale, int64_t count, hipStream_t s);
void launch_global_avg_pool(const float* x, int nc, int spatial, float* y, hipStream_t s);
void launch_softmax(const float* x, int rows, int cols, float* y, hipStream_t s);

// fp32 window pooling (max / avg) and elementwise add (+relu): eltwise_pool.hip
struct PoolArgs {
  const float* x;  // [planes][h][w]
  float* y;        // [planes][oh][ow]
  int planes, h, w, oh, ow, kh, kw, sh, sw, pt, pb, pl, pr;
  int is_max, exclusive;
};
void launch_pool2d(const PoolArgs& a, hipStream_t s);
void launch_pool2d_max_i8(const PoolArgs& a, hipStream_t s);  // x / y are int8 planes behind the float pointers
void launch_eltwise_add(const float* x, const float* y, float* o, int64_t count, int relu, hipStream_t s);

}  // namespace plhip
