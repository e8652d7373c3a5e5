// gemm_wide_i8.hip — the 1x1-convolution GEMM, third generation: ONE wide tile per CU, every operand byte in flight at once.
//
// Replaces the same reference code as gemm_i8.hip / gemm_tr_i8.hip (gemm_prepack_int8, lite/backends/arm/math/
// gemm_prepacked_int8.cc:2582-2744 hot loop, :643-796 epilogue; packb_int8 :3285; the batch loop of conv1x1s1_gemm_int8,
// conv_impl.cc:260-331) for the layers whose GEMM is SMALL per CU: MobileNetV1's 14x14 and 7x7 pointwise convs
// (512 -> 512: 50 k outputs x K 512 per CU at batch 128).
//
// Why a third kernel (round-2 evidence, profiles/r02_final_gemm_timeline_pw8.txt + tools/ingest_bench.hip, round 3):
//   * the ring kernels keep 3-4 K-steps (48-64 KB per CU) in flight and re-fetch the weight panel once per 128-column
//     tile: 430 KB of operand ingest per CU for the 512 -> 512 layer, taken in at ~21 B/clk;
//   * that rate is NOT what a CU can take in: LDS-DMA and register loads of 1-KiB pieces run at 56-59 B/clk/CU from L2
//     (tools/ingest_bench.hip); a K-step takes ~1000 cycles because the next one's bytes are still travelling
//     (64 KB in flight / 21 B/clk = 3000 cycles of loaded latency), and prologue, K loop and epilogue of the 424 short
//     blocks add up instead of overlapping.
// Here a block owns a 256 (m) x 32*NTT (n) tile = the whole share of one CU (1 block per CU, 8 waves), so
//   * the weight panel is read ONCE per CU: wave w loads ITS 32 rows x K straight into registers (fragment order, as
//     pack_weights_kernel wrote them: 1 KiB contiguous per load, the fastest form there is), K <= 1024;
//   * the activation tile (K x 32*NTT bytes, <= 128 KiB) is LDS-resident for the whole K loop: every DMA piece has its own
//     slot, so ALL loads of the tile are issued before the first MFMA (no ring, no slot reuse, no issue inside the loop)
//     and a K-step only waits for bytes that were requested K-steps * 2 instructions ago;
//   * operand reads as in gemm_tr_i8.hip: ds_read_b64_tr_b8 on the raw NCHW rows (no VALU in the K loop), activations
//     are the A operand, so a lane owns ONE output channel and 16 consecutive columns per 32 x 32 tile after two
//     v_permlane32_swap; the int8 tile leaves through a wave-private LDS image as 16-byte pieces of whole rows.
// LDS image of one K-step: [group of 8 chunks (128 columns)][kg = k/8][row q = k%8][slot s][16 B], chunk j = s ^ 2(q>>1).
// One DMA instruction = one (group, kg): 8 lanes walk 128 contiguous bytes of ONE k row (consecutive lanes on different
// rows measured 2-4x slower from L2, tools/ingest_bench.hip rows128T); the XOR keeps the transposed read conflict-free
// (a half-wave reads chunk pair (2t, 2t+1) of rows 0..7: slots 2((t&3) ^ (q>>1)) + parity, 16 distinct 16-byte bank slots).
// Column space, end-aligned last 16-byte chunk of an image, `skip`: as in gemm_i8_dma_kernel (gemm_i8.hip).
#include "gemm_wide_kernel.h"

namespace plhip {

// ---- diagnostic timeline (PLHIP_GEMM_DEBUG & 32; never set in production): the kernels write here through GemmArgs::stamps
__device__ unsigned long long g_wide_stamps[512 * 8 * WIDE_STAMP_SLOTS];
unsigned long long* wide_stamps_ptr() {
  static unsigned long long* p = nullptr;
  if (!p) (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_wide_stamps));
  return p;
}

int debug_read_wide_stamps(void* dst, size_t bytes) {
  const size_t cap = sizeof(unsigned long long) * 512 * 8 * WIDE_STAMP_SLOTS;
  if (bytes > cap) bytes = cap;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wide_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

static int g_wide_ntt_override = -1;  // tests / A-B runs: plhip_debug_wide_ntt (-1 = PLHIP_WIDE_NTT or automatic)
void debug_set_wide_ntt(int v) { g_wide_ntt_override = v; }

static int wide_env() {  // PLHIP_GEMM_WIDE: 1 (default) on, 0 = second-generation kernels only (A/B runs)
  const int v = knob("GEMM_WIDE", 1);
  return v;
}

// The tile of a launch: n tiles per block so that the blocks fill the CUs once with as little idle tail as possible.
// Returns 0 when the shape is outside this kernel (the caller falls back to the ring kernels).
int gemm_wide_ntt(const GemmArgs& g) {
  if (!wide_env() || g.im_kw != 0 || g.res || g.y2) return 0;
  if (g.K != g.KS * 32 || (g.KS != 4 && g.KS != 8 && g.KS != 16 && g.KS != 32)) return 0;
  // dense slabs only: the column space IS the output row (an im2col buffer whose rows are padded to a multiple of 4 has
  // HWX > HWY: its pad columns must not be stored, and the whole-chunk stores here would spill into the next row)
  if (g.M < 256 || g.HWX < 16 || g.HWX != g.HWY) return 0;
  const int CPI = (g.HWX + 15) >> 4;
  const long chunks = (long)g.NB * CPI;
  if (chunks * 16 >= ((long)1 << 31) - 4096) return 0;
  const int force_env = knob("WIDE_NTT", 0);
  const int force = g_wide_ntt_override >= 0 ? g_wide_ntt_override : force_env;
  const int mblocks = (g.M + 255) / 256;
  int best = 0;
  double best_cost = 1e30;
  const int cands[3] = {4, 7, 8};
  for (int i = 0; i < 3; ++i) {
    const int ntt = cands[i];
    if (force && ntt != force) continue;
    {  // the activation tile + the staging images must fit the LDS (wide_lds_bytes)
      const int c1 = 2 * ntt > 8 ? 2 * ntt - 8 : 0;
      if ((long)g.KS * 4 * (1024 + c1 * 128) + 8 * 32 * 48 + 8 * WIDE_STAMP_SLOTS * 8 > 160 * 1024) continue;
    }
    const long nblocks = (chunks + 2 * ntt - 1) / (2 * ntt);
    const long blocks = nblocks * mblocks;
    const long rounds = (blocks + 255) / 256;
    // time ~ rounds x (operand ingest of a tile + a fixed prologue / epilogue share)
    const double cost = (double)rounds * ((double)g.KS * 32 * (256 + 32 * ntt) + 40000.0);
    if (cost < best_cost) {
      best_cost = cost;
      best = ntt;
    }
  }
  return best;
}

bool launch_gemm_wide(const GemmArgs& g_in, int out, hipStream_t s) {
  // 32-bit outputs: the ring kernels are faster on every MobileNetV1 layer (fp32 out, batch 256: 699 vs 764 us over the 13
  // layers, 128 -> 256 @28x28 38.7 vs 54.0 us, 1024 -> 1024 @7x7 30.2 vs 37.6: profiles/r03_final_opbench_f32_*.txt): this
  // kernel's row-per-lane 16-byte stores write 32 contiguous bytes per row and instruction, the ring kernels' epilogue 64.
  // It stays reachable for them through the tile override (tests, A/B runs).
  if (out != OUT_I8 && g_wide_ntt_override < 0 && knob("WIDE_NTT", 0) == 0) return false;
  const int ntt = gemm_wide_ntt(g_in);
  if (!ntt) return false;
  GemmArgs g = g_in;
  g.stamps = (g.dbg & 32) ? wide_stamps_ptr() : nullptr;
  if (ntt == 4) launch_wide_n4(g, out, s);
  else if (g.KS == 32) return false;
  else if (ntt == 7) launch_wide_n7(g, out, s);
  else launch_wide_n8(g, out, s);
  return true;
}

}  // namespace plhip
