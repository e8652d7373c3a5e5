// gemm_wide_n8.hip — the wide-tile GEMM (gemm_wide_i8.hip / gemm_wide_kernel.h) with 8 n tiles per block: its own
// translation unit so that the instantiations compile in parallel.
#include "gemm_wide_kernel.h"

namespace plhip {

void launch_wide_n8(const GemmArgs& g, int out, hipStream_t s) {
  if (g.KS == 4) launch_wide_o<8, 4>(g, out, s);
  else if (g.KS == 8) launch_wide_o<8, 8>(g, out, s);
  else if (g.KS == 16) launch_wide_o<8, 16>(g, out, s);
}

}  // namespace plhip
