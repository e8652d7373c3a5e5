// gemm_wide_n7.hip — the wide-tile GEMM (gemm_wide_i8.hip / gemm_wide_kernel.h) with 7 n tiles per block: its own
// translation unit so that the instantiations compile in parallel.
#include "gemm_wide_kernel.h"

namespace plhip {

void launch_wide_n7(const GemmArgs& g, int out, hipStream_t s) {
  if (g.KS == 4) launch_wide_o<7, 4>(g, out, s);
  else if (g.KS == 8) launch_wide_o<7, 8>(g, out, s);
  else if (g.KS == 16) launch_wide_o<7, 16>(g, out, s);
}

}  // namespace plhip
