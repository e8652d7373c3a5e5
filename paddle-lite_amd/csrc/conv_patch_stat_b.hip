// conv_patch_stat_b.hip — conv_patch_i8.hip's kernel for M <= 64 with register-resident weights (ResNet50's res2 3x3 layers):
// 2 m tiles x 2 pixel groups per half.  Its own translation unit so that the variants compile in parallel.
#include "conv_patch_kernel.h"

namespace plhip {
void launch_patch_stat_b(const PatchArgs& a, int out, hipStream_t s) { launch_patch_o<1, 2, 2, 5, 3, true>(a, out, s); }
}  // namespace plhip
