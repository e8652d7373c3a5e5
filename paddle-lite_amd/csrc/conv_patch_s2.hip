// conv_patch_s2.hip — conv_patch_i8.hip's kernel over the 4 phase planes of every channel: 3x3 STRIDE-2 convs (ResNet50's
// downsampling 3x3 convs) in 6 slabs of 2 / 1 tap rows per 32 channels.  Its own translation unit (parallel build).
#include "conv_patch_kernel.h"

namespace plhip {
void launch_patch_s2(const PatchArgs& a, int out, hipStream_t s) { launch_patch_o<2, 4, 1, 4, 3, false, 2>(a, out, s); }
}  // namespace plhip
