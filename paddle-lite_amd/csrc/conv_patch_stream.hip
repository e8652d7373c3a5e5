// conv_patch_stream.hip — conv_patch_i8.hip's kernel for C > 64 (ResNet50's res3 / res4 3x3 layers): the weight fragments of
// a (chunk, column shift) travel through the ring with the slab pair.  Its own translation unit (parallel build).
#include "conv_patch_kernel.h"

namespace plhip {
void launch_patch_stream_a(const PatchArgs& a, int out, hipStream_t s) { launch_patch_o<2, 4, 1, 5, 3, false>(a, out, s); }
}  // namespace plhip
