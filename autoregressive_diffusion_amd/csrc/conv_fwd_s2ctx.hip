#include "conv_fwd_common.h"
#include "conv_glds.h"
#include "conv_stream.h"

// DART training layout.  big_tile: 0/1/2 = register-staged kernels (conv_kernels.h), >= 3 = persistent LDS-DMA kernel
// (conv_glds.h: 8 waves, 16x16-pixel workgroup tiles) wherever the shape allows it.
template <int NT>
static int glds_pick(const OnirisConvArgs& a, hipStream_t st) {
  if constexpr (NT == 1)            // one 32-channel chunk: all three phases of a tile are resident, copies run a tile ahead
    if (a.Cin == 32 && a.big_tile != 7) return launch_conv_glds<1, 16, 8, 1, 1, true, true>(a, st);
  return launch_conv_glds<NT, 16, 8, 1>(a, st);
}

int conv_dispatch_s2ctx(const OnirisConvArgs& a, hipStream_t st) {
  // 32 -> <= 32 channels (the 64x64 level): frames streamed through an LDS ring, weights in registers (conv_stream.h)
  if (a.big_tile >= 4 && conv_stream_ok(a)) return launch_conv_stream(a, st);
  if (a.big_tile >= 3 && conv_glds_ok(a, 16, 16, (a.CoutP % 64 == 0) ? 64 : 32))
    return (a.CoutP % 64 == 0) ? glds_pick<2>(a, st) : glds_pick<1>(a, st);
  // 8x8 images: two whole frames per workgroup, 4 position waves x 2 channel waves (32 channels each)
  if (a.big_tile >= 3 && a.W == 8 && a.H == 8 && a.CoutP % 64 == 0 && conv_glds_ok(a, 8, 8, 64))
    return launch_conv_glds<1, 8, 8, 1, 2>(a, st);
  // the register-staged kernels do not report clips: the flag says "assume the forward clipped" (the backward pre-pass then reads
  // the clipped output and masks the gradient -- always correct), whatever the caller predicted about the kernel family
  if (a.clip_flag) {
    const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)a.clip_flag, 1, 1, st);
    if (e != hipSuccess) { oniris_set_error("conv: clip_flag fill failed: %s", hipGetErrorString(e)); return ONIRIS_ELAUNCH; }
  }
  return conv3x3_pick<2, true>(a, st);
}
