#include "conv_fwd_common.h"
#include "conv_glds.h"
#include "conv1x1_glds.h"
#include "conv_plain_stream.h"
#include "conv1x1_few.h"

// Plain 3x3 convs (the 2-D training steps, stem / non-gated layers): with big_tile >= 3 and an even frame count they
// run on the persistent LDS-DMA kernel as "two slots of T/2 frames, no context phases".
int conv_dispatch_s1(const OnirisConvArgs& a, hipStream_t st) {
  // 32 -> <= 32 channels on 16-pixel-wide images (the 64x64 level in the 2-D steps): frames streamed past register-resident weights
  if (a.big_tile >= 4 && !(a.big_tile & 128) && conv_plain_stream_ok(a)) return launch_conv_plain_stream(a, st);
  if (a.big_tile >= 3) {
    OnirisConvArgs b = a;
    b.S = 2; b.T = a.T / 2;
    b.ctx_T = 0; b.ctx_bstride = 0;
    if (conv_glds_ok(a, 16, 16, (a.CoutP % 64 == 0) ? 64 : 32, true))
      return (a.CoutP % 64 == 0) ? launch_conv_glds<2, 16, 8, 1, 1, false>(b, st) : launch_conv_glds<1, 16, 8, 1, 1, false>(b, st);
    if (a.W == 8 && a.H == 8 && a.CoutP % 64 == 0 && conv_glds_ok(a, 8, 8, 64, true))
      return launch_conv_glds<1, 8, 8, 1, 2, false>(b, st);
  }
  if (a.clip_flag) {       // the register-staged kernels do not report clips: "assume clipped" (see conv_dispatch_s2ctx)
    const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)a.clip_flag, 1, 1, st);
    if (e != hipSuccess) { oniris_set_error("conv: clip_flag fill failed: %s", hipGetErrorString(e)); return ONIRIS_ELAUNCH; }
  }
  return conv3x3_pick<1, false>(a, st);
}
int conv_dispatch_1x1(const OnirisConvArgs& a, hipStream_t st) {
  if (a.x2) {                                       // two-source input + activation side output: the register-staged kernel only
    if (a.x_split <= 0 || a.x_split >= a.Cin || a.x_split % 8 != 0 || a.Cin % 8 != 0 || a.S != 1 ||
        (a.big_tile >= 3 && conv1x1_glds_ok(a))) {
      oniris_set_error("conv_fwd: x2 / act_out (concatenated input) is served by the register-staged 1x1 kernel only "
                       "(x_split a multiple of 8 inside (0, Cin), fewer than 8192 positions)");
      return ONIRIS_EUNSUPPORTED;
    }
  }
  if (a.big_tile >= 3 && conv1x1_glds_ok(a)) return launch_conv1x1_glds(a, st);     // persistent LDS-DMA GEMM
  // at most one 32 x 32 tile per CU: the four waves of a workgroup split the K of one tile (conv1x1_few.h; big_tile bit 512: off)
  if (!(a.big_tile & (64 | 512)) && conv1x1_few_ok(a)) return launch_conv1x1_few(a, st);
  // a handful of tiles (one generated frame of the cached sampler): 32-channel output tiles -- twice the workgroups, half the weight
  // rows and MFMAs per K round and workgroup, the activation rounds of a two-source launch dealt to twice as many blocks; the launch
  // is a latency chain, not a bandwidth problem (rollout 34.1 -> 34.9 frames/s, round 6).  Same K order per output: same bits.
  // (Wider K rounds -- every load of the tile in flight at once -- were measured SLOWER: 8.1 -> 11.1 us per launch; the unrolled
  // code of a 256-channel round is fetched through a cold instruction cache on every launch.)
  // (from 768 input channels on the launch may be a split-K pair, whose slice count follows the tile count: left as it was)
  if (!(a.big_tile & 64) && a.Cin < 768 && (long long)a.B * cdiv(a.T * a.H * a.W, 128) * (a.CoutP / 32) <= 256) {
    // ... and 32- or 64-position tiles (one / two waves) while that still leaves at most one workgroup per CU: a workgroup pulls its
    // x rows and weight rows at the per-CU streaming rate (~20 B / clock), 128 rows x 256 channels are 64 KB = 1.4 us by themselves
    // (38.0 -> 38.7 frames/s)
    const long long per32 = (long long)a.B * (a.CoutP / 32);
    if (per32 * cdiv(a.T * a.H * a.W, 32) <= 256) return launch_conv_fwd<1, 1, 64, 1, false, 16, 1>(a, st);
    if (per32 * cdiv(a.T * a.H * a.W, 64) <= 256) return launch_conv_fwd<1, 1, 64, 1, false, 16, 2>(a, st);
    return launch_conv_fwd<1, 1, 64, 1, false, 16>(a, st);
  }
  if (a.CoutP % 64 == 0) return launch_conv_fwd<1, 1, 64, 2, false, 16>(a, st);
  // 96 output channels (the 32 -> 96 dgrad of the 64x64-level skip conv): one workgroup per pixel tile instead of three
  // that re-read the same input rows (414 -> ~190 us at B = 8)
  if (a.CoutP % 96 == 0) return launch_conv_fwd<1, 1, 64, 3, false, 16>(a, st);
  return launch_conv_fwd<1, 1, 64, 1, false, 16>(a, st);
}
