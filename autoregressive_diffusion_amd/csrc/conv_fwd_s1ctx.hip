#include "conv_fwd_common.h"
#include "conv_eval1.h"
// Eval layout (S == 1, context = cached frames).  One NEW frame per sequence (the sampler's cached evaluations) goes to
// the weight-streaming kernel of conv_eval1.h (big_tile >= 3; bit 4 of big_tile switches it off for A/B); everything
// else -- prefill over several frames, small images, Cin not a multiple of 32 -- to the register-staged kernels.
int conv_dispatch_s1ctx(const OnirisConvArgs& a, hipStream_t st) {
  if (a.big_tile >= 3 && !(a.big_tile & 16) && conv_eval1_ok(a)) return launch_conv_eval1(a, st);
  if (a.ctx_prod_mode != 0) {
    oniris_set_error("conv_fwd: ctx_prod_mode %d needs the one-frame kernel (S = T = 1, cached pair, Cin %% 32 == 0, H, W %% 8 == 0, big_tile >= 3)", a.ctx_prod_mode);
    return ONIRIS_EUNSUPPORTED;
  }
  return conv3x3_pick<1, true>(a, st);
}
