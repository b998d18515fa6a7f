// C-ABI entry for the gated causal convolution (argument validation + variant selection).
#include "conv_fwd_common.h"

extern "C" int oniris_conv_fwd(const OnirisConvArgs* args, oniris_stream_t stream_) {
  hipStream_t st = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(args && args->w_own && ((args->x && args->out) || args->ctx_prod_mode == 3), "conv_fwd: null pointer");
  const OnirisConvArgs& a = *args;
  if (a.ctx_prod_mode != 0 && !(a.S == 1 && a.ctx && a.taps == 9 && a.ctx_prod)) {
    oniris_set_error("conv_fwd: ctx_prod_mode %d is for the one-frame cached evaluation (S == 1, context path, ctx_prod given)", a.ctx_prod_mode);
    return ONIRIS_EUNSUPPORTED;
  }
  ONIRIS_CHECK_ARG(a.taps == 9 || a.taps == 1, "conv_fwd: taps must be 1 or 9 (got %d)", a.taps);
  ONIRIS_CHECK_ARG(a.B > 0 && a.T > 0 && a.H > 0 && a.W > 0 && (a.S == 1 || a.S == 2), "conv_fwd: bad sizes");
  ONIRIS_CHECK_ARG(a.Cin % 8 == 0 && a.Cout % 8 == 0, "conv_fwd: Cin, Cout must be multiples of 8 (%d,%d)", a.Cin, a.Cout);
  ONIRIS_CHECK_ARG(a.CoutP % 32 == 0 && a.CinP % 64 == 0 && a.CoutP >= a.Cout && a.CinP >= a.Cin,
                   "conv_fwd: bad padded sizes CoutP=%d CinP=%d", a.CoutP, a.CinP);
  ONIRIS_CHECK_ARG(a.epi != ONIRIS_EPI_MPSUM || a.res, "conv_fwd: EPI_MPSUM needs res");
  ONIRIS_CHECK_ARG(a.epi != ONIRIS_EPI_EMB_SILU || (a.escale && a.out2), "conv_fwd: EPI_EMB_SILU needs escale/out2");
  ONIRIS_CHECK_ARG(a.x2 == nullptr || (a.taps == 1 && a.ctx == nullptr), "conv_fwd: x2 (concatenated input) is a 1x1, context-free option");
  const bool has_ctx = a.ctx != nullptr;
  ONIRIS_CHECK_ARG(!has_ctx || (a.w_ctx && a.taps == 9), "conv_fwd: context path needs w_ctx and taps == 9");
  // clip_flag is answered by the LDS-DMA / streaming 3x3 kernels of the training layouts only (conv_dispatch_s2ctx, _s1); every
  // other variant reports "assume clipped", which is always correct for the backward (it then reads the clipped output)
  if (a.clip_flag && (a.taps == 1 || (a.S == 1 && has_ctx))) {
    const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)a.clip_flag, 1, 1, st);
    if (e != hipSuccess) { oniris_set_error("conv: clip_flag fill failed: %s", hipGetErrorString(e)); return ONIRIS_ELAUNCH; }
  }
  if (a.taps == 1) {
    ONIRIS_CHECK_ARG(a.S == 1, "conv_fwd: 1x1 variant expects S == 1 (fold slots into T)");
    return conv_dispatch_1x1(a, st);
  }
  if (a.S == 2) {
    ONIRIS_CHECK_ARG(has_ctx, "conv_fwd: S == 2 is the DART training layout and needs the context path");
    return conv_dispatch_s2ctx(a, st);
  }
  return has_ctx ? conv_dispatch_s1ctx(a, st) : conv_dispatch_s1(a, st);
}
