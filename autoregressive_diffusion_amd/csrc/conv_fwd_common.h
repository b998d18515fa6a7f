// dispatch glue shared by the conv_fwd_*.hip translation units (split so that hipcc builds them in parallel)
#pragma once
#include "conv_kernels.h"

template <int S, bool HAS_CTX, int NT>
static int conv3x3_pick_patch(const OnirisConvArgs& a, hipStream_t st) {
  const int W = a.W, H = a.H;
  if constexpr (S == 2 && NT == 2) {   // DART training layout: 8-wave workgroups (256 positions) halve the weight-slab
    // LDS traffic per MFMA (the slab is shared by twice as many positions) -- when that still leaves >= 1 workgroup per CU
    const long long wg8 = (long long)a.B * cdiv(a.T * H * W, 256) * (a.CoutP / 64);
    if (a.big_tile == 2 || (a.big_tile == 1 && wg8 >= 256)) {
      if (W >= 16 && W % 16 == 0 && H % 16 == 0) return launch_conv_fwd<S, 9, 32, NT, HAS_CTX, 16, 8>(a, st);
      if (W == 8 && H % 8 == 0) return launch_conv_fwd<S, 9, 32, NT, HAS_CTX, 8, 8>(a, st);
    }
  }
  if (W >= 16 && W % 16 == 0 && H % 8 == 0) return launch_conv_fwd<S, 9, 32, NT, HAS_CTX, 16>(a, st);
  if (W == 8 && H % 8 == 0) return launch_conv_fwd<S, 9, 32, NT, HAS_CTX, 8>(a, st);
  if (W == 4 && H % 4 == 0) return launch_conv_fwd<S, 9, 32, NT, HAS_CTX, 4>(a, st);
  if (W == 2 && H % 2 == 0) return launch_conv_fwd<S, 9, 32, NT, HAS_CTX, 2>(a, st);
  oniris_set_error("conv_fwd: unsupported image size %dx%d for a 3x3 kernel", H, W);
  return ONIRIS_EUNSUPPORTED;
}

template <int S, bool HAS_CTX>
static int conv3x3_pick(const OnirisConvArgs& a, hipStream_t st) {
  if (a.CoutP % 64 == 0) return conv3x3_pick_patch<S, HAS_CTX, 2>(a, st);
  return conv3x3_pick_patch<S, HAS_CTX, 1>(a, st);
}

int conv_dispatch_s2ctx(const OnirisConvArgs& a, hipStream_t st);
int conv_dispatch_s1ctx(const OnirisConvArgs& a, hipStream_t st);
int conv_dispatch_s1(const OnirisConvArgs& a, hipStream_t st);
int conv_dispatch_1x1(const OnirisConvArgs& a, hipStream_t st);
