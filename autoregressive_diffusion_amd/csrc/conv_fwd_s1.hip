#include "conv_fwd_common.h"
int conv_dispatch_s1(const OnirisConvArgs& a, hipStream_t st) { return conv3x3_pick<1, false>(a, st); }
int conv_dispatch_1x1(const OnirisConvArgs& a, hipStream_t st) {
  if (a.CoutP % 64 == 0) return launch_conv_fwd<1, 1, 64, 2, false, 16>(a, st);
  return launch_conv_fwd<1, 1, 64, 1, false, 16>(a, st);
}
