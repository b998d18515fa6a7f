#include "conv_fwd_common.h"
int conv_dispatch_s2ctx(const OnirisConvArgs& a, hipStream_t st) { return conv3x3_pick<2, true>(a, st); }
