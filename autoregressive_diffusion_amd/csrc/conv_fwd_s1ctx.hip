#include "conv_fwd_common.h"
int conv_dispatch_s1ctx(const OnirisConvArgs& a, hipStream_t st) { return conv3x3_pick<1, true>(a, st); }
