// 1x1 convolution of a FEW output tiles (one generated frame per sequence in the cached sampler, edm2/sampler.py:12-85: the skip convs
// of the decoder Blocks with their mp_cat + mp_silu on the way in, the attention projections with their mp_sum + clip, the embedding
// GEMMs; networks_edm2.py:62-94, attention_modules.py:29-33) -- round 6.
//
// Such a launch is a latency chain, not a bandwidth problem: 16 .. 64 workgroups, each walking its whole K in 64-channel rounds of
// load -> LDS -> barrier -> four MFMAs (conv_fwd_kernel: ~0.45 us per round, 0.9 us with the two-source conversion in front of the LDS
// store; 6.3 us at 256 input channels, 11.8 us at 512 with the concatenation).  Here the four waves of a workgroup share ONE 32-position
// x 32-channel tile and split its K: wave w takes the rounds w, w + 4, ... (one round at 256 channels, two at 512), every operand
// goes global -> registers in MFMA fragment layout (lane (r, h): 8 consecutive channels of row r -- a 64-channel round of a row is one
// 128-byte line shared by 8 loads), no LDS staging, no barrier inside the K loop; the four partial tiles meet in LDS once.
//   D[co][pos] += W[co][k] . X[pos][k]     (MFMA 32x32x16 bf16 -> fp32; lane = position)
// Two-source input (OnirisConvArgs.x2 / act_out): the lane scales and rounds its own fragment (cat_w1 / cat_w2, bf16: what
// oniris_act_fwd stores as xo) and, in the workgroup of the channel block the round is dealt to, stores mp_silu of it -- the same
// operations in the same order as conv_fwd_kernel's.
// Not bit-identical to conv_fwd_kernel (K is summed in four interleaved parts); deterministic (fixed wave order in the reduction).
#pragma once
#include "conv_kernels.h"

__global__ __launch_bounds__(256) void conv1x1_few_kernel(const ConvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ float red[4 * 16 * 64];               // [wave][accumulator register][lane]
  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int M = a.T * a.H * a.W, Cin = a.Cin;
  const int pos0 = (int)blockIdx.x * 32, b = (int)blockIdx.y, cob = (int)blockIdx.z, co0 = cob * 32;
  const int pos = pos0 + r;
  const bool pvalid = pos < M;
  const size_t prow = (size_t)b * M + (pvalid ? pos : 0);
  const bf16* wrow = (const bf16*)a.w_own + (size_t)(co0 + r) * a.CinP;
  const bf16* xg = (const bf16*)a.x;
  const bf16* x2g = (const bf16*)a.x2;
  const int C1 = x2g ? a.x_split : Cin;
  const int nround = (Cin + 63) >> 6;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 1
  for (int ch = wave; ch < nround; ch += 4) {
    const int c0 = ch * 64 + h * 8;
    u32x4 wv[4], xv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = c0 + ks * 16;
      wv[ks] = (c < a.CinP) ? *(const u32x4*)(wrow + c) : u32x4{0u, 0u, 0u, 0u};
      xv[ks] = u32x4{0u, 0u, 0u, 0u};
      if (pvalid && c < Cin)
        xv[ks] = (c < C1) ? *(const u32x4*)(xg + prow * C1 + c) : *(const u32x4*)(x2g + prow * (Cin - C1) + (c - C1));
    }
    if (x2g) {
      const bool act_mine = a.act_out && (ch % (int)gridDim.z) == cob;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = c0 + ks * 16;
        const float w = (c < C1) ? a.cat_w1 : a.cat_w2;
        const bf16x8 in = __builtin_bit_cast(bf16x8, xv[ks]);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(in[k]) * w);
        xv[ks] = __builtin_bit_cast(u32x4, o);
        if (act_mine && pvalid && c < Cin) {
          bf16x8 av;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float z = bf2f(o[k]);
            av[k] = f2bf(z * sigmoid_fast(z) * (1.0f / 0.596f));
          }
          *(bf16x8*)((bf16*)a.act_out + prow * Cin + c) = av;
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = mfma32(__builtin_bit_cast(bf16x8, wv[ks]), __builtin_bit_cast(bf16x8, xv[ks]), acc);
  }
  // the four partial tiles meet in LDS; wave w finishes accumulator registers 4 w .. 4 w + 3 = output channels co0 + 8 w + 4 h + (0..3)
#pragma unroll
  for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
  __syncthreads();
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = wave * 4 + t;
    v[t] = ((red[(0 * 16 + i) * 64 + lane] + red[(1 * 16 + i) * 64 + lane]) + red[(2 * 16 + i) * 64 + lane]) + red[(3 * 16 + i) * 64 + lane];
  }
  const int co = co0 + 8 * wave + 4 * h;
  if (!pvalid || co >= a.Cout) return;
  const size_t o = prow * a.Cout + co;
  if (a.epi == ONIRIS_EPI_MPSUM) {
    const bf16x4 rv = *(const bf16x4*)((const bf16*)a.res + o);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float z = a.ta * bf2f(rv[t]) + a.tb * v[t];
      if (a.clip > 0.f) z = fminf(fmaxf(z, -a.clip), a.clip);
      v[t] = z;
    }
  }
  bf16x4 ov;
#pragma unroll
  for (int t = 0; t < 4; ++t) ov[t] = f2bf(v[t]);
  *(bf16x4*)((bf16*)a.out + o) = ov;
#endif
}

// the launches conv1x1_few_kernel serves: at most one 32 x 32 tile per CU, plain or mp_sum epilogue, no side outputs of the training path
static inline bool conv1x1_few_ok(const OnirisConvArgs& a) {
  const long long M = (long long)a.T * a.H * a.W;
  return a.taps == 1 && a.S == 1 && !a.ctx && (a.epi == ONIRIS_EPI_NONE || a.epi == ONIRIS_EPI_MPSUM) && !a.out2 && !a.ctx_out &&
         !a.coef_own && !a.coef_ctx && a.Cin % 8 == 0 && a.Cout % 4 == 0 && a.B <= 65535 && a.CoutP / 32 <= 65535 &&
         (!a.x2 || (a.x_split > 0 && a.x_split < a.Cin && a.x_split % 8 == 0)) &&
         (long long)a.B * ((M + 31) / 32) * (a.CoutP / 32) <= 256 && M < (1LL << 31) / 32;
}

static int launch_conv1x1_few(const OnirisConvArgs& a, hipStream_t stream) {
  ConvDev d;
  memset(&d, 0, sizeof(d));
  d.a = a;
  const int M = a.T * a.H * a.W;
  oniris_launch(conv1x1_few_kernel, dim3((unsigned)cdiv(M, 32), (unsigned)a.B, (unsigned)(a.CoutP / 32)), dim3(256), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
