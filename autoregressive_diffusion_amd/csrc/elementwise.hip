// Fused magnitude-preserving glue kernels (HBM-bound; 16-byte vector loads/stores, one pass each).
// They replace the chains of elementwise ops of Block.forward (reference edm2/networks_edm2.py:62-94):
//   act_fwd / act_bwd     mp_cat (utils.py:128-134) and/or pixel norm (utils.py:83-88) followed by mp_silu (:112)
//   emb_silu_bwd          backward of  u = mp_silu(y * c[n,co])  (networks_edm2.py:75-77) incl. the per-(frame,
//                         channel) reduction for c; the forward lives in the conv epilogue (ONIRIS_EPI_EMB_SILU)
//   mpsum_bwd             backward of  out = clip(ta*res + tb*v) (networks_edm2.py:86,93), forward = conv epilogue
//   resample_down/up      2x2 mean / nearest x2 (utils.py:94-107 with f=[1,1]) and their adjoints
#include <cstdlib>
#include "common.h"
#include "../../include/oniris.h"

#define SILU_SCALE (1.0f / 0.596f)

__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_fast(z); }
__device__ __forceinline__ float dsilu_f(float z) {
  const float sg = sigmoid_fast(z);
  return sg * (1.f + z * (1.f - sg));
}

// ---------------------------------------------------------------------------------------------------------------
// v = concat(w1 * x[C1], w2 * skip[C2]) ; NORM: v <- v / (eps + |v|/sqrt(C)) ; xo = v ; a = silu(v)/0.596
// one thread = 8 channels of one pixel; G = C/8 threads per pixel (NORM needs G to be a power of two <= 64)
// rs != 0: x is resampled on the way in (Block.forward resamples before anything else, networks_edm2.py:63): rs = 1 the
// 2x2 mean, rs = 2 nearest x2, Ho x Wo = the OUTPUT grid (= the grid of pix); the resampled value is rounded to bf16 first,
// exactly what the separate resample pass stored.
template <bool NORM, bool NT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ skip,
                                                      bf16* __restrict__ xo, bf16* __restrict__ a,
                                                      float* __restrict__ sden, long long npix, int C1, int C2, float w1,
                                                      float w2, int rs, int Ho, int Wo) {
  const int C = C1 + C2, G = C >> 3;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long pix = gid / G;
  const int cg = (int)(gid % G);
  const bool ok = pix < npix;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (ok) {
    const int c = cg * 8;
    bf16x8 in;
    float w;
    if (c < C1) {
      w = w1;
      if (rs == 0) in = ldv<NT>((const bf16x8*)(x + pix * C1 + c));
      else {
        const int xo_ = (int)(pix % Wo), yo_ = (int)((pix / Wo) % Ho);
        const long long n = pix / ((long long)Wo * Ho);
        if (rs == 1) {
          const int Hi = Ho * 2, Wi = Wo * 2;
          float acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const bf16x8 t = *(const bf16x8*)(x + ((n * Hi + 2 * yo_ + dy) * Wi + 2 * xo_ + dx) * C1 + c);
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] += bf2f(t[i]);
            }
#pragma unroll
          for (int i = 0; i < 8; ++i) in[i] = f2bf(acc[i] * 0.25f);
        } else {
          in = *(const bf16x8*)(x + ((n * (Ho >> 1) + (yo_ >> 1)) * (Wo >> 1) + (xo_ >> 1)) * C1 + c);
        }
      }
    }
    else { in = ldv<NT>((const bf16x8*)(skip + pix * C2 + (c - C1))); w = w2; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(in[i]) * w;
  }
  if (NORM) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    for (int o = 1; o < G; o <<= 1) ss += __shfl_xor(ss, o);
    const float s = 1e-4f + sqrtf(ss) * rsqrtf((float)C);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= inv;
    if (ok && cg == 0 && sden) sden[pix] = s;
  }
  if (!ok) return;
  bf16x8 o, av;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    o[i] = f2bf(v[i]);
    av[i] = f2bf(silu_f(bf2f(o[i])) * SILU_SCALE);
  }
  if (xo) stv<NT>((bf16x8*)(xo + pix * C + cg * 8), o);
  stv<NT>((bf16x8*)(a + pix * C + cg * 8), av);
}

// g = dxo + da * silu'(xo)/0.596 ; NORM: g <- (g - xo * sum(g*xo) * k) / s ; dx = w1*g[:C1] (+ dadd), dskip = w2*g[C1:]
// dadd: a second gradient of x that is already complete (the decoder's gradient of an encoder output that is also a skip
// connection): added here instead of by a separate pass over the three tensors
template <bool NORM, bool NT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ da, const bf16* __restrict__ dxo,
                                                      const bf16* __restrict__ xo, const float* __restrict__ sden,
                                                      bf16* __restrict__ dx, bf16* __restrict__ dskip,
                                                      const bf16* __restrict__ dadd, long long npix,
                                                      int C1, int C2, float w1, float w2, float dxo_scale) {
  const int C = C1 + C2, G = C >> 3;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long pix = gid / G;
  const int cg = (int)(gid % G);
  const bool ok = pix < npix;
  float g[8], xv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { g[i] = 0.f; xv[i] = 0.f; }
  if (ok) {
    const bf16x8 xin = ldv<NT>((const bf16x8*)(xo + pix * C + cg * 8));
    const bf16x8 dain = ldv<NT>((const bf16x8*)(da + pix * C + cg * 8));
#pragma unroll
    for (int i = 0; i < 8; ++i) { xv[i] = bf2f(xin[i]); g[i] = bf2f(dain[i]) * dsilu_f(xv[i]) * SILU_SCALE; }
    if (dxo) {
      const bf16x8 d2 = ldv<NT>((const bf16x8*)(dxo + pix * C + cg * 8));
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] += dxo_scale * bf2f(d2[i]);
    }
  }
  if (NORM) {
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) dot += g[i] * xv[i];
    for (int o = 1; o < G; o <<= 1) dot += __shfl_xor(dot, o);
    const float s = ok ? sden[pix] : 1.f;
    // xn = x/s, s = eps + n/sqrt(C):  dx = (g - xn * sum(g*xn) * s / (n*sqrt(C))) / s,   n/sqrt(C) = s - eps,  n*sqrt(C) = C*(s-eps)
    const float nsc = (s - 1e-4f) * (float)C;
    const float k = (nsc > 0.f) ? dot * s / nsc : 0.f;
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (g[i] - xv[i] * k) * inv;
  }
  if (!ok) return;
  const int c = cg * 8;
  bf16x8 o;
  if (c < C1) {
    if (dadd) {
      const bf16x8 d3 = ldv<NT>((const bf16x8*)(dadd + pix * C1 + c));
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(g[i] * w1 + bf2f(d3[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(g[i] * w1);
    }
    stv<NT>((bf16x8*)(dx + pix * C1 + c), o);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(g[i] * w2);
    stv<NT>((bf16x8*)(dskip + pix * C2 + (c - C1)), o);
  }
}

extern "C" int oniris_act_fwd(const void* x, const void* skip, void* xo, void* a, float* sden, int64_t npix, int C1,
                              int C2, float w1, float w2, int norm, int resample, int Ho, int Wo, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int C = C1 + C2;
  ONIRIS_CHECK_ARG(x && a && npix > 0 && C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0 && (C2 == 0 || skip),
                   "act_fwd: bad arguments");
  ONIRIS_CHECK_ARG(resample == 0 || ((resample == 1 || resample == 2) && Ho > 0 && Wo > 0 && npix % ((int64_t)Ho * Wo) == 0 &&
                                     (resample == 1 || (Ho % 2 == 0 && Wo % 2 == 0))),
                   "act_fwd: resample needs the output grid (even for nearest x2)");
  ONIRIS_CHECK_ARG(!norm || ((C / 8) <= 64 && ((C / 8) & (C / 8 - 1)) == 0 && sden), "act_fwd: pixel norm needs C/8 = 2^k <= 64");
  const long long nthr = npix * (C / 8);
  const dim3 grid((unsigned)((nthr + 255) / 256));
  const bool nt = (long long)npix * C * 2 >= oniris_ew_nt_bytes();
#define ACT_FWD_LAUNCH(NORM_, NT_) ONIRIS_KLAUNCH((act_fwd_kernel<NORM_, NT_>), grid, dim3(256), 0, stream, (const bf16*)x, (const bf16*)skip, (bf16*)xo, (bf16*)a, sden, (long long)npix, C1, C2, w1, w2, resample, Ho, Wo)
  if (norm) { if (nt) ACT_FWD_LAUNCH(true, true); else ACT_FWD_LAUNCH(true, false); }
  else { if (nt) ACT_FWD_LAUNCH(false, true); else ACT_FWD_LAUNCH(false, false); }
#undef ACT_FWD_LAUNCH
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_act_bwd(const void* da, const void* dxo, const void* xo, const float* sden, void* dx, void* dskip,
                              const void* dadd, int64_t npix, int C1, int C2, float w1, float w2, int norm,
                              float dxo_scale, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int C = C1 + C2;
  ONIRIS_CHECK_ARG(da && xo && dx && npix > 0 && C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0 && (C2 == 0 || dskip),
                   "act_bwd: bad arguments");
  ONIRIS_CHECK_ARG(!norm || ((C / 8) <= 64 && ((C / 8) & (C / 8 - 1)) == 0 && sden), "act_bwd: pixel norm needs C/8 = 2^k <= 64");
  const long long nthr = npix * (C / 8);
  const dim3 grid((unsigned)((nthr + 255) / 256));
  const bool nt = (long long)npix * C * 2 >= oniris_ew_nt_bytes();
#define ACT_BWD_LAUNCH(NORM_, NT_) ONIRIS_KLAUNCH((act_bwd_kernel<NORM_, NT_>), grid, dim3(256), 0, stream, (const bf16*)da, (const bf16*)dxo, (const bf16*)xo, sden, (bf16*)dx, (bf16*)dskip, (const bf16*)dadd, (long long)npix, C1, C2, w1, w2, dxo_scale)
  if (norm) { if (nt) ACT_BWD_LAUNCH(true, true); else ACT_BWD_LAUNCH(true, false); }
  else { if (nt) ACT_BWD_LAUNCH(false, true); else ACT_BWD_LAUNCH(false, false); }
#undef ACT_BWD_LAUNCH
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// u = silu(y*c)/0.596:  dz = du*silu'(y*c)/0.596 ; dy = dz*c ; dc[n][co] += sum_pixels dz*y
// grid = (frames, pixel slices); threads: channel group = tid % G, pixel lane = tid / G
template <bool NT>
__global__ __launch_bounds__(256) void emb_silu_bwd_kernel(const bf16* __restrict__ du, const bf16* __restrict__ y,
                                                           const float* __restrict__ c, bf16* __restrict__ dy,
                                                           float* __restrict__ dc, int P, int C, int pix_per_block,
                                                           int c_pitch) {
  __shared__ float acc[512];
  const int n = blockIdx.x, G = C >> 3;
  const int cg = threadIdx.x % G, pl = threadIdx.x / G, npl = 256 / G;
  for (int i = threadIdx.x; i < C; i += 256) acc[i] = 0.f;
  __syncthreads();
  float cv[8], part[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cv[i] = 0.f; part[i] = 0.f; }
  if (pl < npl) {
    const float4 c0 = *(const float4*)(c + (size_t)n * c_pitch + cg * 8), c1 = *(const float4*)(c + (size_t)n * c_pitch + cg * 8 + 4);
    cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w; cv[4] = c1.x; cv[5] = c1.y; cv[6] = c1.z; cv[7] = c1.w;
    const int p0 = blockIdx.y * pix_per_block;
    const int p1 = min(P, p0 + pix_per_block);
    for (int p = p0 + pl; p < p1; p += npl) {
      const size_t off = ((size_t)n * P + p) * C + cg * 8;
      const bf16x8 duv = ldv<NT>((const bf16x8*)(du + off)), yv = ldv<NT>((const bf16x8*)(y + off));
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float yy = bf2f(yv[i]);
        const float dz = bf2f(duv[i]) * dsilu_f(yy * cv[i]) * SILU_SCALE;
        o[i] = f2bf(dz * cv[i]);
        part[i] += dz * yy;
      }
      stv<NT>((bf16x8*)(dy + off), o);
    }
    if ((G & (G - 1)) != 0) {                        // (channel groups not a power of two: every lane adds its own)
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(&acc[cg * 8 + i], part[i]);
    }
  }
  if ((G & (G - 1)) == 0) {
    // lanes of a wave with the same channel group (cg = lane % G; every thread is a pixel lane then) add up by shuffles:
    // G lanes per wave reach the LDS accumulators instead of 64 (see gconv_bwd_fused_kernel)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = part[i];
      for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if ((int)(threadIdx.x & 63) < G) atomicAdd(&acc[cg * 8 + i], v);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(dc + (size_t)n * C + i, acc[i]);
}

// dc is zeroed by a KERNEL, not by hipMemsetAsync: a memset node captured into a hipGraph was observed (ROCm 7.0/7.2,
// gfx950) to leave stale values behind from the second replay on -- the 2-D training graph, the only one that takes this
// path, replayed garbage emb-scale gradients (emb_noise / emb_label / emb_linear / emb_gain) while eager launches and
// the first replay were right.  (This was the "graph replay nondeterminism" noted in round 1.)
__global__ void zero_f32_kernel(float* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

extern "C" int oniris_emb_silu_bwd(const void* du, const void* y, const float* c, void* dy, float* dc, int N, int P,
                                   int C, int c_pitch, int dc_is_zero, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(du && y && c && dy && dc && N > 0 && P > 0 && C > 0 && C % 8 == 0 && C <= 512,
                   "emb_silu_bwd: bad arguments (C %% 8 == 0, C <= 512)");
  int slices = 1;
  const int npl = 256 / (C / 8) > 0 ? 256 / (C / 8) : 1;
  // (every block ends with C global atomics: fewer, longer blocks are faster -- see oniris_gconv_bwd_fused)
  static int tgt = -1;                               // (ONIRIS_EMB_SILU_BLOCKS: A/B knob)
  if (tgt < 0) { const char* e = getenv("ONIRIS_EMB_SILU_BLOCKS"); tgt = e ? atoi(e) : 1024; }
  while (slices < 16 && P / (slices * 2) >= npl * 4 && (long long)N * slices < tgt) slices *= 2;
  const int ppb = cdiv(P, slices);
  const size_t ndc = (size_t)N * C;
  if (!dc_is_zero) ONIRIS_KLAUNCH(zero_f32_kernel, dim3((unsigned)((ndc + 255) / 256)), dim3(256), 0, stream, dc, ndc);
  const int cp = c_pitch > 0 ? c_pitch : C;
  ONIRIS_CHECK_ARG(cp >= C && cp % 4 == 0, "emb_silu_bwd: c_pitch must be a multiple of 4 and >= C");
  if ((long long)N * P * C * 2 >= oniris_ew_nt_bytes())
    ONIRIS_KLAUNCH(emb_silu_bwd_kernel<true>, dim3(N, slices), dim3(256), 0, stream, (const bf16*)du, (const bf16*)y, c,
                       (bf16*)dy, dc, P, C, ppb, cp);
  else
    ONIRIS_KLAUNCH(emb_silu_bwd_kernel<false>, dim3(N, slices), dim3(256), 0, stream, (const bf16*)du, (const bf16*)y, c,
                       (bf16*)dy, dc, P, C, ppb, cp);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// out = clip(ta*res + tb*v): given g = d out  ->  dres = ta*g*[|out|<clip], dv = tb*g*[|out|<clip]
template <bool NT>
__global__ void mpsum_bwd_kernel(const bf16* __restrict__ g, const bf16* __restrict__ out, bf16* __restrict__ dres,
                                 bf16* __restrict__ dv, size_t n8, float ta, float tb, float clip) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 gv = ldv<NT>((const bf16x8*)(g + i * 8));
    bf16x8 a, b;
    if (clip > 0.f) {
      const bf16x8 ov = ldv<NT>((const bf16x8*)(out + i * 8));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float gg = (fabsf(bf2f(ov[k])) < clip) ? bf2f(gv[k]) : 0.f;
        a[k] = f2bf(gg * ta); b[k] = f2bf(gg * tb);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float gg = bf2f(gv[k]); a[k] = f2bf(gg * ta); b[k] = f2bf(gg * tb); }
    }
    stv<NT>((bf16x8*)(dres + i * 8), a);
    if (dv) stv<NT>((bf16x8*)(dv + i * 8), b);
  }
}

extern "C" int oniris_mpsum_bwd(const void* g, const void* out, void* dres, void* dv, int64_t numel, float ta, float tb,
                                float clip, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(g && dres && numel > 0 && numel % 8 == 0 && (clip <= 0.f || out), "mpsum_bwd: bad arguments");
  const size_t n8 = (size_t)numel / 8;
  size_t nb = (n8 + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (numel * 2 >= oniris_ew_nt_bytes())
    ONIRIS_KLAUNCH(mpsum_bwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)g, (const bf16*)out,
                       (bf16*)dres, (bf16*)dv, n8, ta, tb, clip);
  else
    ONIRIS_KLAUNCH(mpsum_bwd_kernel<false>, dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)g, (const bf16*)out,
                       (bf16*)dres, (bf16*)dv, n8, ta, tb, clip);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// The aliasing protocol for a PLAIN conv with the mp_sum + clip epilogue (the 2-D training steps' conv_res1): dv = tb * g * mask and
// dres = ta * g * mask are scaled copies of g, so nothing is written -- dgrad / wgrad read g with the coefficient tb, the
// residual's consumer with the scale ta -- unless the forward really clipped something (OnirisConvArgs.clip_flag != 0): only
// then this launch reads the clipped output and masks g IN PLACE (g must be a buffer this backward owns).  Flag clear: every
// thread leaves after one scalar load.
__global__ void mpsum_mask_kernel(bf16* __restrict__ g, const bf16* __restrict__ out, size_t n8, float clip,
                                  const int* __restrict__ flag) {
  if (*flag == 0) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    bf16x8 gv = *(const bf16x8*)(g + i * 8);
    const bf16x8 ov = *(const bf16x8*)(out + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (!(fabsf(bf2f(ov[k])) < clip)) gv[k] = f2bf(0.f);
    *(bf16x8*)(g + i * 8) = gv;
  }
}

extern "C" int oniris_mpsum_mask(void* g, const void* out, int64_t numel, float clip, const int32_t* clip_flag,
                                 oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(g && out && clip_flag && clip > 0.f && numel > 0 && numel % 8 == 0, "mpsum_mask: bad arguments");
  const size_t n8 = (size_t)numel / 8;
  size_t nb = (n8 + 255) / 256;
  if (nb > 2048) nb = 2048;
  ONIRIS_KLAUNCH(mpsum_mask_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (bf16*)g, (const bf16*)out, n8, clip,
                     (const int*)clip_flag);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// resample: mode 0: out[n][y][x] = mean of the 2x2 input block (H,W = INPUT size);  mode 1: nearest x2 (H,W = INPUT size)
// `scale` multiplies the result (adjoints: down^T = 0.25 * up, up^T = 4 * down -> pass scale accordingly)
__global__ void resample_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, const bf16* __restrict__ add,
                                long long nout8, int H, int W, int C, int mode, float scale) {
  const int G = C >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nout8; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    long long p = i / G;
    float v[8];
    if (mode == 0) {
      const int Wo = W >> 1, Ho = H >> 1;
      const int xo = (int)(p % Wo); p /= Wo;
      const int yo = (int)(p % Ho); const long long n = p / Ho;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const bf16x8 a = *(const bf16x8*)(in + ((n * H + 2 * yo + dy) * W + 2 * xo + dx) * C + cg * 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += bf2f(a[k]);
        }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= 0.25f * scale;
    } else {
      const int Wo = W << 1, Ho = H << 1;
      const int xo = (int)(p % Wo); p /= Wo;
      const int yo = (int)(p % Ho); const long long n = p / Ho;
      const bf16x8 a = *(const bf16x8*)(in + ((n * H + (yo >> 1)) * W + (xo >> 1)) * C + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = bf2f(a[k]) * scale;
    }
    if (add) {                                        // (see act_bwd_kernel: a second, complete gradient of the same tensor)
      const bf16x8 d3 = *(const bf16x8*)(add + i * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += bf2f(d3[k]);
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k]);
    *(bf16x8*)(out + i * 8) = o;
  }
}

extern "C" int oniris_resample(const void* in, void* out, const void* add, int64_t N, int H, int W, int C, int mode, float scale,
                               oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C % 8 == 0 && (mode == 1 || (H % 2 == 0 && W % 2 == 0)),
                   "resample: bad arguments");
  const long long npix_out = (mode == 0) ? N * (H / 2) * (W / 2) : N * (H * 2LL) * (W * 2);
  const long long n8 = npix_out * (C / 8);
  long long nb = (n8 + 255) / 256;
  if (nb > 16384) nb = 16384;
  ONIRIS_KLAUNCH(resample_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)in, (bf16*)out, (const bf16*)add, n8,
                     H, W, C, mode, scale);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// resample with a general separable filter (reference utils.py:94-107: f of even length L, normalised to sum 1, pad = (L-1)/2):
//   mode 0 "down": depthwise conv2d with outer(f, f), stride 2, zero padding pad           (H, W = INPUT size)
//   mode 1 "up":   depthwise conv_transpose2d with 4 * outer(f, f), stride 2, padding pad  (H, W = INPUT size)
// `scale` multiplies the result, `add` (optional, shaped like out) is added: as in oniris_resample the adjoints are the other
// mode with another scale (down^T = up * 0.25, up^T = down * 4).  One thread per (output pixel, 8 channels), fp32 sums, one
// bf16 rounding.  [1, 1] gives exactly oniris_resample's 2x2 mean / nearest x2; the networks of the BASELINE configurations
// use only that one (networks_edm2.py:26), so this pass is about completeness, not the roofline.
struct ResampleTaps { float f[8]; int L; };

__global__ void resample_filter_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, const bf16* __restrict__ add,
                                       long long nout8, int H, int W, int C, int mode, float scale, ResampleTaps tp) {
  const int G = C >> 3, L = tp.L, pad = (L - 1) / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nout8; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    long long p = i / G;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    if (mode == 0) {
      const int Wo = W >> 1, Ho = H >> 1;
      const int xo = (int)(p % Wo); p /= Wo;
      const int yo = (int)(p % Ho); const long long n = p / Ho;
      for (int a_ = 0; a_ < L; ++a_) {
        const int y = 2 * yo + a_ - pad;
        if ((unsigned)y >= (unsigned)H) continue;
        for (int b_ = 0; b_ < L; ++b_) {
          const int x = 2 * xo + b_ - pad;
          if ((unsigned)x >= (unsigned)W) continue;
          const float w = tp.f[a_] * tp.f[b_];
          const bf16x8 t = *(const bf16x8*)(in + ((n * H + y) * W + x) * C + cg * 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += w * bf2f(t[k]);
        }
      }
    } else {
      const int Wo = W << 1, Ho = H << 1;
      const int X = (int)(p % Wo); p /= Wo;
      const int Y = (int)(p % Ho); const long long n = p / Ho;
      for (int a_ = (Y + pad) & 1; a_ < L; a_ += 2) {          // taps whose input row (Y + pad - a) / 2 is an integer
        const int y = (Y + pad - a_) >> 1;
        if (Y + pad - a_ < 0 || y >= H) continue;
        for (int b_ = (X + pad) & 1; b_ < L; b_ += 2) {
          const int x = (X + pad - b_) >> 1;
          if (X + pad - b_ < 0 || x >= W) continue;
          const float w = 4.f * tp.f[a_] * tp.f[b_];
          const bf16x8 t = *(const bf16x8*)(in + ((n * H + y) * W + x) * C + cg * 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += w * bf2f(t[k]);
        }
      }
    }
    bf16x8 o;
    if (add) {
      const bf16x8 d = *(const bf16x8*)(add + i * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k] * scale + bf2f(d[k]));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k] * scale);
    }
    *(bf16x8*)(out + i * 8) = o;
  }
}

extern "C" int oniris_resample_filter(const void* in, void* out, const void* add, int64_t N, int H, int W, int C, int mode,
                                      const float* taps, int ntaps, float scale, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(in && out && taps && N > 0 && H > 0 && W > 0 && C % 8 == 0 && (mode == 0 || mode == 1) &&
                   (mode == 1 || (H % 2 == 0 && W % 2 == 0)), "resample_filter: bad arguments");
  ONIRIS_CHECK_ARG(ntaps >= 2 && ntaps <= 8 && ntaps % 2 == 0, "resample_filter: an even number of taps, 2 .. 8 (got %d)", ntaps);
  ResampleTaps tp;
  for (int i = 0; i < 8; ++i) tp.f[i] = i < ntaps ? taps[i] : 0.f;
  tp.L = ntaps;
  const long long npix_out = (mode == 0) ? N * (H / 2) * (W / 2) : N * (H * 2LL) * (W * 2);
  const long long n8 = npix_out * (C / 8);
  long long nb = (n8 + 255) / 256;
  if (nb > 16384) nb = 16384;
  ONIRIS_KLAUNCH(resample_filter_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)in, (bf16*)out,
                     (const bf16*)add, n8, H, W, C, mode, scale, tp);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// DART training input / loss (edm2/loss.py:17-47 + Precond.forward, networks_edm2.py:278-297) without the ~25
// activation-sized fp32 passes of the eager formulation.  Slot n = (b, s, t), s = 0 clean | 1 noised (S = 1: 2-D
// steps); x[n] = images[b,t] + sigma[b, s*T+t] * noise[b, s*T+t]  (NCHW fp32 inputs, never materialised).
//   dart_input:   UNet input c_in*x, channels-last bf16, channel C = 1 (the reference's ones channel), rest 0
//   dart_loss:    per noised frame  mean_{c,h,w} (c_skip*x + c_out*out_gain*F - images)^2   (one block per frame)
//   dart_loss_bwd: dF (bf16, zero for the clean slots) and per-frame partial sums of d out_gain
__global__ __launch_bounds__(256) void dart_input_kernel(const float* __restrict__ img, const float* __restrict__ noise,
                                                         const float* __restrict__ sigma, bf16* __restrict__ xcl, int S,
                                                         int T, int C, int HW, float sd, float* __restrict__ c_noise_out, int cpad) {
  const int n = blockIdx.y, b = n / (S * T), st = n % (S * T), t = st % T;
  const float sg = sigma[b * S * T + st];
  if (c_noise_out && blockIdx.x == 0 && threadIdx.x == 0) c_noise_out[n] = logf(sg) / 4.f;     // c_noise (networks_edm2.py:291)
  const float cin = 1.f / sqrtf(sd * sd + sg * sg);
  const float* ip = img + (size_t)(b * T + t) * C * HW;
  const bool has_noise = noise != nullptr;        // NULL: no noise term (Precond's input side in eval)
  const float* np_ = has_noise ? noise + (size_t)(b * S * T + st) * C * HW : ip;      // (aliases ip: never a null load)
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    bf16 o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float v = 0.f;
      if (c < C) {
        const float xi = ip[(size_t)c * HW + p];
        v = has_noise ? cin * __fadd_rn(xi, __fmul_rn(sg, np_[(size_t)c * HW + p])) : cin * xi;
      }
      else if (c == C) v = 1.f;
      o[c] = f2bf(v);
    }
    uint4* dst = (uint4*)(xcl + ((size_t)n * HW + p) * cpad);
    dst[0] = *(const uint4*)&o[0];
    dst[1] = *(const uint4*)&o[8];
    for (int c8 = 2; c8 < cpad / 8; ++c8) dst[c8] = make_uint4(0u, 0u, 0u, 0u);     // (cpad = 32: the widths the streaming kernels take)
  }
}

// (1024 threads per block: only the noised half of the frames does work -- 128 blocks at the gym shape --, so the width of
// a block is what keeps loads in flight; the frame's sum stays ONE block's deterministic reduction)
template <bool BWD>
__global__ __launch_bounds__(1024) void dart_loss_kernel(const bf16* __restrict__ F, const float* __restrict__ img,
                                                        const float* __restrict__ noise, const float* __restrict__ sigma,
                                                        const float* __restrict__ out_gain, const float* __restrict__ g,
                                                        float* __restrict__ losses, bf16* __restrict__ dF,
                                                        float* __restrict__ dgain, int S, int T, int C, int HW, float sd) {
  __shared__ float red[16];
  const int n = blockIdx.x, b = n / (S * T), st = n % (S * T), s = st / T, t = st % T;
  if (s != S - 1) {                               // clean half: not part of the loss (loss.py:38 uses out[:, -T:])
    if (BWD)
      for (int p = threadIdx.x; p < HW; p += (int)blockDim.x) *(uint4*)(dF + ((size_t)n * HW + p) * 8) = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const float sg = sigma[b * S * T + st], og = out_gain[0];
  const float den = sg * sg + sd * sd;
  const float cskip = sd * sd / den, cout = sg * sd / sqrtf(den);
  const float* ip = img + (size_t)(b * T + t) * C * HW;
  const float* np_ = noise + (size_t)(b * S * T + st) * C * HW;
  const float inv = 1.f / (float)(C * HW);
  const float gl = BWD ? g[b * T + t] * 2.f * inv : 0.f;
  float acc = 0.f;
  for (int p = threadIdx.x; p < HW; p += (int)blockDim.x) {
    const uint4 fv = *(const uint4*)(F + ((size_t)n * HW + p) * 8);
    const bf16* f = (const bf16*)&fv;
    bf16 o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float d = 0.f;
      if (c < C) {
        const float im = ip[(size_t)c * HW + p];
        const float x = __fadd_rn(im, __fmul_rn(sg, np_[(size_t)c * HW + p]));
        const float fo = bf2f(f[c]);
        const float e = cskip * x + cout * (fo * og) - im;
        if (BWD) { d = gl * e * cout; acc += d * fo; d *= og; }
        else acc += e * e;
      }
      o[c] = f2bf(d);
    }
    if (BWD) *(uint4*)(dF + ((size_t)n * HW + p) * 8) = *(const uint4*)&o[0];
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    if (BWD) dgain[b * T + t] = acc;
    else losses[b * T + t] = acc * inv;
  }
}

extern "C" int oniris_dart_input(const float* images, const float* noise, const float* sigma, void* xcl, int B, int S,
                                 int T, int C, int H, int W, float sigma_data, float* c_noise_out, int cpad, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(images && sigma && xcl && B > 0 && (S == 1 || S == 2) && T > 0 && C > 0 && C < 16 && H > 0 && W > 0,
                   "dart_input: bad arguments");
  ONIRIS_CHECK_ARG(cpad >= 16 && cpad % 8 == 0 && cpad <= 64, "dart_input: cpad must be 16 ... 64 and a multiple of 8 (got %d)", cpad);
  const int HW = H * W;
  int gx = cdiv(HW, 256);
  if (gx > 64) gx = 64;
  ONIRIS_KLAUNCH(dart_input_kernel, dim3(gx, B * S * T), dim3(256), 0, stream, images, noise, sigma, (bf16*)xcl, S, T, C,
                     HW, sigma_data, c_noise_out, cpad);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_dart_loss(const void* F, const float* images, const float* noise, const float* sigma,
                                const float* out_gain, float* losses, int B, int S, int T, int C, int H, int W,
                                float sigma_data, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(F && images && noise && sigma && out_gain && losses && B > 0 && (S == 1 || S == 2) && T > 0 && C > 0 &&
                   C <= 8 && H > 0 && W > 0, "dart_loss: bad arguments");
  ONIRIS_KLAUNCH(dart_loss_kernel<false>, dim3(B * S * T), dim3(1024), 0, stream, (const bf16*)F, images, noise, sigma,
                     out_gain, (const float*)nullptr, losses, (bf16*)nullptr, (float*)nullptr, S, T, C, H * W, sigma_data);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_dart_loss_bwd(const void* F, const float* images, const float* noise, const float* sigma,
                                    const float* out_gain, const float* dlosses, void* dF, float* dgain_part, int B, int S,
                                    int T, int C, int H, int W, float sigma_data, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(F && images && noise && sigma && out_gain && dlosses && dF && dgain_part && B > 0 && (S == 1 || S == 2) &&
                   T > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "dart_loss_bwd: bad arguments");
  ONIRIS_KLAUNCH(dart_loss_kernel<true>, dim3(B * S * T), dim3(1024), 0, stream, (const bf16*)F, images, noise, sigma,
                     out_gain, dlosses, (float*)nullptr, (bf16*)dF, dgain_part, S, T, C, H * W, sigma_data);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Tail of EDM2Loss.__call__ (edm2/loss.py:32-46) + MultiNoiseLoss.add_data (edm2/loss_weight.py:30-39) in one launch, no host
// round trip: per (sequence, frame) of the noised half
//   l    = mse * (sigma^2 + sd^2) / (sigma * sd)^2                                   (loss.py:34-38)
//   m    = 10 ^ (c0/2 + sum_n c[2n-1] cos(n log10 sigma) + c[2n] sin(n log10 sigma))  (loss_weight.py:104-111,126-131)
//   out[0] = mean(l / m), out[1] = mean(l) (the un-weighted loss the loops log), dcoef = d out[0] / d mse = w / (m * B*T)
// and (sigma, l, t) appended to the 10 000-entry history the Fourier fit reads (rings + a device-side write counter:
// entry number `count + i` lives in slot (count + i) % cap, so the last `cap` entries are always present).
__global__ __launch_bounds__(256) void loss_tail_kernel(const float* __restrict__ mse, const float* __restrict__ sigma,
                                                        const float* __restrict__ coef, float* __restrict__ out,
                                                        float* __restrict__ dcoef, float* __restrict__ ring_sigma,
                                                        float* __restrict__ ring_loss, int* __restrict__ ring_pos,
                                                        long long* __restrict__ count, int cap, int n, int T,
                                                        int sig_pitch, int sig_off, int nterms, float sd) {
  __shared__ float red[16];
  const long long base = count ? *count : 0;
  float a0 = 0.f, a1 = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int b = i / T, t = i - b * T;
    const float sg = sigma[(size_t)b * sig_pitch + sig_off + t];
    const float w = (sg * sg + sd * sd) / ((sg * sd) * (sg * sd));
    const float l = mse[i] * w;
    const float xl = log10f(sg);
    float e = 0.5f * coef[0];
    for (int k = 1; k < nterms; ++k) e += coef[2 * k - 1] * cosf(k * xl) + coef[2 * k] * sinf(k * xl);
    const float inv_m = exp10f(-e);
    a0 += l * inv_m;
    a1 += l;
    dcoef[i] = w * inv_m / (float)n;
    if (ring_sigma && i >= n - cap) {
      const int j = (int)((base + i) % cap);
      ring_sigma[j] = sg;
      ring_loss[j] = l;
      ring_pos[j] = t;
    }
  }
  a0 = block_sum(a0, red);
  a1 = block_sum(a1, red);
  if (threadIdx.x == 0) {
    out[0] = a0 / (float)n;
    out[1] = a1 / (float)n;
    if (count && ring_sigma) *count = base + n;
  }
}

extern "C" int oniris_loss_tail(const float* mse, const float* sigma, const float* coef, float* out, float* dcoef,
                                float* ring_sigma, float* ring_loss, int* ring_pos, long long* count, int cap, int B, int T,
                                int sig_pitch, int sig_off, int nterms, float sigma_data, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(mse && sigma && coef && out && dcoef && B > 0 && T > 0 && sig_pitch >= sig_off + T && sig_off >= 0 &&
                   nterms >= 1 && sigma_data > 0.f, "loss_tail: bad arguments");
  ONIRIS_CHECK_ARG(!ring_sigma || (ring_loss && ring_pos && count && cap > 0), "loss_tail: incomplete history ring");
  ONIRIS_KLAUNCH(loss_tail_kernel, dim3(1), dim3(256), 0, stream, mse, sigma, coef, out, dcoef, ring_sigma, ring_loss,
                     ring_pos, count, cap, B * T, T, sig_pitch, sig_off, nterms, sigma_data);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Precond.forward's output side in eval (networks_edm2.py:293-297): D = c_skip * x + c_out * out_gain * F, with F the raw
// channels-last bf16 UNet output [N][HW][8] and x, D fp32 [N][C][HW] (one block column per frame)
__global__ __launch_bounds__(256) void precond_out_kernel(const bf16* __restrict__ F, const float* __restrict__ x,
                                                          const float* __restrict__ sigma, const float* __restrict__ out_gain,
                                                          float* __restrict__ D, int C, int HW, float sd) {
  const int n = blockIdx.y;
  const float sg = sigma[n], og = out_gain[0];
  const float den = sg * sg + sd * sd;
  const float cskip = sd * sd / den, cout = sg * sd / sqrtf(den) * og;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const bf16x8 f = *(const bf16x8*)(F + ((size_t)n * HW + p) * 8);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) {
        const size_t o = ((size_t)n * C + c) * HW + p;
        D[o] = cskip * x[o] + cout * bf2f(f[c]);
      }
  }
}

extern "C" int oniris_precond_out(const void* F, const float* x, const float* sigma, const float* out_gain, float* D,
                                  int N, int C, int H, int W, float sigma_data, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(F && x && sigma && out_gain && D && N > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "precond_out: bad arguments");
  int gx = cdiv(H * W, 256);
  if (gx > 64) gx = 64;
  ONIRIS_KLAUNCH(precond_out_kernel, dim3(gx, N), dim3(256), 0, stream, (const bf16*)F, x, sigma, out_gain, D, C, H * W,
                     sigma_data);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// The sampler's update between two UNet evaluations (edm2/sampler.py:66-76 of the reference) in one launch, fp32:
//   mode 0 (Euler):  d = (x_hat - x_pred) / t_a ;  x_out = x_hat + dt * d ;  d_io <- d
//   mode 1 (Heun):   d' = (x_aux - x_pred) / t_a ; x_out = x_hat <- x_hat + dt * (0.5 * d_io + 0.5 * d')
// (same operations in the same order as the reference's tensor expressions).  sigma_buf (nsig floats, optional) receives
// the NEXT evaluation's sigma, so the graph replay that follows needs no fill kernel of its own.
__global__ __launch_bounds__(256) void sampler_update_kernel(int mode, float* x_hat, const float* __restrict__ x_pred,
                                                             float* __restrict__ d_io, const float* x_aux,   // (x_aux, x_out, x_hat
                                                             float* x_out, size_t n, float t_a, float dt,     //  may be one buffer)
                                                             float* __restrict__ sigma_buf, int nsig, float sigma_next) {
#pragma clang fp contract(off)     // every product and sum is rounded on its own, like the separate tensor operations
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (sigma_buf && i < (size_t)nsig) sigma_buf[i] = sigma_next;
  if (i >= n) return;
  const float xh = x_hat[i];
  if (mode == 0) {
    const float d = (xh - x_pred[i]) / t_a;
    d_io[i] = d;
    const float s_ = dt * d;
    x_out[i] = xh + s_;
  } else {
    const float dp = (x_aux[i] - x_pred[i]) / t_a;
    const float a_ = 0.5f * d_io[i], b_ = 0.5f * dp;
    const float m_ = dt * (a_ + b_);
    const float xn = xh + m_;
    x_hat[i] = xn;
    if (x_out != x_hat) x_out[i] = xn;
  }
}

extern "C" int oniris_sampler_update(int mode, float* x_hat, const float* x_pred, float* d_io, const float* x_aux, float* x_out,
                                     size_t n, float t_a, float dt, float* sigma_buf, int nsig, float sigma_next,
                                     oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG((mode == 0 || mode == 1) && x_hat && x_pred && d_io && x_out && n > 0 && t_a != 0.f && (mode == 0 || x_aux) &&
                   nsig >= 0 && (size_t)nsig <= n, "sampler_update: bad arguments");
  ONIRIS_KLAUNCH(sampler_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mode, x_hat, x_pred, d_io,
                     x_aux, x_out, n, t_a, dt, sigma_buf, nsig, sigma_next);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// All Gating modules of a net in one launch (edm2/conv.py:113-127, eval): for layer l and frame slot n
//   g = lo + (1 - lo) * hi * sigmoid(c_noise[n] * mult0 + off0 + log1p(n % T + nctx[l]) * mult1 + off1),
//   lo = sigmoid(min_gating), hi = sigmoid(max_gating);   ca = (1 - g) / sqrt((1 - g)^2 + g^2),  cb = g / sqrt(...)
// params [L][6] = mult0, mult1, off0, off1, min_gating, max_gating.  (Training keeps the torch autograd formulation.)
__global__ void gates_kernel(const float* __restrict__ c_noise, const float* __restrict__ params,
                             const int* __restrict__ nctx, float* __restrict__ ca, float* __restrict__ cb, int N, int T) {
  const int l = blockIdx.y;
  const float* p = params + l * 6;
  const float lo = 1.f / (1.f + expf(-p[4])), hi = 1.f / (1.f + expf(-p[5]));
  const float nc = nctx ? (float)nctx[l] : 0.f;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float pos = log1pf((float)(n % T) + nc);
    const float sv = c_noise[n] * p[0] + p[2] + pos * p[1] + p[3];
    const float g = lo + (1.f - lo) * hi * (1.f / (1.f + expf(-sv)));
    const float r = rsqrtf((1.f - g) * (1.f - g) + g * g);
    ca[(size_t)l * N + n] = (1.f - g) * r;
    cb[(size_t)l * N + n] = g * r;
  }
}

extern "C" int oniris_gates(const float* c_noise, const float* params, const int32_t* nctx, float* ca, float* cb, int L,
                            int N, int T, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(c_noise && params && ca && cb && L > 0 && N > 0 && T > 0, "gates: bad arguments");
  ONIRIS_KLAUNCH(gates_kernel, dim3(cdiv(N, 256) > 64 ? 64 : cdiv(N, 256), L), dim3(256), 0, stream, c_noise, params,
                     (const int*)nctx, ca, cb, N, T);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The UNet's embedding in eval (networks_edm2.py:204-216) in one launch, fp32 throughout:
//   f = sqrt(2) * cos(c_noise[n] * freqs + phases)                              (MPFourier, utils.py:139-150)
//   e = What_noise . f,   What = w / (eps + |w_row| / sqrt(fan)) / sqrt(fan)    (MPConv in eval, conv.py:14-21,36-42)
//   with labels: e = mp_sum(e, What_label[:, label[n]] * sqrt(L), t = 1/3)      (one-hot input)
//   emb[n] = silu(e) / 0.596  -> bf16 [N][cemb]
// One block per frame slot; a thread owns output channels co = tid, tid + 256, ...
__global__ __launch_bounds__(256) void embed_eval_kernel(const float* __restrict__ c_noise, const long long* __restrict__ labels,
                                                         const float* __restrict__ freqs, const float* __restrict__ phases,
                                                         const float* __restrict__ w_noise, const float* __restrict__ w_label,
                                                         bf16* __restrict__ emb, int cn, int cemb, int L) {
  __shared__ float four[512];
  const int n = blockIdx.x;
  const float x = c_noise[n];
  for (int i = threadIdx.x; i < cn; i += 256) four[i] = 1.4142135623730951f * cosf(x * freqs[i] + phases[i]);
  __syncthreads();
  const long long lab = (labels && w_label) ? labels[n] : -1;
  for (int co = threadIdx.x; co < cemb; co += 256) {
    const float* wr = w_noise + (size_t)co * cn;
    float dot = 0.f, ss = 0.f;
    for (int i = 0; i < cn; ++i) { const float w = wr[i]; dot += w * four[i]; ss += w * w; }
    const float rs = rsqrtf((float)cn);
    float e = dot * rs / (1e-4f + sqrtf(ss) * rs);
    if (lab >= 0 && lab < L) {
      const float* wl = w_label + (size_t)co * L;
      float sl = 0.f;
      for (int i = 0; i < L; ++i) sl += wl[i] * wl[i];
      const float rl = rsqrtf((float)L);
      const float el = wl[lab] * sqrtf((float)L) * rl / (1e-4f + sqrtf(sl) * rl);
      const float t = 1.f / 3.f;
      e = (e + (el - e) * t) * rsqrtf((1.f - t) * (1.f - t) + t * t);
    }
    emb[(size_t)n * cemb + co] = f2bf(e * sigmoid_fast(e) * (1.f / 0.596f));
  }
}

extern "C" int oniris_embed_eval(const float* c_noise, const int64_t* labels, const float* freqs, const float* phases,
                                 const float* w_noise, const float* w_label, void* emb, int N, int cnoise, int cemb,
                                 int label_dim, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(c_noise && freqs && phases && w_noise && emb && N > 0 && cnoise > 0 && cnoise <= 512 && cemb > 0,
                   "embed_eval: bad arguments (cnoise <= 512)");
  ONIRIS_CHECK_ARG(!labels || !w_label || label_dim > 0, "embed_eval: label_dim missing");
  ONIRIS_KLAUNCH(embed_eval_kernel, dim3(N), dim3(256), 0, stream, c_noise, (const long long*)labels, freqs, phases,
                     w_noise, w_label, (bf16*)emb, cnoise, cemb, label_dim);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Training-side conditioning prelude (gates, embedding, emb scales) as a handful of fused launches with explicit
// adjoints, instead of ~130 forward / ~150 backward torch elementwise launches on kilobyte-sized tensors per step
// (the host cannot enqueue those faster than the GPU drains them: ~2 ms of idle GPU per step around the optimizer).

// Adjoint of gates_kernel: dparams [L][6] (mult0, mult1, off0, off1, min_gating, max_gating) from dca, dcb [L][N].
//   d ca / d g = -g r^3,  d cb / d g = (1 - g) r^3   (r = ((1-g)^2 + g^2)^-1/2);   one block per layer.
__global__ __launch_bounds__(256) void gates_bwd_kernel(const float* __restrict__ c_noise, const float* __restrict__ params,
                                                        const int* __restrict__ nctx, const float* __restrict__ dca,
                                                        const float* __restrict__ dcb, float* __restrict__ dparams, int N, int T) {
  __shared__ float red[6][4];
  const int l = blockIdx.x;
  const float* p = params + l * 6;
  const float lo = 1.f / (1.f + expf(-p[4])), hi = 1.f / (1.f + expf(-p[5]));
  const float nc = nctx ? (float)nctx[l] : 0.f;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // dmult0, dmult1, d(off0 + off1), -, dlo, dhi
  for (int n = threadIdx.x; n < N; n += 256) {
    const float cn = c_noise[n];
    const float pos = log1pf((float)(n % T) + nc);
    const float sv = cn * p[0] + p[2] + pos * p[1] + p[3];
    const float s = 1.f / (1.f + expf(-sv));
    const float g = lo + (1.f - lo) * hi * s;
    const float r = rsqrtf((1.f - g) * (1.f - g) + g * g);
    const float r3 = r * r * r;
    const float dg = r3 * ((1.f - g) * dcb[(size_t)l * N + n] - g * dca[(size_t)l * N + n]);
    const float dsv = dg * (1.f - lo) * hi * s * (1.f - s);
    acc[0] += dsv * cn; acc[1] += dsv * pos; acc[2] += dsv;
    acc[4] += dg * (1.f - hi * s); acc[5] += dg * (1.f - lo) * s;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    float* o = dparams + l * 6;
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[2];
    o[4] = t[4] * lo * (1.f - lo); o[5] = t[5] * hi * (1.f - hi);
  }
}

extern "C" int oniris_gates_bwd(const float* c_noise, const float* params, const int32_t* nctx, const float* dca,
                                const float* dcb, float* dparams, int L, int N, int T, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(c_noise && params && dca && dcb && dparams && L > 0 && N > 0 && T > 0, "gates_bwd: bad arguments");
  ONIRIS_KLAUNCH(gates_bwd_kernel, dim3(L), dim3(256), 0, stream, c_noise, params, (const int*)nctx, dca, dcb, dparams, N, T);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// c[n][j] = 1 + c_all[n][j] * gain[seg[j]]   (networks_edm2.py:78, every Block at once: c_all = the row-concatenated
// emb_linear GEMM [N][Ctot] bf16, seg[j] = the Block column j belongs to, gain = the Blocks' emb_gain); fp32 out.
__global__ void emb_scale_kernel(const bf16* __restrict__ c_all, const float* __restrict__ gain, const int* __restrict__ seg,
                                 float* __restrict__ c, int N, int Ctot) {
  const size_t total = (size_t)N * Ctot;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % Ctot);
    c[i] = 1.f + bf2f(c_all[i]) * gain[seg[j]];
  }
}

// adjoint: dc_all = dc * gain[seg] (bf16), dgain_part[k][chunk] = sum over the chunk's rows n and the columns
// [start[k], start[k+1]) of dc * c_all; grid (Block k, row chunk): the caller adds the EMB_BWD_CHUNKS partial sums per k
// (no atomics: deterministic).  One block per k walked 65 K elements serially (99 us for 3.8 MB at the gym size).
#define EMB_BWD_CHUNKS ONIRIS_EMB_BWD_CHUNKS
__global__ __launch_bounds__(256) void emb_scale_bwd_kernel(const float* __restrict__ dc, const bf16* __restrict__ c_all,
                                                            const float* __restrict__ gain, const int* __restrict__ start,
                                                            bf16* __restrict__ dc_all, float* __restrict__ dgain_part, int N, int Ctot) {
  __shared__ float red[4];
  const int k = blockIdx.x, ch = blockIdx.y;
  const int j0 = start[k], w = start[k + 1] - j0;
  const int n0 = (int)((long long)N * ch / EMB_BWD_CHUNKS), n1 = (int)((long long)N * (ch + 1) / EMB_BWD_CHUNKS);
  const float g = gain[k];
  float acc = 0.f;
  for (int i = threadIdx.x; i < (n1 - n0) * w; i += 256) {
    const int n = n0 + i / w, j = j0 + i % w;
    const size_t at = (size_t)n * Ctot + j;
    const float d = dc[at];
    acc += d * bf2f(c_all[at]);
    dc_all[at] = f2bf(d * g);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dgain_part[k * EMB_BWD_CHUNKS + ch] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int oniris_emb_scale(const void* c_all, const float* gain, const int32_t* seg, float* c, int N, int Ctot,
                                oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(c_all && gain && seg && c && N > 0 && Ctot > 0, "emb_scale: bad arguments");
  const size_t total = (size_t)N * Ctot;
  ONIRIS_KLAUNCH(emb_scale_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)c_all, gain, (const int*)seg, c, N, Ctot);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_emb_scale_bwd(const float* dc, const void* c_all, const float* gain, const int32_t* start, void* dc_all,
                                    float* dgain_part, int N, int Ctot, int K, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(dc && c_all && gain && start && dc_all && dgain_part && N > 0 && Ctot > 0 && K > 0, "emb_scale_bwd: bad arguments");
  ONIRIS_KLAUNCH(emb_scale_bwd_kernel, dim3(K, EMB_BWD_CHUNKS), dim3(256), 0, stream, dc, (const bf16*)c_all, gain,
                     (const int*)start, (bf16*)dc_all, dgain_part, N, Ctot);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// Inputs of the two embedding linears (networks_edm2.py:204-212): four [N][cnP] bf16 = sqrt(2) cos(c_noise f + phi)
// (MPFourier, utils.py:139-150; zero-padded to cnP), onehot [N][LP] bf16 = sqrt(L) at the label, else 0 (NULL: no labels).
__global__ void embed_pre_kernel(const float* __restrict__ c_noise, const long long* __restrict__ labels,
                                 const float* __restrict__ freqs, const float* __restrict__ phases, bf16* __restrict__ four,
                                 bf16* __restrict__ onehot, int N, int cn, int cnP, int L, int LP) {
  const int n = blockIdx.x;
  const float x = c_noise[n];
  for (int i = threadIdx.x; i < cnP; i += blockDim.x)
    four[(size_t)n * cnP + i] = f2bf(i < cn ? 1.4142135623730951f * cosf(x * freqs[i] + phases[i]) : 0.f);
  if (onehot) {
    const long long lab = labels[n];
    for (int i = threadIdx.x; i < LP; i += blockDim.x) onehot[(size_t)n * LP + i] = f2bf(i == lab ? sqrtf((float)L) : 0.f);
  }
}

extern "C" int oniris_embed_pre(const float* c_noise, const int64_t* labels, const float* freqs, const float* phases, void* four,
                                void* onehot, int N, int cnoise, int cnoiseP, int label_dim, int labelP, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(c_noise && freqs && phases && four && N > 0 && cnoise > 0 && cnoiseP >= cnoise, "embed_pre: bad arguments");
  ONIRIS_CHECK_ARG(!onehot || (labels && label_dim > 0 && labelP >= label_dim), "embed_pre: labels missing");
  ONIRIS_KLAUNCH(embed_pre_kernel, dim3(N), dim3(64), 0, stream, c_noise, (const long long*)labels, freqs, phases,
                     (bf16*)four, (bf16*)onehot, N, cnoise, cnoiseP, label_dim, labelP);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// emb = mp_silu(mp_sum(e1, e2, t)) (utils.py:118-123,101-102; e2 NULL: emb = mp_silu(e1)), bf16 in / out, fp32 inside;
// the adjoint recomputes the pre-activation from e1, e2.
__global__ void embed_post_kernel(const bf16* __restrict__ e1, const bf16* __restrict__ e2, bf16* __restrict__ emb, size_t n,
                                  float t) {
  const float den = rsqrtf((1.f - t) * (1.f - t) + t * t);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float e = bf2f(e1[i]);
    if (e2) e = (e + (bf2f(e2[i]) - e) * t) * den;
    emb[i] = f2bf(e * sigmoid_fast(e) * (1.f / 0.596f));
  }
}

__global__ void embed_post_bwd_kernel(const bf16* __restrict__ demb, const bf16* __restrict__ e1, const bf16* __restrict__ e2,
                                      bf16* __restrict__ de1, bf16* __restrict__ de2, size_t n, float t) {
  const float den = rsqrtf((1.f - t) * (1.f - t) + t * t);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float e = bf2f(e1[i]);
    if (e2) e = (e + (bf2f(e2[i]) - e) * t) * den;
    const float s = sigmoid_fast(e);
    const float de = bf2f(demb[i]) * (1.f / 0.596f) * s * (1.f + e * (1.f - s));
    if (e2) { de1[i] = f2bf(de * (1.f - t) * den); de2[i] = f2bf(de * t * den); }
    else de1[i] = f2bf(de);
  }
}

extern "C" int oniris_embed_post(const void* e1, const void* e2, void* emb, size_t n, float t, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(e1 && emb && n > 0, "embed_post: bad arguments");
  ONIRIS_KLAUNCH(embed_post_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)e1, (const bf16*)e2, (bf16*)emb, n, t);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_embed_post_bwd(const void* demb, const void* e1, const void* e2, void* de1, void* de2, size_t n, float t,
                                     oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(demb && e1 && de1 && n > 0 && (!e2 || de2), "embed_post_bwd: bad arguments");
  ONIRIS_KLAUNCH(embed_post_bwd_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)demb, (const bf16*)e1, (const bf16*)e2, (bf16*)de1, (bf16*)de2, n, t);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
