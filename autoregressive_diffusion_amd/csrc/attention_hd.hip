// Attention heads of 8, 16, 24, ... 56 channels (every multiple of 8 below 64; round 4: 8 / 16 / 32) (Block(channels_per_head=...), reference networks_edm2.py:28; the reference's own
// consistency tests build their nets with 16: edm2/consistency_test.py:39,61).
//
// The attention kernels of this library are written for 64-channel heads (every BASELINE configuration).  Other head
// sizes are served by PADDING: q, k, v of a d-channel head are laid out as a 64-channel head whose channels d..63 are zero
// -- q.k is unchanged by zeros, P.V leaves zeros in the padding -- so every attention kernel (training forward / backward,
// dense per-frame, causal prefill, decode) runs unmodified on (tokens, heads * 64) tensors; the softmax scale 1 / sqrt(d)
// instead of 1 / 8 rides on q.  What IS specific to d lives here: the per-head pixel norm over d channels
// (attention_modules.py:48-49 / utils.py:83-88), the rotary embedding with its rotation partner d / 2 channels away
// (RoPe.py:34-57) and their adjoints.  One thread per (token, q|k|v, head); this path is about generality (test-sized
// nets), not about the roofline.
#include "common.h"
#include "../../include/oniris.h"

#define HD_SCALE_LOG2E 1.4426950408889634f

// rope: bit 0 = rotate q, bit 1 = rotate k (tables [pos][d], fp32, the fp16-rounded values of ops.rope_tables);
// position of a token = ((token / P) % seq_frames + pos_off) % pos_mod   (seq_frames: frames per sequence of the tensor)
template <int D>
__global__ void qkv_norm_hd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ q, bf16* __restrict__ k,
                                   bf16* __restrict__ v, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                   const float* __restrict__ scale_t, long long nitem, int heads, int P, int pos_mod,
                                   int pos_off, int rope, int seq_frames) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nitem) return;
  const int hd = (int)(gid % heads), s = (int)((gid / heads) % 3);
  const long long tok = gid / (3 * heads);
  const bf16* src = qkv + (tok * 3 + s) * heads * D + hd * D;
  float f[D], ss = 0.f;
#pragma unroll
  for (int i = 0; i < D; i += 8) {
    const bf16x8 x = *(const bf16x8*)(src + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[i + j] = bf2f(x[j]); ss += f[i + j] * f[i + j]; }
  }
  // q carries log2(e) / sqrt(d): the attention kernels take their log2-domain scores straight from q'.k
  const float inv = ((s == 0) ? HD_SCALE_LOG2E * rsqrtf((float)D) : 1.f) / (1e-4f + sqrtf(ss) * rsqrtf((float)D));
  float u[D];
#pragma unroll
  for (int i = 0; i < D; ++i) u[i] = f[i] * inv;
  const bool rot = (s == 0 && (rope & 1)) || (s == 1 && (rope & 2));
  if (rot) {
    const size_t tb = (size_t)(((tok / P) % seq_frames + pos_off) % pos_mod) * D;
    float w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const float partner = (i < D / 2) ? -u[i + D / 2] : u[i - D / 2];            // rotate_half: [-x2, x1]
      const float val = u[i] * cos_t[tb + i] + partner * sin_t[tb + i];
      w[i] = (s == 0) ? val * scale_t[tb + i] : val / scale_t[tb + i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) u[i] = w[i];
  }
  bf16* dst = ((s == 0) ? q : (s == 1) ? k : v) + (tok * heads + hd) * 64;
#pragma unroll
  for (int i = 0; i < 64; i += 8) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf((i + j < D) ? u[(i + j < D) ? i + j : 0] : 0.f);
    *(bf16x8*)(dst + i) = o;
  }
}

// adjoint: dq (w.r.t. the q the attention backward believes in: the 64-channel-head model q.k / 8, i.e. alpha * q_d with
// alpha = 8 / sqrt(d)), dk, dv in the padded layout -> dqkv [tokens][3 * heads * d]
template <int D>
__global__ void qkv_norm_hd_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dq,
                                       const bf16* __restrict__ dk, const bf16* __restrict__ dv, bf16* __restrict__ dqkv,
                                       const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                       const float* __restrict__ scale_t, long long nitem, int heads, int P, int pos_mod,
                                       int pos_off, int rope, int seq_frames) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nitem) return;
  const int hd = (int)(gid % heads), s = (int)((gid / heads) % 3);
  const long long tok = gid / (3 * heads);
  const bf16* src = qkv + (tok * 3 + s) * heads * D + hd * D;
  const bf16* gsrc = ((s == 0) ? dq : (s == 1) ? dk : dv) + (tok * heads + hd) * 64;
  float f[D], g[D], ss = 0.f;
#pragma unroll
  for (int i = 0; i < D; i += 8) {
    const bf16x8 x = *(const bf16x8*)(src + i), gv = *(const bf16x8*)(gsrc + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[i + j] = bf2f(x[j]); g[i + j] = bf2f(gv[j]); ss += f[i + j] * f[i + j]; }
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] *= 8.f * rsqrtf((float)D);
  }
  const bool rot = (s == 0 && (rope & 1)) || (s == 1 && (rope & 2));
  if (rot) {
    const size_t tb = (size_t)(((tok / P) % seq_frames + pos_off) % pos_mod) * D;
    float gs[D], gc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const float gv = (s == 0) ? g[i] * scale_t[tb + i] : g[i] / scale_t[tb + i];
      gc[i] = gv * cos_t[tb + i];
      gs[i] = gv * sin_t[tb + i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i)            // adjoint of rotate_half: channel i receives +(g sin)[i + d/2] resp. -(g sin)[i - d/2]
      g[i] = gc[i] + ((i < D / 2) ? gs[i + D / 2] : -gs[i - D / 2]);
  }
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) dot += f[i] * g[i];
  const float rd = rsqrtf((float)D), n = sqrtf(ss), sden = 1e-4f + n * rd;
  const float k1 = 1.f / sden, k2 = (n > 0.f) ? dot * rd / (sden * sden * n) : 0.f;
  bf16* dst = dqkv + (tok * 3 + s) * heads * D + hd * D;
#pragma unroll
  for (int i = 0; i < D; i += 8) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(g[i + j] * k1 - f[i + j] * k2);
    *(bf16x8*)(dst + i) = o;
  }
}

// rotary embedding of an already normalised, padded tensor (eval: every cached key is re-rotated for the grown key count,
// RoPe.py:55-57): mode 1 = q (times scale), 2 = k (divided by scale); one thread per (token, head)
template <int D>
__global__ void rope_hd_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, const float* __restrict__ cos_t,
                               const float* __restrict__ sin_t, const float* __restrict__ scale_t, long long nitem, int heads,
                               int P, int pos_mod, int pos_off, int mode, int seq_frames) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nitem) return;
  const long long tok = gid / heads;
  const bf16* src = x + gid * 64;
  float u[D];
#pragma unroll
  for (int i = 0; i < D; i += 8) {
    const bf16x8 xv = *(const bf16x8*)(src + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) u[i + j] = bf2f(xv[j]);
  }
  const size_t tb = (size_t)(((tok / P) % seq_frames + pos_off) % pos_mod) * D;
  bf16* dst = out + gid * 64;
#pragma unroll
  for (int i = 0; i < 64; i += 8) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = i + j;
      float val = 0.f;
      if (c < D) {
        const int cc = (c < D) ? c : 0;
        const float partner = (cc < D / 2) ? -u[(cc + D / 2) % D] : u[(cc + D - D / 2) % D];
        val = u[cc] * cos_t[tb + cc] + partner * sin_t[tb + cc];
        val = (mode == 1) ? val * scale_t[tb + cc] : val / scale_t[tb + cc];
      }
      o[j] = f2bf(val);
    }
    *(bf16x8*)(dst + i) = o;
  }
}

#define HD_DISPATCH(D_, ...)                                 \
  switch (D_) {                                              \
    case 8: { constexpr int D = 8; __VA_ARGS__; } break;     \
    case 16: { constexpr int D = 16; __VA_ARGS__; } break;   \
    case 24: { constexpr int D = 24; __VA_ARGS__; } break;   \
    case 32: { constexpr int D = 32; __VA_ARGS__; } break;   \
    case 40: { constexpr int D = 40; __VA_ARGS__; } break;   \
    case 48: { constexpr int D = 48; __VA_ARGS__; } break;   \
    case 56: { constexpr int D = 56; __VA_ARGS__; } break;   \
    default: oniris_set_error("attention head dimension %d: a multiple of 8 below 64 (padded path) or 64", D_); return ONIRIS_EUNSUPPORTED; \
  }

extern "C" int oniris_qkv_norm_hd(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                                  const float* scale_t, int64_t n_tokens, int heads, int head_dim, int P, int pos_mod,
                                  int pos_off, int rope, int seq_frames, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && q && k && v && n_tokens > 0 && heads > 0 && P > 0 && pos_mod > 0 && pos_off >= 0,
                   "qkv_norm_hd: bad arguments");
  ONIRIS_CHECK_ARG(seq_frames > 0, "qkv_norm_hd: seq_frames");
  ONIRIS_CHECK_ARG(rope == 0 || (cos_t && sin_t && scale_t), "qkv_norm_hd: rotary tables missing");
  const long long nitem = (long long)n_tokens * 3 * heads;
  const dim3 grid((unsigned)((nitem + 127) / 128));
  HD_DISPATCH(head_dim, ONIRIS_KLAUNCH(qkv_norm_hd_kernel<D>, grid, dim3(128), 0, stream, (const bf16*)qkv, (bf16*)q,
                                           (bf16*)k, (bf16*)v, cos_t, sin_t, scale_t, nitem, heads, P, pos_mod, pos_off, rope, seq_frames))
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_norm_hd_bwd(const void* qkv, const void* dq, const void* dk, const void* dv, void* dqkv,
                                      const float* cos_t, const float* sin_t, const float* scale_t, int64_t n_tokens,
                                      int heads, int head_dim, int P, int pos_mod, int pos_off, int rope, int seq_frames,
                                      oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && dq && dk && dv && dqkv && n_tokens > 0 && heads > 0 && P > 0 && pos_mod > 0 && pos_off >= 0,
                   "qkv_norm_hd_bwd: bad arguments");
  ONIRIS_CHECK_ARG(seq_frames > 0, "qkv_norm_hd_bwd: seq_frames");
  ONIRIS_CHECK_ARG(rope == 0 || (cos_t && sin_t && scale_t), "qkv_norm_hd_bwd: rotary tables missing");
  const long long nitem = (long long)n_tokens * 3 * heads;
  const dim3 grid((unsigned)((nitem + 127) / 128));
  HD_DISPATCH(head_dim, ONIRIS_KLAUNCH(qkv_norm_hd_bwd_kernel<D>, grid, dim3(128), 0, stream, (const bf16*)qkv,
                                           (const bf16*)dq, (const bf16*)dk, (const bf16*)dv, (bf16*)dqkv, cos_t, sin_t,
                                           scale_t, nitem, heads, P, pos_mod, pos_off, rope, seq_frames))
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_rope_hd(const void* x, void* out, const float* cos_t, const float* sin_t, const float* scale_t,
                              int64_t n_tokens, int heads, int head_dim, int P, int pos_mod, int pos_off, int mode,
                              int seq_frames, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(x && out && cos_t && sin_t && scale_t && n_tokens > 0 && heads > 0 && P > 0 && pos_mod > 0 &&
                   pos_off >= 0 && (mode == 1 || mode == 2) && seq_frames > 0, "rope_hd: bad arguments");
  const long long nitem = (long long)n_tokens * heads;
  const dim3 grid((unsigned)((nitem + 127) / 128));
  HD_DISPATCH(head_dim, ONIRIS_KLAUNCH(rope_hd_kernel<D>, grid, dim3(128), 0, stream, (const bf16*)x, (bf16*)out, cos_t,
                                           sin_t, scale_t, nitem, heads, P, pos_mod, pos_off, mode, seq_frames))
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
