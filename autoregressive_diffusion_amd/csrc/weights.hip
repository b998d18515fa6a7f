// Forced weight normalisation + bf16 packing for every weight of the net in ONE launch, and its backward.
// Replaces NormalizedWeight.forward (reference edm2/conv.py:14-21) + the per-call `.to(x.dtype)` of MPConv
// (edm2/conv.py:37,63).  HBM-bound row reductions: one 256-thread workgroup per output-channel row,
// wave-shuffle + LDS reduce, fp32 math, coalesced row reads.
#include <cstdlib>
#include "common.h"
#include "../../include/oniris.h"

#define W_EPS 1e-4f

__device__ __forceinline__ const OnirisWeightDesc* find_desc(const OnirisWeightDesc* d, int n, int row) {
  int lo = 0, hi = n - 1;                     // last desc with row_start <= row
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (d[mid].row_start <= row) lo = mid; else hi = mid - 1;
  }
  return d + lo;
}

// packed row index of original output channel `co`  (qkv: (m c s) -> (s m c), see attention_modules.py:48)
__device__ __forceinline__ int perm_row(const OnirisWeightDesc* d, int co) {
  if (!d->perm3) return co;
  const int C = d->cout / 3;
  return (co % 3) * C + co / 3;
}

__device__ __forceinline__ const OnirisWeightDesc* find_desc_tile(const OnirisWeightDesc* d, int n, int tile) {
  int lo = 0, hi = n - 1;                     // last desc with tile_start <= tile
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (d[mid].tile_start <= tile) lo = mid; else hi = mid - 1;
  }
  return d + lo;
}

// Pass 1: one workgroup per output-channel row: the two normalisations (training: the first is written back) and the
// forward packing wf[tap][co][ci] -- written tap by tap with ci along the lanes (128-byte runs; the first version
// walked the parameter order and wrote 2-byte elements with a stride of CoutP*CinP).
#define WROW_TAPS 27         // LDS transpose rounds of the row kernels: WROW_LDS floats hold [taps][wrow_ch(taps) + 1]
#define WROW_LDS 4640
// input channels per round: as many as fit, so that a round's taps * ch / 8 sixteen-byte slab items keep all 256 lanes busy
__device__ __forceinline__ int wrow_ch(int taps) { return taps <= 9 ? 512 : taps <= 18 ? 256 : 128; }
__global__ __launch_bounds__(256) void weight_prep_kernel(const OnirisWeightDesc* descs, int ndesc, int training) {
  __shared__ float red[16];
  __shared__ float wrow_lds[WROW_LDS];
  const int row = blockIdx.x;
  const OnirisWeightDesc* d = find_desc(descs, ndesc, row);
  const int co = row - d->row_start;
  const int taps = d->taps, cin = d->cin;
  const int fan = cin * taps;
  float* w = d->w + (size_t)co * fan;
  const float rs = rsqrtf((float)fan);
  if (co == 0 && threadIdx.x == 0 && d->nsplit) *d->nsplit = 0;      // a new step: no weight-gradient slab is valid yet

  float ss = 0.f;
  for (int e = threadIdx.x; e < fan; e += 256) { float v = w[e]; ss += v * v; }
  ss = block_sum(ss, red);
  float inv1 = 1.f;
  if (training) {                                               // forced normalisation (stored back)
    const float den = W_EPS + sqrtf(ss) * rs;
    // at the fixed point |w|/sqrt(fan) = 1 - eps the map is the identity; in fp32 it is the identity up to the last
    // bits, and rewriting the parameters with those makes two training-mode forwards on the same input differ (the
    // bf16 roundings downstream amplify any perturbation to their own size within a few layers).  Within 3 ulp of 1
    // the row is left as it is: repeated forwards without an optimizer step are bit-reproducible.
    inv1 = (fabsf(den - 1.f) < 4e-7f) ? 1.f : 1.f / den;
  }
  float ss2 = 0.f;
  for (int e = threadIdx.x; e < fan; e += 256) {
    float v = w[e] * inv1;
    if (training) w[e] = v;
    ss2 += v * v;
  }
  ss2 = block_sum(ss2, red);          // (its barriers also order the write-back before the re-reads below)
  const float scale = d->gain * rs / (W_EPS + sqrtf(ss2) * rs) * (training ? 1.f : inv1);   // second normalise * gain/sqrt(fan_in)

  const int cop = perm_row(d, co);
  unsigned short* wf = (unsigned short*)d->wf;
  if (!wf) return;
  if (taps > 1 && taps <= WROW_TAPS && (cin & 7) == 0) {
    // [ci][tap] -> [tap][ci] through LDS, wrow_ch(taps) input channels at a time: the parameters are read in their own order
    // (consecutive addresses; the form below gathers 4-byte values with a stride of taps*4 bytes: ~64 cache lines per wave
    // instruction, which is what bounded this kernel on the long rows of the 310 M net) and the packed row is written 16
    // bytes per lane.  Row stride wrow_ch + 1 floats: the transposed accesses are conflict-free.
    const int CH = wrow_ch(taps);
    for (int c0 = 0; c0 < cin; c0 += CH) {
      const int cw = min(CH, cin - c0), n = cw * taps;
      if (c0) __syncthreads();
      for (int e = threadIdx.x; e < n; e += 256) {
        const int ci = e / taps, tap = e - ci * taps;
        wrow_lds[tap * (CH + 1) + ci] = w[c0 * taps + e] * scale;
      }
      __syncthreads();
      const int c8 = cw >> 3;
      for (int it = threadIdx.x; it < taps * c8; it += 256) {
        const int tap = it / c8, ci = (it - tap * c8) * 8;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(wrow_lds[tap * (CH + 1) + ci + j]);
        *(bf16x8*)(wf + ((size_t)tap * d->CoutP + cop) * d->CinP + c0 + ci) = o;
      }
    }
    return;
  }
  for (int q = threadIdx.x; q < fan; q += 256) {                  // q = tap*cin + ci: ci contiguous across the lanes
    const int tap = q / cin, ci = q - tap * cin;
    wf[((size_t)tap * d->CoutP + cop) * d->CinP + ci] = __builtin_bit_cast(unsigned short, f2bf(w[ci * taps + tap] * scale));
  }
}

// Pass 2: the dgrad packing wb[flipped tap][ci][co] is the transpose of wf: one workgroup per (32 packed rows, tap),
// 32 x 64 tiles through LDS, 128-byte reads and 64-byte writes.
__global__ __launch_bounds__(256) void weight_wb_kernel(const OnirisWeightDesc* descs, int ndesc) {
  __shared__ __attribute__((aligned(16))) unsigned short t_lds[32 * 72];
  const OnirisWeightDesc* d = find_desc_tile(descs, ndesc, blockIdx.x);
  const int tap = blockIdx.y;
  if (tap >= d->taps || d->wb == nullptr || d->wf == nullptr) return;
  const int cop0 = (blockIdx.x - d->tile_start) * 32;
  const int nrow = min(32, d->cout - cop0);
  const int per_t = d->kt > 0 ? d->taps / d->kt : d->taps;        // spatial taps per temporal slice (9 or 1)
  const int j = tap / per_t, k = tap - j * per_t;
  const int tb = j * per_t + (per_t - 1 - k);                     // spatially flipped tap, same temporal slice
  const unsigned short* wf = (const unsigned short*)d->wf + ((size_t)tap * d->CoutP + cop0) * d->CinP;
  unsigned short* wb = (unsigned short*)d->wb + (size_t)tb * d->CoutPb * d->CinPb + cop0;
  const int tid = threadIdx.x;
  const int lr = tid >> 3, lp = tid & 7;                          // load: 32 rows x 8 parts of 8 ci
  const int sc_ = tid >> 2, sp = tid & 3;                         // store: 64 ci x 4 parts of 8 rows
  for (int c0 = 0; c0 < d->cin; c0 += 64) {
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (lr < nrow && c0 + lp * 8 < d->CinP) v = *(const u32x4*)(wf + (size_t)lr * d->CinP + c0 + lp * 8);
    *(u32x4*)(t_lds + lr * 72 + lp * 8) = v;
    __syncthreads();
    if (c0 + sc_ < d->cin) {
      unsigned short o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = t_lds[(sp * 8 + r) * 72 + sc_];
      unsigned short* dst = wb + (size_t)(c0 + sc_) * d->CinPb + sp * 8;
      if (sp * 8 + 7 < nrow) *(u32x4*)dst = *(const u32x4*)o;
      else
        for (int r = 0; r < 8; ++r) if (sp * 8 + r < nrow) dst[r] = o[r];
    }
    __syncthreads();
  }
}

// grad(w_hat) of   W = gain/sqrt(f) * w_hat / (eps + |w_hat|/sqrt(f))   given dW (packed bf16 split-K slabs, from wgrad)
//
// One workgroup per output-channel row.  The slabs are laid out [co][tap][ci] (what the wgrad kernels produce), the
// parameter and its gradient [co][ci][tap]: the row is transposed on the way, wrow_ch(taps) input channels at a time through LDS
// (row stride wrow_ch + 1 floats: conflict-free both ways), so that the slabs are read 16 bytes per lane along ci AND the
// parameters / gradients are walked in their own order.  The first version walked w / grad in slab order (4-byte accesses
// with a stride of taps*4 bytes, ~64 cache lines per wave instruction on three of its four streams): 1.6 TB/s on the 310 M
// net, where this kernel was 18 % of the step.  The fp32 row sum G is parked in `dws` in PARAMETER order between the two
// passes (the second needs the row's dot product).  A full-row LDS image (no `dws` round trip) was tried and is slower:
// 56 KB of LDS per workgroup leave two workgroups per CU, and the kernel lives on loads in flight.
__global__ __launch_bounds__(256) void weight_bwd_kernel(const OnirisWeightDesc* descs, int ndesc, int chunk_max_nsp) {
  __shared__ float red[16];
  __shared__ float Gs[WROW_LDS];
  const int row = blockIdx.x;
  const OnirisWeightDesc* d = find_desc(descs, ndesc, row);
  if (d->dwp == nullptr || d->dws == nullptr || d->grad == nullptr) return;
  const int co = row - d->row_start;
  const int taps = d->taps, cin = d->cin;
  const int fan = cin * taps;
  const float* w = d->w + (size_t)co * fan;
  float* g = d->grad + (size_t)co * fan;
  const int cop = perm_row(d, co);
  const bf16* dwp = (const bf16*)d->dwp;
  float* dws = d->dws;
  const float rs = rsqrtf((float)fan);

  const int nsp = d->nsplit ? *d->nsplit : 1;
  if (nsp <= 0) return;                                         // no wgrad ran for this weight in this step
  const size_t slab = (size_t)taps * d->CoutP * d->CinP;
  // (chunk_max_nsp: A/B knob -- with 128-channel rounds weights with many slabs were faster in the form below, whose slab
  // reads keep every lane busy; with rounds sized by wrow_ch() the transposing form wins for every slab count)
  const bool chunked = (cin & 7) == 0 && taps <= WROW_TAPS && nsp <= chunk_max_nsp;
  float* gp = dws + (size_t)co * fan;                           // chunked form: G of this row in parameter order
  // pass 1 (the row of this workgroup is ONE contiguous run of taps * CinP values per slab): G = fp32 sum over the bf16
  // split-K slabs, dot = <G, w>, nn = |w|^2
  float dot = 0.f, nn = 0.f;
  if (chunked) {
    const int CH = wrow_ch(taps);
    for (int c0 = 0; c0 < cin; c0 += CH) {
      const int cw = min(CH, cin - c0), c8 = cw >> 3;
      if (c0) __syncthreads();
      // 16 bytes (8 slab values) per lane and up to 8 independent slab reads in flight
      for (int it = threadIdx.x; it < taps * c8; it += 256) {
        const int tap = it / c8, ci = (it - tap * c8) * 8;
        const size_t pi = ((size_t)cop * taps + tap) * d->CinP + c0 + ci;
        float G[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) G[u][j] = 0.f;
        int s_ = 0;
        for (; s_ + 7 < nsp; s_ += 8) {
          bf16x8 t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = *(const bf16x8*)(dwp + (size_t)(s_ + u) * slab + pi);
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) G[u & 3][j] += bf2f(t[u][j]);
        }
        for (; s_ < nsp; ++s_) {
          const bf16x8 t = *(const bf16x8*)(dwp + (size_t)s_ * slab + pi);
#pragma unroll
          for (int j = 0; j < 8; ++j) G[s_ & 3][j] += bf2f(t[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) Gs[tap * (CH + 1) + ci + j] = (G[0][j] + G[1][j]) + (G[2][j] + G[3][j]);
      }
      __syncthreads();
      const int n = cw * taps;
      for (int e = threadIdx.x; e < n; e += 256) {               // parameter order: e = ci*taps + tap
        const int ci = e / taps, tap = e - ci * taps;
        const float G = Gs[tap * (CH + 1) + ci], v = w[c0 * taps + e];
        gp[c0 * taps + e] = G;
        dot += G * v; nn += v * v;
      }
    }
  } else
  for (int q = threadIdx.x; q < fan; q += 256) {
    const int tap = q / cin, ci = q - tap * cin;
    const size_t pi = ((size_t)cop * taps + tap) * d->CinP + ci;
    float G0 = 0.f, G1 = 0.f, G2 = 0.f, G3 = 0.f;            // 4 independent load chains (HBM latency)
    int s_ = 0;
    for (; s_ + 3 < nsp; s_ += 4) {
      G0 += bf2f(dwp[(size_t)s_ * slab + pi]);       G1 += bf2f(dwp[(size_t)(s_ + 1) * slab + pi]);
      G2 += bf2f(dwp[(size_t)(s_ + 2) * slab + pi]); G3 += bf2f(dwp[(size_t)(s_ + 3) * slab + pi]);
    }
    for (; s_ < nsp; ++s_) G0 += bf2f(dwp[(size_t)s_ * slab + pi]);
    const float G = (G0 + G1) + (G2 + G3);
    dws[pi] = G;
    const float v = w[ci * taps + tap];
    dot += G * v; nn += v * v;
  }
  dot = block_sum(dot, red);          // (block_sum's barriers also order the dws writes before pass 2)
  nn = block_sum(nn, red);
  const float n = sqrtf(nn);
  const float s = W_EPS + n * rs;
  const float c = d->gain * rs;
  const float k1 = c / s;
  const float k2 = (n > 0.f) ? c * dot * rs / (s * s * n) : 0.f;
  if (chunked) {
    for (int e = threadIdx.x; e < fan; e += 256) g[e] += k1 * gp[e] - k2 * w[e];
    return;
  }
  for (int q = threadIdx.x; q < fan; q += 256) {
    const int tap = q / cin, ci = q - tap * cin;
    const size_t pi = ((size_t)cop * taps + tap) * d->CinP + ci;
    const int e = ci * taps + tap;
    g[e] += k1 * dws[pi] - k2 * w[e];
  }
}

extern "C" int oniris_weight_prep(const OnirisWeightDesc* descs_dev, int ndesc, int total_rows, int total_tiles,
                                  int training, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(descs_dev && ndesc > 0 && total_rows > 0 && total_tiles > 0, "weight_prep: bad arguments");
  ONIRIS_KLAUNCH(weight_prep_kernel, dim3(total_rows), dim3(256), 0, stream, descs_dev, ndesc, training);
  ONIRIS_LAUNCH_CHECK();
  ONIRIS_KLAUNCH(weight_wb_kernel, dim3(total_tiles, 18), dim3(256), 0, stream, descs_dev, ndesc);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_weight_bwd(const OnirisWeightDesc* descs_dev, int ndesc, int total_rows, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(descs_dev && ndesc > 0 && total_rows > 0, "weight_bwd: bad arguments");
  static int chunk_max_nsp = -1;          // (ONIRIS_WBWD_CHUNK_NSP: A/B knob of the transposing form's slab-count limit)
  if (chunk_max_nsp < 0) { const char* e = getenv("ONIRIS_WBWD_CHUNK_NSP"); chunk_max_nsp = e ? atoi(e) : 1 << 30; }
  ONIRIS_KLAUNCH(weight_bwd_kernel, dim3(total_rows), dim3(256), 0, stream, descs_dev, ndesc, chunk_max_nsp);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Optimizer step of gym_train.py:105-108 / cs_train.py:117-121 on the flat fp32 buffers:
//   clip_grad_norm_(params, max_norm)  ->  AdamW  ->  PowerFunctionEMA.update (edm2/phema.py:101-106, two profiles)
// as ONE elementwise pass (16 B/lane): the clip coefficient comes from a device scalar (the squared gradient norm,
// oniris_sqnorm) so nothing synchronises with the host, and each EMA copy is lerp'ed towards the freshly updated
// parameter while it is still in registers.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2, float gscale, const float* __restrict__ gnorm_sq, float max_norm,
                             float* __restrict__ ema0, float w0, float* __restrict__ ema1, float w1) {
  if (bc1 == 0.f) {                             // step 0 = "these parameters had no gradient": only the EMA copies follow
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride)
      for (size_t k = i; k < n && k < i + 4; ++k) {
        if (ema0) ema0[k] += w0 * (p[k] - ema0[k]);
        if (ema1) ema1[k] += w1 * (p[k] - ema1[k]);
      }
    return;
  }
  if (gnorm_sq) {                               // torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))
    const float coef = max_norm / (sqrtf(*gnorm_sq) * fabsf(gscale) + 1e-6f);
    if (coef < 1.f) gscale *= coef;
  }
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 P = *(float4*)(p + i), G = *(const float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
    float* pp = (float*)&P; float* gg = (float*)&G; float* mm = (float*)&M; float* vv = (float*)&V;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gg[k] * gscale;
      mm[k] = b1 * mm[k] + (1.f - b1) * gr;
      vv[k] = b2 * vv[k] + (1.f - b2) * gr * gr;
      const float mh = mm[k] / bc1, vh = vv[k] / bc2;
      pp[k] = pp[k] * (1.f - lr * wd) - lr * mh / (sqrtf(vh) + eps);
    }
    *(float4*)(p + i) = P; *(float4*)(m + i) = M; *(float4*)(v + i) = V;
    if (ema0) {
      float4 E = *(float4*)(ema0 + i);
      E.x += w0 * (P.x - E.x); E.y += w0 * (P.y - E.y); E.z += w0 * (P.z - E.z); E.w += w0 * (P.w - E.w);
      *(float4*)(ema0 + i) = E;
    }
    if (ema1) {
      float4 E = *(float4*)(ema1 + i);
      E.x += w1 * (P.x - E.x); E.y += w1 * (P.y - E.y); E.z += w1 * (P.z - E.z); E.w += w1 * (P.w - E.w);
      *(float4*)(ema1 + i) = E;
    }
  }
  if (i < n && i + 3 >= n) {
    for (size_t k = i; k < n; ++k) {
      const float gr = g[k] * gscale;
      m[k] = b1 * m[k] + (1.f - b1) * gr;
      v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
      p[k] = p[k] * (1.f - lr * wd) - lr * (m[k] / bc1) / (sqrtf(v[k] / bc2) + eps);
      if (ema0) ema0[k] += w0 * (p[k] - ema0[k]);
      if (ema1) ema1[k] += w1 * (p[k] - ema1[k]);
    }
  }
}

static int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, float grad_scale, const float* gnorm_sq, float max_norm,
                        float* ema0, float w0, float* ema1, float w1, hipStream_t stream) {
  ONIRIS_CHECK_ARG(p && g && m && v && step >= 0, "adamw: bad arguments");
  ONIRIS_CHECK_ARG(!gnorm_sq || max_norm > 0.f, "adamw: clipping needs max_norm > 0");
  if (n == 0) return ONIRIS_OK;
  const float bc1 = step ? 1.f - powf(beta1, (float)step) : 0.f, bc2 = step ? 1.f - powf(beta2, (float)step) : 0.f;
  if (step == 0 && !ema0 && !ema1) return ONIRIS_OK;
  size_t nb = (n / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb == 0) nb = 1;
  ONIRIS_KLAUNCH(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, grad_scale, gnorm_sq, max_norm, ema0, w0, ema1, w1);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, float grad_scale, oniris_stream_t stream_) {
  return launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, 0.f, nullptr, 0.f,
                      nullptr, 0.f, (hipStream_t)stream_);
}

extern "C" int oniris_adamw_clip_ema(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, int step, float grad_scale,
                                     const float* gnorm_sq, float max_norm, float* ema0, float ema_w0, float* ema1,
                                     float ema_w1, oniris_stream_t stream_) {
  return launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, gnorm_sq, max_norm, ema0,
                      ema_w0, ema1, ema_w1, (hipStream_t)stream_);
}

// sum of squares of a flat fp32 buffer -> out[0] (device).  Two stages through out[1 .. 1+ONIRIS_SQNORM_WS) so the
// result is deterministic (no float atomics): <= 1024 block partials, then one block adds them in a fixed order.
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
  __shared__ float red[16];
  float acc = 0.f;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    const float4 G = *(const float4*)(g + i);
    acc += G.x * G.x + G.y * G.y + G.z * G.z + G.w * G.w;
  }
  if (i < n && i + 3 >= n)
    for (size_t k = i; k < n; ++k) acc += g[k] * g[k];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += part[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = acc;
}

extern "C" int oniris_sqnorm(const float* g, size_t n, float* out, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(g && out, "sqnorm: null pointer");
  size_t nb = (n / 4 + 255) / 256;
  if (nb > ONIRIS_SQNORM_WS) nb = ONIRIS_SQNORM_WS;
  if (nb == 0) nb = 1;
  ONIRIS_KLAUNCH(sqnorm_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, g, n, out + 1);
  ONIRIS_LAUNCH_CHECK();
  ONIRIS_KLAUNCH(sqnorm_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)(out + 1), (int)nb, out);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward pre-pass of the gated conv (one pass over dout): per frame-slot n
//   S1[n] = sum dout*out, S2[n] = sum dout*y3,  d ca[n] = (S1 - cb*S2)/ca  (= sum dout*y2),  d cb[n] = S2,
//   dy3[b,t] = cb[b,0,t]*dout[b,0,t] + cb[b,1,t]*dout[b,1,t]
// dout/out [B][S][T][PC], y3/dy3 [B][T][PC].  grid = B*T frames, HBM-bound, 16 B/lane loads.
__global__ __launch_bounds__(256) void gconv_bwd_prep_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                             const bf16* __restrict__ y3, const float* __restrict__ ca,
                                                             const float* __restrict__ cb,
                                                             float* __restrict__ S1, float* __restrict__ S2,
                                                             bf16* __restrict__ dy3, int S, int T, size_t PC) {
  __shared__ float red[16];
  const int bt = blockIdx.x, b = bt / T, t = bt % T;
  const bf16* y3f = y3 + (size_t)bt * PC;
  bf16* dy3f = dy3 + (size_t)bt * PC;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  float c[2];
  const bf16* df[2];
  const bf16* of[2];
  for (int s = 0; s < S; ++s) {
    const size_t n = (size_t)(b * S + s) * T + t;
    c[s] = cb[n]; df[s] = dout + n * PC; of[s] = out + n * PC;
  }
  // blockIdx.y = slice of the frame (B*T blocks alone leave half of the chip idle, each streaming 1.5 MB by itself: 1.6 TB/s)
  const size_t per = ((PC / 8 + gridDim.y - 1) / gridDim.y) * 8, e_lo = blockIdx.y * per, e_hi = (e_lo + per < PC) ? e_lo + per : PC;
  for (size_t e = e_lo + (size_t)threadIdx.x * 8; e < e_hi; e += 256 * 8) {
    const bf16x8 yv = *(const bf16x8*)(y3f + e);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int s = 0; s < S; ++s) {
      const bf16x8 dv = *(const bf16x8*)(df[s] + e);
      const bf16x8 ov = *(const bf16x8*)(of[s] + e);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float dd = bf2f(dv[i]);
        s1[s] += dd * bf2f(ov[i]); s2[s] += dd * bf2f(yv[i]); acc[i] += c[s] * dd;
      }
    }
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(acc[i]);
    *(bf16x8*)(dy3f + e) = o;
  }
  for (int s = 0; s < S; ++s) {
    const float a = block_sum(s1[s], red), bsum = block_sum(s2[s], red);
    if (threadIdx.x == 0) { const size_t n = (size_t)(b * S + s) * T + t; atomicAdd(S1 + n, (a - cb[n] * bsum) / ca[n]); atomicAdd(S2 + n, bsum); }
  }
}

// Same pre-pass fused with the adjoint of the conv epilogue, so the incoming gradient is read ONCE:
//   mode 1 (EPI_EMB_SILU): g = d u, u = silu(y*c)/0.596  ->  dout = g*silu'(y*c)/0.596*c ; dc[n][co] += sum_pixels(...)*y
//   mode 2 (EPI_MPSUM):    g = d out, out = clip(ta*res + tb*v) -> dres = ta*g*mask ; dout = tb*g*mask
// then S1/S2/dy3 exactly as gconv_bwd_prep_kernel.  raw = y (mode 1) or v (mode 2).
template <int MODE, bool NT>
__global__ __launch_bounds__(1024) void gconv_bwd_fused_kernel(const bf16* g, const bf16* __restrict__ raw,
                                                              const bf16* __restrict__ y3, const float* __restrict__ ca,
                                                              const float* __restrict__ cb, const float* __restrict__ cs,
                                                              const bf16* __restrict__ xo, bf16* dout,
                                                              bf16* __restrict__ dres, bf16* __restrict__ dy3,
                                                              float* __restrict__ dca, float* __restrict__ dcb,
                                                              float* __restrict__ dcs, int T, int P, int C, float ta,
                                                              float tb, float clip, int pix_per_block, int cs_pitch,
                                                              const int* __restrict__ clip_flag, float* __restrict__ ca_scaled) {
  // mode 2, aliasing protocol (include/oniris.h): dgrad / wgrad read g itself with tb folded into the own-frame coefficient;
  // only a forward that really clipped something makes this pass read xo and write the masked gradient (in place)
  const bool alias = MODE == 2 && ca_scaled != nullptr;
  const bool masked = MODE == 2 && clip > 0.f && (!alias || *clip_flag != 0);
  __shared__ float red[16];
  __shared__ float accs[2][512];
  const int bt = blockIdx.x, b = bt / T, t = bt % T;
  const size_t PC = (size_t)P * C;
  const int G8 = C >> 3;
  const int NTH = blockDim.x;                      // 256 or 1024 threads (the host picks: see oniris_gconv_bwd_fused)
  const int cg = threadIdx.x % G8, pl = threadIdx.x / G8, npl = NTH / G8;
  if (MODE == 1)
    for (int i = threadIdx.x; i < 2 * 512; i += NTH) (&accs[0][0])[i] = 0.f;
  __syncthreads();
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  float part[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) part[s][i] = 0.f;
  size_t nn[2];
  float cbv[2], cv[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    nn[s] = (size_t)(b * 2 + s) * T + t;
    cbv[s] = cb[nn[s]];
#pragma unroll
    for (int i = 0; i < 8; ++i) cv[s][i] = 1.f;
    if (MODE == 1 && pl < npl) {
      const float4 c0 = *(const float4*)(cs + nn[s] * cs_pitch + cg * 8), c1 = *(const float4*)(cs + nn[s] * cs_pitch + cg * 8 + 4);
      cv[s][0] = c0.x; cv[s][1] = c0.y; cv[s][2] = c0.z; cv[s][3] = c0.w;
      cv[s][4] = c1.x; cv[s][5] = c1.y; cv[s][6] = c1.z; cv[s][7] = c1.w;
    }
  }
  if (pl < npl) {
    const int p0 = blockIdx.y * pix_per_block, p1 = min(P, p0 + pix_per_block);
    for (int p = p0 + pl; p < p1; p += npl) {
      const size_t off = (size_t)p * C + cg * 8;
      const bf16x8 yv3 = ldv<NT>((const bf16x8*)(y3 + (size_t)bt * PC + off));
      float acc3[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc3[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const size_t o = nn[s] * PC + off;
        const bf16x8 gv = ldv<NT>((const bf16x8*)(g + o));
        const bf16x8 rv = ldv<NT>((const bf16x8*)(raw + o));
        bf16x8 dv, drv;
        bf16x8 xv;
        if (masked) xv = ldv<NT>((const bf16x8*)(xo + o));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float r_ = bf2f(rv[i]);
          float d;
          if (MODE == 1) {
            const float z = r_ * cv[s][i];
            const float sg = sigmoid_fast(z);
            const float dz = bf2f(gv[i]) * (sg * (1.f + z * (1.f - sg))) * (1.0f / 0.596f);
            part[s][i] += dz * r_;
            d = dz * cv[s][i];
          } else {
            float gg = bf2f(gv[i]);
            if (masked && !(fabsf(bf2f(xv[i])) < clip)) gg = 0.f;
            drv[i] = f2bf(gg * ta);
            d = gg * tb;
            if (alias) dv[i] = f2bf(gg);                // (what dgrad / wgrad read: the gradient itself, masked if need be)
          }
          float dr = d;                                 // aliasing: tb * (bf16 gradient), exact in fp32
          if (!alias) { dv[i] = f2bf(d); dr = bf2f(dv[i]); }      // else: the rounded value is what dgrad / wgrad will consume
          s1[s] += dr * r_; s2[s] += dr * bf2f(yv3[i]); acc3[i] += cbv[s] * dr;
        }
        if (!alias || masked) stv<NT>((bf16x8*)(dout + o), dv);   // (aliasing + masked: dout IS g -- every element is read
        if (MODE == 2 && dres) stv<NT>((bf16x8*)(dres + o), drv); //  and rewritten by the same lane)
      }
      bf16x8 o3;
#pragma unroll
      for (int i = 0; i < 8; ++i) o3[i] = f2bf(acc3[i]);
      stv<NT>((bf16x8*)(dy3 + (size_t)bt * PC + off), o3);
    }
    if (MODE == 1 && (G8 & (G8 - 1)) != 0) {                      // (channel groups not a power of two: every lane adds its own)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&accs[s][cg * 8 + i], part[s][i]);
    }
  }
  if (MODE == 1 && (G8 & (G8 - 1)) == 0) {
    // the lanes of a wave that hold the same channel group (cg = lane % G8: G8 a power of two <= 64, every thread of the block
    // is a pixel lane then) add up by shuffles first; G8 lanes per wave reach the LDS accumulators instead of 64
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = part[s][i];
        for (int o = G8; o < 64; o <<= 1) v += __shfl_xor(v, o);
        if ((int)(threadIdx.x & 63) < G8) atomicAdd(&accs[s][cg * 8 + i], v);
      }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float a_ = block_sum(s1[s], red), b_ = block_sum(s2[s], red);
    if (threadIdx.x == 0) {
      atomicAdd(dca + nn[s], (a_ - cbv[s] * b_) / ca[nn[s]]); atomicAdd(dcb + nn[s], b_);
      if (alias && blockIdx.y == 0) ca_scaled[nn[s]] = tb * ca[nn[s]];
    }
  }
  if (MODE == 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += NTH) {
      atomicAdd(dcs + nn[0] * C + i, accs[0][i]); atomicAdd(dcs + nn[1] * C + i, accs[1][i]);
    }
  }
}

extern "C" int oniris_gconv_bwd_fused(int mode, const void* g, const void* raw, const void* y3, const float* coef_own,
                                      const float* coef_ctx, const float* cscale, const void* xo, void* dout, void* dres,
                                      void* dy3, float* d_coef_own, float* d_coef_ctx, float* d_cscale, int B, int T,
                                      int P, int C, float ta, float tb, float clip, int cscale_pitch,
                                      const int32_t* clip_flag, float* coef_own_scaled, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG((mode == 1 || mode == 2) && g && raw && y3 && coef_own && coef_ctx && dout && dy3 && d_coef_own &&
                   d_coef_ctx && B > 0 && T > 0 && P > 0 && C % 8 == 0 && C <= 512, "gconv_bwd_fused: bad arguments");
  ONIRIS_CHECK_ARG(mode != 1 || (cscale && d_cscale), "gconv_bwd_fused: mode 1 needs cscale / d_cscale");
  ONIRIS_CHECK_ARG(mode != 2 || ((dres || coef_own_scaled) && (clip <= 0.f || xo)),
                   "gconv_bwd_fused: mode 2 needs dres (optional under the aliasing protocol) and xo when clipping");
  ONIRIS_CHECK_ARG((mode == 2 || (!clip_flag && !coef_own_scaled)) && (!clip_flag || coef_own_scaled) &&
                   (!coef_own_scaled || clip <= 0.f || clip_flag),
                   "gconv_bwd_fused: coef_own_scaled selects the aliasing protocol (mode 2); with clip > 0 it needs clip_flag");
  ONIRIS_CHECK_ARG(!coef_own_scaled || dout == g, "gconv_bwd_fused: the aliasing protocol masks g in place (pass dout = g)");
  // pixel slices so that the launch covers the chip (B*T alone is ~128 blocks); partial sums meet through atomics,
  // so d_coef_own / d_coef_ctx / d_cscale must be ZERO on entry.
  int slices = 1;
  // mode 1 ends every block with 2*C global atomics (the emb-scale gradient; all slices of a frame add into the same row):
  // measured 64 / 54 / 43 / 37 / 42 us at 4096 / 2048 / 1024 / 512 / 256 blocks of 256 threads (64x64x32ch level) -- few
  // blocks, so they are 1024 threads wide (two per CU fill it); mode 2 is flat from 1024 blocks of 256 up
  static int nth1 = -1;                              // (ONIRIS_GCONV_BWD_THREADS: A/B knob for mode 1)
  if (nth1 < 0) { const char* e = getenv("ONIRIS_GCONV_BWD_THREADS"); nth1 = e ? atoi(e) : 256; }
  const int nth = (mode == 1 && (long long)P * (C / 8) >= 4096) ? nth1 : 256;
  const int npl = nth / (C / 8) > 0 ? nth / (C / 8) : 1;
  static int tgt1 = -1;                              // (ONIRIS_GCONV_BWD_BLOCKS: A/B knob for mode 1)
  if (tgt1 < 0) { const char* e = getenv("ONIRIS_GCONV_BWD_BLOCKS"); tgt1 = e ? atoi(e) : 512; }
  const int tgt = (mode == 1) ? tgt1 : 2048;
  while (slices < 32 && P / (slices * 2) >= npl * 2 && (long long)B * T * slices < tgt) slices *= 2;
  const int ppb = cdiv(P, slices);
  const int csp = cscale_pitch > 0 ? cscale_pitch : C;
  ONIRIS_CHECK_ARG(csp >= C && csp % 4 == 0, "gconv_bwd_fused: cscale_pitch must be a multiple of 4 and >= C");
  const bool nt = 2LL * B * T * P * C * 2 >= oniris_ew_nt_bytes();       // (the gradient tensor: both slots)
#define GCONV_BWD_LAUNCH(MODE_, NT_, NTH_, FLAG_, CAS_) ONIRIS_KLAUNCH((gconv_bwd_fused_kernel<MODE_, NT_>), dim3(B * T, slices), dim3(NTH_), 0, stream, (const bf16*)g, \
                       (const bf16*)raw, (const bf16*)y3, coef_own, coef_ctx, cscale, (const bf16*)xo, (bf16*)dout, \
                       (bf16*)dres, (bf16*)dy3, d_coef_own, d_coef_ctx, d_cscale, T, P, C, ta, tb, clip, ppb, csp, FLAG_, CAS_)
  if (mode == 1) { if (nt) GCONV_BWD_LAUNCH(1, true, nth, nullptr, nullptr); else GCONV_BWD_LAUNCH(1, false, nth, nullptr, nullptr); }
  else { if (nt) GCONV_BWD_LAUNCH(2, true, 256, (const int*)clip_flag, coef_own_scaled); else GCONV_BWD_LAUNCH(2, false, 256, (const int*)clip_flag, coef_own_scaled); }
#undef GCONV_BWD_LAUNCH
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_gconv_bwd_prep(const void* dout, const void* out, const void* y3, const float* coef_own,
                                     const float* coef_ctx, float* S1, float* S2, void* dy3, int B, int S, int T,
                                     int64_t frame_elems,
                                     oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(dout && out && y3 && coef_own && coef_ctx && S1 && S2 && dy3 && B > 0 && T > 0 && (S == 1 || S == 2) &&
                   frame_elems > 0 && frame_elems % 8 == 0, "gconv_bwd_prep: bad arguments");
  int slices = 1;                                    // (d_coef_own / d_coef_ctx are ACCUMULATED: zero on entry)
  while (slices < 16 && (long long)B * T * slices < 1024 && frame_elems / (slices * 2) >= 256 * 8 * 2) slices *= 2;
  ONIRIS_KLAUNCH(gconv_bwd_prep_kernel, dim3(B * T, slices), dim3(256), 0, stream, (const bf16*)dout, (const bf16*)out,
                     (const bf16*)y3, coef_own, coef_ctx, S1, S2, (bf16*)dy3, S, T, (size_t)frame_elems);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
