// fp32 verification path (Precond(use_fp16=False) / forward(force_fp32=True), reference networks_edm2.py:285,294: the switch
// that picks the arithmetic type of the whole net).  Speed is not the point of this file; exact fp32 arithmetic behind the
// same module API is: activations, weights and every product / sum in fp32, the contractions on the matrix cores' exact-f32
// instruction (v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulation), so that the reference's own criterion
// std(diff) <= 3e-4 (edm2/consistency_test.py:23-32) can be held against the fp32 fixtures.
//
//   oniris_conv_f32      implicit-GEMM convolution (1x1 / 3x3, zero spatial padding), channels-last fp32, any channel counts;
//                        the data gradient is the same kernel on the flipped / transposed weights (host side)
//   oniris_wgrad_f32     its weight gradient: GEMM over the positions, split over position chunks, fp32 atomics into a
//                        zeroed dW (the summation order varies at the 1e-7 level: a verification path, not the bit-stable
//                        product path)
//   oniris_attn_f32_fwd / _bwd   softmax attention for ANY head width up to 256 channels with the masks of the path
//                        (dense, frame-causal with a query offset = prefill / cached steps, the DART training mask
//                        table AND mask_mod in closed form) -- also what serves heads wider than
//                        the 64 channels the product kernels are written for (networks_edm2.py:28,39 accepts any width)
#include "common.h"
#include "../../include/oniris.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ------------------------------------------------------------------------------------------------------------------
// conv: D[co][pos] = sum_{tap, ci} W[tap][co][ci] * X[shift(pos, tap)][ci]
// workgroup = 4 waves = 64 co x 64 positions (each wave a 32 x 32 quarter); K in chunks of 16 channels through LDS
struct ConvF32Dev {
  const float* x; const float* w; float* out;
  long long npos; int H, W, Cin, Cout, taps;
};

__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvF32Dev d) {
  __shared__ float As[16][65], Bs[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long pos0 = (long long)blockIdx.x * 64;
  const int co0 = blockIdx.y * 64;
  const int cb = (wave & 1) * 32, pb = (wave >> 1) * 32;
  f32x16_t acc = {0};
  const int HW = d.H * d.W;
  for (int tap = 0; tap < d.taps; ++tap) {
    const int dy = d.taps == 9 ? tap / 3 - 1 : 0, dx = d.taps == 9 ? tap % 3 - 1 : 0;
    for (int ci0 = 0; ci0 < d.Cin; ci0 += 16) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, k = e & 15, r = e >> 4;
        const int ci = ci0 + k, co = co0 + r;
        As[k][r] = (ci < d.Cin && co < d.Cout) ? d.w[((size_t)tap * d.Cout + co) * d.Cin + ci] : 0.f;
        const long long p = pos0 + r;
        float v = 0.f;
        if (ci < d.Cin && p < d.npos) {
          const long long n = p / HW;
          const int rem = (int)(p - n * HW), y = rem / d.W + dy, xx = rem % d.W + dx;
          if (y >= 0 && y < d.H && xx >= 0 && xx < d.W) v = d.x[((n * d.H + y) * d.W + xx) * d.Cin + ci];
        }
        Bs[k][r] = v;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        const float a = As[kk + (lane >> 5)][cb + (lane & 31)];
        const float b = Bs[kk + (lane >> 5)][pb + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
  }
  const long long p = pos0 + pb + (lane & 31);
  if (p < d.npos) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cb + mfma_row(r, lane);
      if (co < d.Cout) d.out[p * d.Cout + co] = acc[r];
    }
  }
}

extern "C" int oniris_conv_f32(const float* x, const float* w, float* out, int64_t N, int H, int W, int Cin, int Cout, int taps,
                               oniris_stream_t stream_) {
  ONIRIS_CHECK_ARG(x && w && out && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv_f32: bad arguments");
  ONIRIS_CHECK_ARG(taps == 1 || taps == 9, "conv_f32: taps must be 1 or 9");
  ConvF32Dev d{x, w, out, (long long)N * H * W, H, W, Cin, Cout, taps};
  const long long nb = (d.npos + 63) / 64;
  ONIRIS_CHECK_ARG(nb < (1LL << 31), "conv_f32: too many positions");
  ONIRIS_KLAUNCH(conv_f32_kernel, dim3((unsigned)nb, (unsigned)cdiv(Cout, 64)), dim3(256), 0, (hipStream_t)stream_, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: dW[tap][co][ci] += sum_pos dY[pos][co] * X[shift(pos, tap)][ci]    (dW zeroed by the caller)
struct WgradF32Dev {
  const float* x; const float* dy; float* dw;
  long long npos; int H, W, Cin, Cout, taps, chunk;
};

__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradF32Dev d) {
  __shared__ float As[16][65], Bs[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nci = (d.Cin + 63) / 64;
  const int co0 = (blockIdx.x / nci) * 64, ci0 = (blockIdx.x % nci) * 64, tap = blockIdx.y;
  const long long p_lo = (long long)blockIdx.z * d.chunk, p_hi = p_lo + d.chunk < d.npos ? p_lo + d.chunk : d.npos;
  const int dy = d.taps == 9 ? tap / 3 - 1 : 0, dx = d.taps == 9 ? tap % 3 - 1 : 0;
  const int cb = (wave & 1) * 32, ib = (wave >> 1) * 32;
  const int HW = d.H * d.W;
  f32x16_t acc = {0};
  for (long long p0 = p_lo; p0 < p_hi; p0 += 16) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, r = e & 63, k = e >> 6;       // consecutive threads: consecutive channels of one position
      const long long p = p0 + k;
      const int co = co0 + r, ci = ci0 + r;
      float a = 0.f, b = 0.f;
      if (p < p_hi) {
        if (co < d.Cout) a = d.dy[p * d.Cout + co];
        if (ci < d.Cin) {
          const long long n = p / HW;
          const int rem = (int)(p - n * HW), y = rem / d.W + dy, xx = rem % d.W + dx;
          if (y >= 0 && y < d.H && xx >= 0 && xx < d.W) b = d.x[((n * d.H + y) * d.W + xx) * d.Cin + ci];
        }
      }
      As[k][r] = a;
      Bs[k][r] = b;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      const float a = As[kk + (lane >> 5)][cb + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][ib + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  const int ci = ci0 + ib + (lane & 31);
  if (ci < d.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cb + mfma_row(r, lane);
      if (co < d.Cout) atomicAdd(&d.dw[((size_t)tap * d.Cout + co) * d.Cin + ci], acc[r]);
    }
  }
}

extern "C" int oniris_wgrad_f32(const float* x, const float* dy, float* dw, int64_t N, int H, int W, int Cin, int Cout, int taps,
                                oniris_stream_t stream_) {
  ONIRIS_CHECK_ARG(x && dy && dw && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "wgrad_f32: bad arguments");
  ONIRIS_CHECK_ARG(taps == 1 || taps == 9, "wgrad_f32: taps must be 1 or 9");
  WgradF32Dev d{x, dy, dw, (long long)N * H * W, H, W, Cin, Cout, taps, 4096};
  const long long nz = (d.npos + d.chunk - 1) / d.chunk;
  ONIRIS_CHECK_ARG(nz <= 65535, "wgrad_f32: too many positions (65535 chunks of 4096)");
  ONIRIS_KLAUNCH(wgrad_f32_kernel, dim3((unsigned)(cdiv(Cout, 64) * cdiv(Cin, 64)), (unsigned)taps, (unsigned)nz), dim3(256), 0,
                 (hipStream_t)stream_, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// attention.  q [BH][Lq][D], k / v [BH][Lk][D] fp32 contiguous; one WAVE per query row (forward, dQ) or per key row
// (dK / dV), 8 rows per workgroup; the other side streams through LDS in chunks of 64 rows.
// mask_mode 0: every key.  1: frame-causal, key frame <= query frame + q_frame_off (frames of P tokens: prefill,
// q_frame_off = cached frames for steps against a cache).  2: the DART training mask over 2T frames (clean | noised),
// table AND mask_mod (attention_masking.py:27-53 as compiled FlexAttention evaluates it):
//   clean q: clean keys of frames <= qf;  noised q: clean keys of frames < fpb * floor(qf / fpb), fpb = max(1, 128 / P), and
//   its own noised frame.
struct AttnF32Dev {
  const float* q; const float* k; const float* v; float* out; float* lse;
  const float* dout; const float* delta; float* dq; float* dk; float* dv;
  int Lq, Lk, D, mask_mode, P, T, q_frame_off;
  float scale;
};

__device__ __forceinline__ bool attn_allowed(const AttnF32Dev& d, int qi, int kj) {
  if (d.mask_mode == 0) return true;
  const int qf = qi / d.P, kf = kj / d.P;
  if (d.mask_mode == 1) return kf <= qf + d.q_frame_off;
  const int qs = qf / d.T, ks = kf / d.T, qt = qf % d.T, kt = kf % d.T;
  const int fpb = d.P >= 128 ? 1 : 128 / d.P;
  if (qs == 0) return ks == 0 && kt <= qt;
  if (ks == 0) return kt < fpb * (qt / fpb);
  return kt == qt;
}

#define AF_ROWS 8          // waves per workgroup
#define AF_MAXU 4          // channels per lane: D <= 256

// LDS: [64][D + 1] x 2 (the streamed side) + AF_ROWS x D (this wave's own row) [+ AF_ROWS x D for the second own row]
__global__ __launch_bounds__(64 * AF_ROWS) void attn_f32_fwd_kernel(const AttnF32Dev d) {
  extern __shared__ float smem[];
  const int D = d.D, DP = D + 1;
  float* Ks = smem;
  float* Vs = Ks + 64 * DP;
  float* Qs = Vs + 64 * DP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, qi = blockIdx.x * AF_ROWS + wave;
  const bool live = qi < d.Lq;
  const float* kbase = d.k + (size_t)bh * d.Lk * D;
  const float* vbase = d.v + (size_t)bh * d.Lk * D;
  float* qrow = Qs + wave * D;
  if (live)
    for (int c = lane; c < D; c += 64) qrow[c] = d.q[((size_t)bh * d.Lq + qi) * D + c];
  float m = -1e30f, l = 0.f, acc[AF_MAXU] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < d.Lk; k0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * D; e += 64 * AF_ROWS) {
      const int r = e / D, c = e - r * D;
      const bool in = k0 + r < d.Lk;
      Ks[r * DP + c] = in ? kbase[(size_t)(k0 + r) * D + c] : 0.f;
      Vs[r * DP + c] = in ? vbase[(size_t)(k0 + r) * D + c] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const int kj = k0 + lane;
    const bool ok = kj < d.Lk && attn_allowed(d, qi, kj);
    float s = 0.f;
    for (int c = 0; c < D; ++c) s += qrow[c] * Ks[lane * DP + c];
    s *= d.scale;
    const float mn = fmaxf(m, wave_max(ok ? s : -1e30f));
    const float p = ok ? __expf(s - mn) : 0.f;
    const float corr = __expf(m - mn);
    l = l * corr + wave_sum(p);
    m = mn;
#pragma unroll
    for (int u = 0; u < AF_MAXU; ++u) acc[u] *= corr;
    for (int j = 0; j < 64; ++j) {
      const float pj = __shfl(p, j);
      if (pj != 0.f) {
#pragma unroll
        for (int u = 0; u < AF_MAXU; ++u)
          if (lane + 64 * u < D) acc[u] += pj * Vs[j * DP + lane + 64 * u];
      }
    }
  }
  if (live) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int u = 0; u < AF_MAXU; ++u)
      if (lane + 64 * u < D) d.out[((size_t)bh * d.Lq + qi) * D + lane + 64 * u] = acc[u] * inv;
    if (lane == 0) d.lse[(size_t)bh * d.Lq + qi] = l > 0.f ? m + __logf(l) : -1e30f;
  }
}

// dQ[qi] = scale * sum_j P_ij (dP_ij - delta_i) K_j,   P = exp(scale q.k - lse),  dP_ij = dO_i . V_j
__global__ __launch_bounds__(64 * AF_ROWS) void attn_f32_dq_kernel(const AttnF32Dev d) {
  extern __shared__ float smem[];
  const int D = d.D, DP = D + 1;
  float* Ks = smem;
  float* Vs = Ks + 64 * DP;
  float* Qs = Vs + 64 * DP;
  float* Os = Qs + AF_ROWS * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, qi = blockIdx.x * AF_ROWS + wave;
  const bool live = qi < d.Lq;
  const float* kbase = d.k + (size_t)bh * d.Lk * D;
  const float* vbase = d.v + (size_t)bh * d.Lk * D;
  float* qrow = Qs + wave * D;
  float* orow = Os + wave * D;
  float lse = 0.f, delta = 0.f;
  if (live) {
    for (int c = lane; c < D; c += 64) {
      qrow[c] = d.q[((size_t)bh * d.Lq + qi) * D + c];
      orow[c] = d.dout[((size_t)bh * d.Lq + qi) * D + c];
    }
    lse = d.lse[(size_t)bh * d.Lq + qi];
    delta = d.delta[(size_t)bh * d.Lq + qi];
  }
  float acc[AF_MAXU] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < d.Lk; k0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * D; e += 64 * AF_ROWS) {
      const int r = e / D, c = e - r * D;
      const bool in = k0 + r < d.Lk;
      Ks[r * DP + c] = in ? kbase[(size_t)(k0 + r) * D + c] : 0.f;
      Vs[r * DP + c] = in ? vbase[(size_t)(k0 + r) * D + c] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const int kj = k0 + lane;
    const bool ok = kj < d.Lk && attn_allowed(d, qi, kj);
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < D; ++c) {
      s += qrow[c] * Ks[lane * DP + c];
      dp += orow[c] * Vs[lane * DP + c];
    }
    const float p = ok ? __expf(s * d.scale - lse) : 0.f;
    const float ds = p * (dp - delta) * d.scale;
    for (int j = 0; j < 64; ++j) {
      const float dj = __shfl(ds, j);
      if (dj != 0.f) {
#pragma unroll
        for (int u = 0; u < AF_MAXU; ++u)
          if (lane + 64 * u < D) acc[u] += dj * Ks[j * DP + lane + 64 * u];
      }
    }
  }
  if (live) {
#pragma unroll
    for (int u = 0; u < AF_MAXU; ++u)
      if (lane + 64 * u < D) d.dq[((size_t)bh * d.Lq + qi) * D + lane + 64 * u] = acc[u];
  }
}

// dV[kj] = sum_i P_ij dO_i;  dK[kj] = scale * sum_i P_ij (dP_ij - delta_i) Q_i;  queries stream through LDS
__global__ __launch_bounds__(64 * AF_ROWS) void attn_f32_dkv_kernel(const AttnF32Dev d) {
  extern __shared__ float smem[];
  const int D = d.D, DP = D + 1;
  float* Qs = smem;
  float* Os = Qs + 64 * DP;
  float* Ks = Os + 64 * DP;
  float* Vs = Ks + AF_ROWS * D;
  float* Ls = Vs + AF_ROWS * D;          // [64] lse | [64] delta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, kj = blockIdx.x * AF_ROWS + wave;
  const bool live = kj < d.Lk;
  const float* qbase = d.q + (size_t)bh * d.Lq * D;
  const float* obase = d.dout + (size_t)bh * d.Lq * D;
  float* krow = Ks + wave * D;
  float* vrow = Vs + wave * D;
  if (live)
    for (int c = lane; c < D; c += 64) {
      krow[c] = d.k[((size_t)bh * d.Lk + kj) * D + c];
      vrow[c] = d.v[((size_t)bh * d.Lk + kj) * D + c];
    }
  float ak[AF_MAXU] = {0.f, 0.f, 0.f, 0.f}, av[AF_MAXU] = {0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < d.Lq; q0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * D; e += 64 * AF_ROWS) {
      const int r = e / D, c = e - r * D;
      const bool in = q0 + r < d.Lq;
      Qs[r * DP + c] = in ? qbase[(size_t)(q0 + r) * D + c] : 0.f;
      Os[r * DP + c] = in ? obase[(size_t)(q0 + r) * D + c] : 0.f;
    }
    if (threadIdx.x < 64) {
      const bool in = q0 + threadIdx.x < d.Lq;
      Ls[threadIdx.x] = in ? d.lse[(size_t)bh * d.Lq + q0 + threadIdx.x] : 0.f;
      Ls[64 + threadIdx.x] = in ? d.delta[(size_t)bh * d.Lq + q0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const int qi = q0 + lane;
    const bool ok = qi < d.Lq && attn_allowed(d, qi, kj);
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < D; ++c) {
      s += Qs[lane * DP + c] * krow[c];
      dp += Os[lane * DP + c] * vrow[c];
    }
    const float p = ok ? __expf(s * d.scale - Ls[lane]) : 0.f;
    const float ds = p * (dp - Ls[64 + lane]) * d.scale;
    for (int j = 0; j < 64; ++j) {
      const float pj = __shfl(p, j), dj = __shfl(ds, j);
      if (pj != 0.f) {
#pragma unroll
        for (int u = 0; u < AF_MAXU; ++u)
          if (lane + 64 * u < D) {
            av[u] += pj * Os[j * DP + lane + 64 * u];
            ak[u] += dj * Qs[j * DP + lane + 64 * u];
          }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int u = 0; u < AF_MAXU; ++u)
      if (lane + 64 * u < D) {
        d.dk[((size_t)bh * d.Lk + kj) * D + lane + 64 * u] = ak[u];
        d.dv[((size_t)bh * d.Lk + kj) * D + lane + 64 * u] = av[u];
      }
  }
}

static int attn_f32_check(const OnirisAttnF32Args* a, const char* who) {
  ONIRIS_CHECK_ARG(a && a->q && a->k && a->v, "%s: null tensor", who);
  ONIRIS_CHECK_ARG(a->BH > 0 && a->Lq > 0 && a->Lk > 0 && a->D > 0 && a->D <= 64 * AF_MAXU, "%s: bad sizes (head width 1..256)", who);
  ONIRIS_CHECK_ARG(a->mask_mode >= 0 && a->mask_mode <= 2, "%s: mask_mode 0 / 1 / 2", who);
  if (a->mask_mode != 0) ONIRIS_CHECK_ARG(a->P > 0, "%s: tokens per frame P", who);
  if (a->mask_mode == 2)
    ONIRIS_CHECK_ARG(a->T > 0 && a->Lq == 2 * a->T * a->P && a->Lk == a->Lq && (a->P >= 128 || 128 % a->P == 0),
                     "%s: training mask needs Lq = Lk = 2 T P and P a divisor of 128 (or >= 128)", who);
  return ONIRIS_OK;
}

static AttnF32Dev attn_f32_dev(const OnirisAttnF32Args* a) {
  AttnF32Dev d{};
  d.q = a->q; d.k = a->k; d.v = a->v; d.out = a->out; d.lse = a->lse;
  d.dout = a->dout; d.delta = a->delta; d.dq = a->dq; d.dk = a->dk; d.dv = a->dv;
  d.Lq = a->Lq; d.Lk = a->Lk; d.D = a->D; d.mask_mode = a->mask_mode; d.P = a->P > 0 ? a->P : 1; d.T = a->T > 0 ? a->T : 1;
  d.q_frame_off = a->q_frame_off; d.scale = a->scale;
  return d;
}

template <typename K>
static int attn_f32_launch(K kern, const AttnF32Dev& d, int rows, int BH, size_t lds, hipStream_t stream) {
  static_assert(sizeof(void*) == 8, "");
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      oniris_set_error("attn_f32: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
      return ONIRIS_ELAUNCH;
    }
  }
  ONIRIS_KLAUNCH(kern, dim3((unsigned)cdiv(rows, AF_ROWS), (unsigned)BH), dim3(64 * AF_ROWS), lds, stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_attn_f32_fwd(const OnirisAttnF32Args* a, oniris_stream_t stream_) {
  int rc = attn_f32_check(a, "attn_f32_fwd");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(a->out && a->lse, "attn_f32_fwd: out / lse");
  const AttnF32Dev d = attn_f32_dev(a);
  const size_t lds = sizeof(float) * ((size_t)2 * 64 * (d.D + 1) + (size_t)AF_ROWS * d.D);
  return attn_f32_launch(attn_f32_fwd_kernel, d, d.Lq, a->BH, lds, (hipStream_t)stream_);
}

extern "C" int oniris_attn_f32_bwd(const OnirisAttnF32Args* a, oniris_stream_t stream_) {
  int rc = attn_f32_check(a, "attn_f32_bwd");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(a->lse && a->dout && a->delta && a->dq && a->dk && a->dv, "attn_f32_bwd: lse / dout / delta / dq / dk / dv");
  const AttnF32Dev d = attn_f32_dev(a);
  const size_t lds_q = sizeof(float) * ((size_t)2 * 64 * (d.D + 1) + (size_t)2 * AF_ROWS * d.D);
  rc = attn_f32_launch(attn_f32_dq_kernel, d, d.Lq, a->BH, lds_q, (hipStream_t)stream_);
  if (rc) return rc;
  return attn_f32_launch(attn_f32_dkv_kernel, d, d.Lk, a->BH, lds_q + sizeof(float) * 128, (hipStream_t)stream_);
}
