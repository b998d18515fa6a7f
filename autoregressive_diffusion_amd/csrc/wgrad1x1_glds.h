// Weight gradient of the 1x1 convolutions (attention qkv / proj, skip convs, the grouped emb_linear GEMM):
//   dW[co][ci] = sum_pos dy[pos][co] * x[pos][ci]
// a GEMM with a huge reduction dimension (positions) and a small output.  The 3x3 kernel's 64x64 output tile gives a
// wave 8 MFMAs per staged position tile -- latency-bound (~100 TFLOP/s).  Here a workgroup owns a 128 co x 128 ci
// output tile (each of the 4 waves of a K-group a 64x64 quarter = 2x2 MFMA tiles, so every fragment feeds two MFMAs),
// stages 64-position tiles of dy and x with LDS-DMA (256-byte rows whose four 64-byte granules are XOR-swizzled with
// row&3 on the source side: the four rows of a transposing read hit disjoint banks), double-buffered, two K-groups
// per workgroup (8 waves), partial sums of the groups summed in LDS, one fp32 slab per workgroup column.
#pragma once
#include "lds_dma.h"

template <int NG>
__global__ __launch_bounds__(256 * NG, 2) void wgrad1x1_glds_kernel(const WgradDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KT = 64, ROWB = 256, TILEB = KT * ROWB, BUFB = 2 * TILEB;       // dy tile | x tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[NG * 2 * BUFB];
  const OnirisWgradArgs& a = d.a[0];
  const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 8, gtid = tid & 255, wave4 = (tid >> 6) & 3;
  const int wr = wave4 & 1, wc = wave4 >> 1;
  const int Cin = a.Cin, Cout = a.Cout;
  const int cib = blockIdx.y % d.ncib, cob = blockIdx.y / d.ncib;
  const int co0 = cob * 128, ci0 = cib * 128;
  const int bx = blockIdx.x, gxg = gridDim.x;
  const int mtot = a.B * a.T * a.H * a.W;                   // positions
  const int ntiles = (mtot + KT - 1) / KT;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  constexpr int OOB = (int)0x80000000;
  constexpr int NI = KT * 16 / 256;                         // pieces per thread and tensor
  int dvoff[NI], xvoff[NI], prow[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = i * 256 + gtid;
    const int row = e >> 4, gp = (e & 15) ^ ((row & 3) << 2);
    prow[i] = row;
    dvoff[i] = (co0 + gp * 8 < Cout) ? (row * Cout + co0 + gp * 8) * 2 : OOB;
    xvoff[i] = (ci0 + gp * 8 < Cin) ? (row * Cin + ci0 + gp * 8) * 2 : OOB;
  }
  const i32x4 rs_dy = make_rsrc(a.dy, mtot * Cout * 2);
  const i32x4 rs_x = make_rsrc(a.x, mtot * Cin * 2);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;

  auto issue = [&](int tile, int bsel) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (kg * 2 + bsel) * BUFB + wave4 * 1024);
    const int q0 = tile * KT, left = mtot - q0;             // rows >= left are past the end: zeros
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool ok = prow[i] < left;
      dma16(rs_dy, ok ? dvoff[i] : OOB, q0 * Cout * 2, dst + i * 4096);
      dma16(rs_x, ok ? xvoff[i] : OOB, q0 * Cin * 2, dst + TILEB + i * 4096);
    }
  };

  // transposing-read addresses (lane -> row q of a 4-row group, 8-byte column slot); granule swizzle = q << 6
  const int hh = lane >> 5, q = (lane & 15) >> 2;
  const int cslot = (lane & 3) * 8 + 32 * ((lane >> 4) & 1);
  int aoff[2], boff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    aoff[m] = (8 * hh + q) * ROWB + ((wr * 128 + m * 64 + cslot) ^ (q << 6));
    boff[m] = TILEB + (8 * hh + q) * ROWB + ((wc * 128 + m * 64 + cslot) ^ (q << 6));
  }
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto trf = [&](const unsigned char* p0) __attribute__((always_inline)) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * ROWB));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };

  const int tstep = gxg * NG;
  int tile = bx * NG + kg, bsel = 0;
  if (tile < ntiles) issue(tile, 0);
#pragma unroll 1
  for (int t0 = bx * NG; t0 < ntiles; t0 += tstep) {
    const bool have = tile < ntiles;
    dma_wait();
    __syncthreads();
    if (tile + tstep < ntiles) issue(tile + tstep, bsel ^ 1);
    if (have) {
      const unsigned char* buf = smem + (kg * 2 + bsel) * BUFB;
      bf16x8 af[2][2], bf_[2][2];                           // [k-step parity][tile]
#pragma unroll
      for (int m = 0; m < 2; ++m) { af[0][m] = trf(buf + aoff[m]); bf_[0][m] = trf(buf + boff[m]); }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int ks = 0; ks < KT / 16; ++ks) {
        if (ks + 1 < KT / 16) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            af[(ks + 1) & 1][m] = trf(buf + aoff[m] + (ks + 1) * 16 * ROWB);
            bf_[(ks + 1) & 1][m] = trf(buf + boff[m] + (ks + 1) * 16 * ROWB);
          }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(af[ks & 1][m], bf_[ks & 1][n], acc[m][n]);
        if (ks + 1 < KT / 16) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    tile += tstep;
    bsel ^= 1;
  }

  if (bx == 0 && blockIdx.y == 0 && tid == 0 && a.nsplit_out) *a.nsplit_out = gxg;
  if constexpr (NG > 1) {
    float* red = (float*)smem;                             // [4 waves][16][64] floats per 32x32 tile
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        __syncthreads();
        if (kg == 1) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) red[(wave4 * 16 + rr) * 64 + lane] = acc[m][n][rr];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) acc[m][n][rr] += red[(wave4 * 16 + rr) * 64 + lane];
        }
      }
    if (kg != 0) return;
  }
  bf16* slab = (bf16*)a.dwp + (size_t)bx * a.taps_total * a.CoutP * a.CinP + (size_t)a.tap0 * a.CinP;   // [co][tap][ci]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int cj = ci0 + wc * 64 + n * 32 + (lane & 31);
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int co = co0 + wr * 64 + m * 32 + mfma_row(rr, lane);
        if (co < a.CoutP && cj < a.CinP) slab[(size_t)co * a.taps_total * a.CinP + cj] = f2bf(acc[m][n][rr]);
      }
    }
#endif
}

static inline bool wgrad1x1_glds_ok(const OnirisWgradArgs& a) {
  const long long m = (long long)a.B * a.T * a.H * a.W;
  return a.taps == 1 && a.scale == nullptr && a.coff == 0 && a.xb_stride == a.T && a.x_T == a.T && a.Cin >= 64 &&
         a.Cout >= 64 && m * a.Cout * 2 < (1LL << 31) && m * a.Cin * 2 < (1LL << 31);
}

static int launch_wgrad1x1_glds(const OnirisWgradArgs& a, hipStream_t stream) {
#ifndef WGRAD1X1_NG
#define WGRAD1X1_NG 2                        // K-groups per workgroup (1: two 4-wave workgroups per CU; A/B)
#endif
  constexpr int NG = WGRAD1X1_NG;
  WgradDev d;
  memset(&d, 0, sizeof(d));
  d.a[0] = a;
  d.ncib = cdiv(a.Cin, 128);
  const int gy = d.ncib * cdiv(a.Cout, 128);
  const long long m = (long long)a.B * a.T * a.H * a.W;
  const int ntiles = (int)((m + 63) / 64);
  int gx = (256 * 2 / NG) / gy;                             // 8 waves per CU in total; one slab per column
  if (gx > a.nsplit_cap) gx = a.nsplit_cap;
  if (gx * NG > ntiles) gx = (ntiles + NG - 1) / NG;
  if (gx < 1) gx = 1;
  auto kern = wgrad1x1_glds_kernel<NG>;
  oniris_launch(kern, dim3(gx, gy), dim3(256 * NG), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
