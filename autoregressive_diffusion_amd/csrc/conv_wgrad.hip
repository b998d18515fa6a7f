// Weight gradient of the (gated causal) convolution as an implicit GEMM over positions:
//   dW[tap][co][ci] += sum_pos dy[pos][co] * x[pos + tap][ci]          (MFMA 32x32x16 bf16 -> fp32)
// Both operands are channel-contiguous in HBM but the reduction runs over positions, so tiles are staged
// row-major ([position][channel]) in LDS and the k-contiguous MFMA fragments are produced by the gfx950
// transposing LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group).  The x tile is the halo image
// of the patch, shared by the 9 taps.  A workgroup owns a (32*CT co) x (32*IT ci) x 9-tap output tile and walks
// a strided subset of the position tiles (split-K) and writes its partial sums to ITS OWN fp32 slab with plain stores
// (two 128-byte row segments per wave store); the slabs are summed by weight_bwd_kernel.  The first version used
// fp32 atomics, which run at ~1.3 TB/s chip-wide and took about half of the kernel time.
#include "conv_kernels.h"

// Up to three sub-problems of identical geometry (H, W, channels, taps) share one launch: the own-frame weight and the
// two context taps of a gated conv.  Workgroup columns [gstart[g], gstart[g+1]) belong to group g; every column
// writes ONE slab, so one launch writes as many slabs as three separate ones used to write each.
#define WGRAD_MAXG 3
struct WgradDev {
  OnirisWgradArgs a[WGRAD_MAXG];
  int ntiles[WGRAD_MAXG], ntt[WGRAD_MAXG], gstart[WGRAD_MAXG + 1];
  int ntx, nty, ncib;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base0, const unsigned char* base1) {
  // two transposing reads -> 8 k-contiguous bf16 for this lane's channel
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

#include "conv_wgrad_glds.h"
#include "conv_wgrad_stream.h"
#include "wgrad1x1_glds.h"

template <int TAPS, int PW, int CT, int IT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradDev d) {
  using P = Patch<PW>;
  constexpr int NTILE = CT * IT;          // 32x32 output tiles per tap in this workgroup
  constexpr int NKS = 4 / NTILE;          // waves sharing one tile split the k-steps
  constexpr int DYB = CT * 64, XB = IT * 64;
  constexpr int DY_ROWB = (CT == 1) ? 64 : 192;
  constexpr int X_ROWB = (IT == 1) ? 64 : 192;
  constexpr int XROWS = (TAPS == 9) ? P::HALO : 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * DY_ROWB + XROWS * X_ROWB];
  unsigned char* dy_lds = smem;
  unsigned char* x_lds = smem + 128 * DY_ROWB;

  int gsel = 0;
  if ((int)blockIdx.x >= d.gstart[1]) gsel = 1;
  if ((int)blockIdx.x >= d.gstart[2]) gsel = 2;
  const OnirisWgradArgs& a = d.a[gsel];
  const int bx = blockIdx.x - d.gstart[gsel], gxg = d.gstart[gsel + 1] - d.gstart[gsel];
  const int g_ntiles = d.ntiles[gsel], g_ntt = d.ntt[gsel];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, T = a.T, HWp = H * W;
  const int my_tile = wave % NTILE, my_ks = wave / NTILE;
  const int ct = my_tile % CT, it = my_tile / CT;
  const int cib = blockIdx.y % d.ncib, cob = blockIdx.y / d.ncib;
  const int co0 = cob * 32 * CT, ci0 = cib * 32 * IT;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const bf16* xg = (const bf16*)a.x;
  const bf16* dyg = (const bf16*)a.dy;
  // lane roles for the transposing reads
  const int grp = lane >> 4, hh = grp >> 1, q = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);

  // Software pipeline (issue early / write late): the global loads of the NEXT position tile are issued into
  // registers before the MFMA section of the current one.
  constexpr int DPARTS = DYB / 16, DTOT = 128 * DPARTS, DNI = DTOT / 256;
  constexpr int XPARTS = XB / 16, XTOT = XROWS * XPARTS, XNI = (XTOT + 255) / 256;
  u32x4 rd[DNI], rx[XNI];
  float rsc[DNI];
  const unsigned short fb = __builtin_bit_cast(unsigned short, f2bf(a.fill));
  const unsigned fill2 = (unsigned)fb | ((unsigned)fb << 16);

  auto load_tile = [&](int tile) __attribute__((always_inline)) {
    int bid = tile;
    int t0 = 0, y0 = 0, x0 = 0, q0 = 0;
    if constexpr (TAPS == 9) {
      const int tx = bid % d.ntx; bid /= d.ntx;
      const int ty = bid % d.nty; bid /= d.nty;
      const int tc = bid % g_ntt; bid /= g_ntt;
      t0 = tc * P::FT; y0 = ty * P::PH; x0 = tx * P::PW;
    } else {
      const int tq = bid % g_ntt; bid /= g_ntt;
      q0 = tq * 128;
    }
    const int b = bid;
#pragma unroll
    for (int i = 0; i < DNI; ++i) {           // dy tile [128 positions][32*CT co]
      const int e = tid + i * 256;
      const int row = e / DPARTS, part = e % DPARTS;
      const int co = co0 + part * 8;
      rd[i] = u32x4{0u, 0u, 0u, 0u};
      rsc[i] = 1.f;
      int t; size_t pix; bool ok;
      if constexpr (TAPS == 9) {
        const int f_ = row / (P::PH * P::PW), py = (row / P::PW) % P::PH, px = row % P::PW;
        t = t0 + f_; ok = t < T; pix = (size_t)t * HWp + (y0 + py) * W + x0 + px;
      } else {
        const int qq = q0 + row;
        ok = qq < T * HWp; t = ok ? qq / HWp : 0; pix = (size_t)qq;
      }
      if (ok && co < a.Cout) {
        rd[i] = *(const u32x4*)(dyg + ((size_t)b * T * HWp + pix) * a.Cout + co);
        if (a.scale) rsc[i] = a.scale[b * T + t];
      }
    }
#pragma unroll
    for (int i = 0; i < XNI; ++i) {           // x halo [XROWS][32*IT ci]
      const int e = tid + i * 256;
      rx[i] = u32x4{0u, 0u, 0u, 0u};
      if (e < XTOT) {
        const int row = e / XPARTS, part = e % XPARTS;
        const int ci = ci0 + part * 8;
        int t, y, x; bool ok;
        if constexpr (TAPS == 9) {
          const int f_ = row / (P::HH * P::HW), rem = row % (P::HH * P::HW);
          y = y0 + rem / P::HW - 1; x = x0 + rem % P::HW - 1; t = t0 + f_;
          ok = t < T && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        } else {
          const int qq = q0 + row;
          ok = qq < T * HWp; t = ok ? qq / HWp : 0; const int pp = ok ? qq % HWp : 0; y = pp / W; x = pp % W;
        }
        if (ok && ci < a.Cin) {
          const int f = t + a.coff;
          if (f >= 0 && f < a.x_T)
            rx[i] = *(const u32x4*)(xg + ((size_t)(b * a.xb_stride + f) * HWp + y * W + x) * a.Cin + ci);
          else
            rx[i] = u32x4{fill2, fill2, fill2, fill2};
        }
      }
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < DNI; ++i) {
      const int e = tid + i * 256;
      u32x4 o = rd[i];
      if (a.scale) {                          // per-frame coefficient folded into dy (bf16 rounding, like a dy2 tensor)
        bf16x8 bv = __builtin_bit_cast(bf16x8, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) bv[k] = f2bf(bf2f(bv[k]) * rsc[i]);
        o = __builtin_bit_cast(u32x4, bv);
      }
      *(u32x4*)(dy_lds + (e / DPARTS) * DY_ROWB + (e % DPARTS) * 16) = o;
    }
#pragma unroll
    for (int i = 0; i < XNI; ++i) {
      const int e = tid + i * 256;
      if (e < XTOT) *(u32x4*)(x_lds + (e / XPARTS) * X_ROWB + (e % XPARTS) * 16) = rx[i];
    }
  };

  if (bx < g_ntiles) load_tile(bx);
#pragma unroll 1
  for (int tile = bx; tile < g_ntiles; tile += gxg) {
    store_tile();
    __syncthreads();
    if (tile + gxg < g_ntiles) load_tile(tile + gxg);
    // ---- MFMA: k = 16 positions per step; (k-step, tap) pairs form one flat sequence whose fragment reads run one
    // step ahead of the MFMAs (double-buffered registers, order pinned with sched_group_barrier)
    {
      constexpr int NK = 8 / NKS, NSTEP = NK * TAPS;
      bf16x8 af[4], bfm[4];                               // rings: the lookahead (<= 3 steps) never laps a live fragment
      auto ld_a = [&](int kb, int ksi) __attribute__((always_inline)) {
        const int ks = ksi * NKS + my_ks;
        const int p0 = ks * 16 + 8 * hh + q, p1 = p0 + 4;
        af[kb] = tr_frag(dy_lds + p0 * DY_ROWB + (ct * 32 + pcol) * 2, dy_lds + p1 * DY_ROWB + (ct * 32 + pcol) * 2);
      };
      auto ld_b = [&](int fb, int st) __attribute__((always_inline)) {
        const int ksi = st / TAPS, tap = st % TAPS;
        const int ks = ksi * NKS + my_ks;
        const int p0 = ks * 16 + 8 * hh + q, p1 = p0 + 4;
        int r0 = p0, r1 = p1;
        if constexpr (TAPS == 9) {
          r0 = ((p0 / (P::PH * P::PW)) * P::HH + (p0 / P::PW) % P::PH) * P::HW + p0 % P::PW;
          r1 = ((p1 / (P::PH * P::PW)) * P::HH + (p1 / P::PW) % P::PH) * P::HW + p1 % P::PW;
        }
        const int off = (TAPS == 9) ? ((tap / 3) * P::HW + (tap % 3)) : 0;
        bfm[fb] = tr_frag(x_lds + (r0 + off) * X_ROWB + (it * 32 + pcol) * 2, x_lds + (r1 + off) * X_ROWB + (it * 32 + pcol) * 2);
      };
      constexpr int LA = (NSTEP > 3) ? 3 : 1;            // fragment reads run LA MFMAs ahead (one wave per SIMD: the
      constexpr int NA0 = (LA + TAPS - 1) / TAPS;         // wave has to cover the LDS latency by itself)
#pragma unroll
      for (int j = 0; j < LA; ++j) {
        if (j % TAPS == 0) ld_a((j / TAPS) & 3, j / TAPS);
        ld_b(j, j);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NA0 + 2 * LA, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int ksi = st / TAPS, tap = st % TAPS;
        const int nx = st + LA;
        if (nx < NSTEP) {
          if (nx % TAPS == 0) ld_a((nx / TAPS) & 3, nx / TAPS);
          ld_b(nx & 3, nx);
        }
        acc[tap] = mfma32(af[ksi & 3], bfm[st & 3], acc[tap]);
        if (nx < NSTEP) {
          if (nx % TAPS == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
    __syncthreads();
  }

  // ---- write this workgroup column's slab  dwp[blockIdx.x][tap0 + tap][co][ci]
  // When several waves of the workgroup share one output tile (NKS > 1) they are first summed through LDS.
  if (bx == 0 && blockIdx.y == 0 && tid == 0 && a.nsplit_out) *a.nsplit_out = gxg;
  const int cj = ci0 + it * 32 + (lane & 31);
  if constexpr (NKS > 1) {
    float* red = (float*)smem;                               // [NKS-1][16][64] floats = 12 KB, staging LDS is free now
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      __syncthreads();
      if (my_ks > 0) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) red[((my_ks - 1) * 16 + rr) * 64 + lane] = acc[tap][rr];
      }
      __syncthreads();
      if (my_ks == 0) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          float v = acc[tap][rr];
#pragma unroll
          for (int w = 0; w < NKS - 1; ++w) v += red[(w * 16 + rr) * 64 + lane];
          acc[tap][rr] = v;
        }
      }
    }
    if (my_ks != 0) return;
  }
  bf16* slab = (bf16*)a.dwp + (size_t)bx * a.taps_total * a.CoutP * a.CinP;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    bf16* base = slab + (size_t)(a.tap0 + tap) * a.CinP;               // slab layout [co][tap][ci]: a weight row is contiguous
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int co = co0 + ct * 32 + mfma_row(rr, lane);
      if (co < a.CoutP && cj < a.CinP) base[(size_t)co * a.taps_total * a.CinP + cj] = f2bf(acc[tap][rr]);
    }
  }
}

template <int TAPS, int PW, int CT, int IT>
static int launch_wgrad(const OnirisWgradArgs* args, int ng, hipStream_t stream) {
  using P = Patch<PW>;
  const OnirisWgradArgs& a = args[0];
  WgradDev d;
  memset(&d, 0, sizeof(d));
  if (TAPS == 9) { d.ntx = a.W / P::PW; d.nty = a.H / P::PH; } else { d.ntx = 1; d.nty = 1; }
  d.ncib = cdiv(a.Cin, 32 * IT);
  const int ncob = cdiv(a.Cout, 32 * CT);
  const int gy = d.ncib * ncob;
  long long tot = 0;
  bool use_glds = false;                                             // LDS-DMA variant (pad_ < 0 forces the register-staged one)
  if constexpr (TAPS == 9 && (PW == 16 || PW == 8)) use_glds = a.pad_ >= 0 && wgrad_glds_ok(args, ng);
  const int ft = (use_glds && PW == 8) ? 1 : P::FT;                  // the LDS-DMA kernel takes 8x8 images one frame at a time
  for (int g = 0; g < ng; ++g) {
    d.a[g] = args[g];
    d.ntt[g] = (TAPS == 9) ? cdiv(args[g].T, ft) : cdiv(args[g].T * a.H * a.W, 128);
    d.ntiles[g] = d.ntx * d.nty * d.ntt[g] * args[g].B;
    tot += d.ntiles[g];
  }
  // ~1-2 workgroups per CU in total; the split-K columns are dealt to the groups in proportion to their position
  // tiles (every column owns one slab of ITS group's weight)
#ifndef WGRAD_NG
// K-groups of 4 waves per workgroup of the LDS-DMA kernel.  1 (round 5): two independent 4-wave workgroups per CU instead of one workgroup of
// two K-groups -- the same waves, registers and LDS per CU, but no common barrier: the groups drift apart, and one's copy issue, waits and
// fragment prologue run under the other's MFMAs (a K-group pair in lockstep had both of a SIMD's waves in the same part of the tile loop at
// any time).  Same-box A/B at B = 8: <2,2,*,16> 25.4 -> 22.4 ms per cycle (roof 0.41 -> 0.47), <2,2,*,8> 9.3 -> 8.5 ms, bench +1.3 %
// (profiles/r05_ab_wgrad_groups.txt); costs twice the split-K slabs of the 64x64-channel form (ops._nsplit_cap).
#define WGRAD_NG 1
#endif
  const int gx_budget = (CT * IT == 4) ? ((use_glds && WGRAD_NG == 1) ? 512 : 256) : 512;
  const int gx_all = gx_budget / gy > ng ? gx_budget / gy : ng;
  int gx_tot = 0;
  for (int g = 0; g < ng; ++g) {
    int gx = (int)((long long)gx_all * d.ntiles[g] / (tot > 0 ? tot : 1));
    if (gx > args[g].nsplit_cap) gx = args[g].nsplit_cap;
    if (gx > d.ntiles[g]) gx = d.ntiles[g];
    if (gx < 1) gx = 1;
    d.gstart[g] = gx_tot;
    gx_tot += gx;
  }
  for (int g = ng; g <= WGRAD_MAXG; ++g) d.gstart[g] = gx_tot;      // empty groups
  if constexpr (TAPS == 9 && (PW == 16 || PW == 8)) {
    if (use_glds) {
#ifndef WGRAD_NG1
#define WGRAD_NG1 1                          // ... of the 32x32-channel tile form (+0.15 %)
#endif
      constexpr int NG = (CT * IT == 4) ? WGRAD_NG : WGRAD_NG1;
      auto kern = conv_wgrad_glds_kernel<CT, IT, NG, PW>;
      oniris_launch(kern, dim3(gx_tot, gy), dim3(256 * NG), stream, d);
      ONIRIS_LAUNCH_CHECK();
      return ONIRIS_OK;
    }
  }
  auto kern = conv_wgrad_kernel<TAPS, PW, CT, IT>;
  oniris_launch(kern, dim3(gx_tot, gy), dim3(256), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

template <int TAPS, int PW>
static int wgrad_pick_tile(const OnirisWgradArgs* a, int ng, hipStream_t stream) {
  if (a[0].Cin > 32 && a[0].Cout > 32) return launch_wgrad<TAPS, PW, 2, 2>(a, ng, stream);
  return launch_wgrad<TAPS, PW, 1, 1>(a, ng, stream);
}

extern "C" int oniris_conv_wgrad_group(const OnirisWgradArgs* args, int ngroups, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(args && ngroups >= 1 && ngroups <= WGRAD_MAXG, "conv_wgrad: 1..%d groups", WGRAD_MAXG);
  const OnirisWgradArgs& a = args[0];
  for (int g = 0; g < ngroups; ++g) {
    const OnirisWgradArgs& b = args[g];
    ONIRIS_CHECK_ARG(b.x && b.dy && b.dwp, "conv_wgrad: null pointer (group %d)", g);
    ONIRIS_CHECK_ARG(b.nsplit_cap >= 1 && b.taps_total >= b.taps + b.tap0 && b.tap0 >= 0,
                     "conv_wgrad: bad slab description (group %d)", g);
    ONIRIS_CHECK_ARG(b.B > 0 && b.T > 0, "conv_wgrad: empty group %d", g);
    ONIRIS_CHECK_ARG(b.H == a.H && b.W == a.W && b.Cin == a.Cin && b.CinP == a.CinP && b.Cout == a.Cout &&
                     b.CoutP == a.CoutP && b.taps == a.taps, "conv_wgrad: groups must share H, W, channels and taps");
  }
  ONIRIS_CHECK_ARG(a.taps == 9 || a.taps == 1, "conv_wgrad: taps must be 1 or 9 (got %d)", a.taps);
  ONIRIS_CHECK_ARG(a.Cin % 8 == 0 && a.Cout % 8 == 0, "conv_wgrad: Cin/Cout must be multiples of 8");
  ONIRIS_CHECK_ARG(a.CoutP % 32 == 0 && a.CinP % 64 == 0 && a.CoutP >= a.Cout && a.CinP >= a.Cin,
                   "conv_wgrad: bad padded sizes");
  if (a.taps == 1) {
    if (ngroups == 1 && a.pad_ >= 0 && wgrad1x1_glds_ok(a)) return launch_wgrad1x1_glds(a, stream);
    return wgrad_pick_tile<1, 16>(args, ngroups, stream);
  }
  const int W = a.W, H = a.H;
  if (wgrad_stream_ok(args, ngroups)) {              // a gated conv's three groups on 32-channel weight blocks: one streaming pass
    const int rc = launch_wgrad_stream(args, stream);
    if (rc != 1) return rc;
  }
  if (W >= 16 && W % 16 == 0 && H % 8 == 0) return wgrad_pick_tile<9, 16>(args, ngroups, stream);
  if (W == 8 && H % 8 == 0) return wgrad_pick_tile<9, 8>(args, ngroups, stream);
  if (W == 4 && H % 4 == 0) return wgrad_pick_tile<9, 4>(args, ngroups, stream);
  if (W == 2 && H % 2 == 0) return wgrad_pick_tile<9, 2>(args, ngroups, stream);
  oniris_set_error("conv_wgrad: unsupported image size %dx%d", H, W);
  return ONIRIS_EUNSUPPORTED;
}

extern "C" int oniris_conv_wgrad(const OnirisWgradArgs* args, oniris_stream_t stream) {
  return oniris_conv_wgrad_group(args, 1, stream);
}
