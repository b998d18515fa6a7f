// Weight gradient of the 3x3 (gated) convolutions on images >= 16 pixels wide, 64x64-channel output tiles:
// LDS-DMA variant of conv_wgrad_kernel<9,16,2,2> (same math, same slabs, bit-identical partial sums per tile order).
//
// Why: the register-staged kernel spends as many cycles on VALU (address math of the loads, the dy*scale pass, ~620
// VALU instructions per position tile and wave) as on its 72 MFMAs, with ONE wave per SIMD (144 accumulators + the
// staging registers) and a prefetch distance of one tile.  Here
//   * dy tile [128 positions][64 co] and x halo [180 rows][64 ci] (128-byte rows) go global -> LDS by
//     `buffer_load ... lds`; per-lane offsets are tile-invariant (dy) or one compare-select per piece (x halo), the
//     tile origin / frame / fill choice is the uniform soffset / resource;
//   * the 16-byte pieces of a row are XOR-swizzled (piece ^ 4*bit1(row)) on the SOURCE side, which makes the four
//     rows a transposing read touches land on disjoint banks without padding;
//   * a workgroup has NG K-groups of 4 waves (two waves per SIMD: 144 accumulator + < 100 other registers), each with its
//     own two LDS buffers (2 x 38.5 KB per group), one barrier per tile; NG = 2: the groups' partial sums meet in LDS at the
//     end and the slab count stays one per workgroup; NG = 1 (what ships since round 5, WGRAD_NG in conv_wgrad.hip): two
//     independent workgroups per CU that do not share a barrier -- see profiles/r05_ab_wgrad_groups.txt;
//   * the per-frame dy coefficient is applied to the A fragment in registers (same bf16 rounding as before).
#pragma once
#include "lds_dma.h"

// PW = 16: 8 x 16 pixel tiles of one frame (128 positions); PW = 8: one whole 8x8 frame (64 positions: four buffers of
// a 128-position tile of two frames would need 166 KB of LDS)
#ifndef WGRAD_ROWORDER
#define WGRAD_ROWORDER 1                     // 0: every (k-step, tap) reads its own x fragment (the round-2..4 loop; A/B: make variant)
#endif
template <int CT, int IT, int NG, int PW = 16>
__global__ __launch_bounds__(256 * NG, 2) void conv_wgrad_glds_kernel(const WgradDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool ROWORDER = WGRAD_ROWORDER != 0;
  constexpr int NPOS = (PW == 16) ? 128 : 64, NKT = NPOS / 16;         // positions and 16-position k-steps per tile
  using P = Patch<PW, NPOS>;
  constexpr int FT = P::FT, HHW = P::HH * P::HW;
  constexpr int TAPS = 9, DROWB = CT * 64, XROWB = IT * 64, DPP = CT * 4, XPP = IT * 4;     // row bytes, 16-B pieces per row
  constexpr int NTILE = CT * IT, NKS = 4 / NTILE;      // waves of a K-group sharing one 32x32 tile split its k-steps
  constexpr int DY_BYTES = NPOS * DROWB, X_BYTES = P::HALO * XROWB, BUFB = DY_BYTES + X_BYTES;
  static_assert((PW == 16 && P::HALO == 180) || (PW == 8 && P::HALO == 100), "tile geometry");
  static_assert(FT == 1 && NKT % NKS == 0, "one frame per tile");
  auto pos_pix = [&](int p, int& f_, int& py, int& px) __attribute__((always_inline)) {     // position in tile -> frame, pixel
    if constexpr (PW == 16) { f_ = 0; py = p >> 4; px = p & 15; } else { f_ = p >> 6; py = (p >> 3) & 7; px = p & 7; }   // (p < 64: f_ = 0)
  };
  __shared__ __attribute__((aligned(16))) unsigned char smem[NG * 2 * BUFB];

  int gsel = 0;
  if ((int)blockIdx.x >= d.gstart[1]) gsel = 1;
  if ((int)blockIdx.x >= d.gstart[2]) gsel = 2;
  const OnirisWgradArgs& a = d.a[gsel];
  const int bx = blockIdx.x - d.gstart[gsel], gxg = d.gstart[gsel + 1] - d.gstart[gsel];
  const int g_ntiles = d.ntiles[gsel], g_ntt = d.ntt[gsel];
  const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 8, gtid = tid & 255, wave4 = (tid >> 6) & 3;
  const int H = a.H, W = a.W, HWp = H * W, Cin = a.Cin, Cout = a.Cout;
  const int my_tile = wave4 % NTILE, my_ks = wave4 / NTILE;
  const int ct = my_tile % CT, it = my_tile / CT;
  const int cib = blockIdx.y % d.ncib, cob = blockIdx.y / d.ncib;
  const int co0 = cob * 32 * CT, ci0 = cib * 32 * IT;
  unsigned char* gbase = smem + kg * 2 * BUFB;          // this K-group's two buffers

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- DMA descriptors
  constexpr int OOB = (int)0x80000000;
  constexpr int DTOT = NPOS * DPP, DNI = (DTOT + 255) / 256, XTOT = P::HALO * XPP, XNI = (XTOT + 255) / 256;
  int dvoff[DNI], xrel[XNI], xhyx[XNI];
#pragma unroll
  for (int i = 0; i < DNI; ++i) {
    const int e = i * 256 + gtid;
    const int row = e / DPP, gp = (CT == 2) ? ((e % DPP) ^ (4 * ((row >> 1) & 1))) : (e % DPP);   // 128-B rows are swizzled
    const int co = co0 + gp * 8;
    int f_, py, px;
    pos_pix(row, f_, py, px);
    dvoff[i] = (e < DTOT && co < Cout) ? ((f_ * HWp + py * W + px) * Cout + co) * 2 : OOB;
  }
#pragma unroll
  for (int i = 0; i < XNI; ++i) {
    const int e = i * 256 + gtid;
    const int row = e / XPP, gp = (IT == 2) ? ((e % XPP) ^ (4 * ((row >> 1) & 1))) : (e % XPP);
    const int f_ = row / HHW, hy = (row % HHW) / P::HW, hx = row % P::HW;
    const int ci = ci0 + gp * 8;
    xhyx[i] = (e < XTOT && ci < Cin) ? ((f_ << 16) | (hy << 8) | hx) : -1;
    xrel[i] = ((f_ * HWp + (hy - 1) * W + (hx - 1)) * Cin + ci) * 2;
  }
  const i32x4 rs_dy = make_rsrc(a.dy, a.B * a.T * HWp * Cout * 2);
  const i32x4 rs_x = make_rsrc(a.x, a.B * a.xb_stride * HWp * Cin * 2);
  const i32x4 rs_f = make_rsrc(oniris_fill_rows, 128);
  const int fillsel = (a.fill != 0.f) ? 64 : 0;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;

  struct Tile { int b, t, y0, x0; };
  auto decode = [&](int id) __attribute__((always_inline)) {
    Tile tl;
    tl.x0 = (id % d.ntx) * P::PW; id /= d.ntx;
    tl.y0 = (id % d.nty) * P::PH; id /= d.nty;
    tl.t = (id % g_ntt) * FT; tl.b = id / g_ntt;
    return tl;
  };
  auto issue = [&](const Tile& tl, int bsel) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (kg * 2 + bsel) * BUFB + wave4 * 1024);
    const int so_dy = (((tl.b * a.T + tl.t) * HWp + tl.y0 * W + tl.x0) * Cout) * 2;
#pragma unroll
    for (int i = 0; i < DNI; ++i) {
      int v = dvoff[i];
      if constexpr (FT > 1) { if (tl.t + ((i * 256 + gtid) / DPP >> 6) >= a.T) v = OOB; }     // ragged last tile
      if (i * 256 + gtid < DTOT) dma16(rs_dy, v, so_dy, dst + i * 4096);
    }
    const int origin = (tl.y0 * W + tl.x0) * Cin * 2;
    if constexpr (FT == 1) {
      const int f = tl.t + a.coff;
      const bool real = f >= 0 && f < a.x_T;
      const int so_x = real ? ((tl.b * a.xb_stride + f) * HWp * Cin) * 2 : 0;
#pragma unroll
      for (int i = 0; i < XNI; ++i) {
        if (i * 256 + gtid < XTOT) {
          const int hy = (xhyx[i] >> 8) & 255, hx = xhyx[i] & 255;
          const bool ok = xhyx[i] >= 0 && (unsigned)(tl.y0 + hy - 1) < (unsigned)H && (unsigned)(tl.x0 + hx - 1) < (unsigned)W;
          if (real) dma16(rs_x, ok ? xrel[i] + origin : OOB, so_x, dst + DY_BYTES + i * 4096);
          else dma16(rs_f, ok ? fillsel : OOB, 0, dst + DY_BYTES + i * 4096);
        }
      }
    } else {                                               // the frame (real / padded) is a per-row property
      const int so_x = (tl.b * a.xb_stride * HWp * Cin) * 2;
      const int shift = (tl.t + a.coff) * HWp * Cin * 2;   // may be negative; voffset + shift >= 0 on the real rows
#pragma unroll
      for (int i = 0; i < XNI; ++i) {
        if (i * 256 + gtid < XTOT) {
          const int hy = (xhyx[i] >> 8) & 255, hx = xhyx[i] & 255, f = tl.t + a.coff + (xhyx[i] >> 16);
          const bool ok = xhyx[i] >= 0 && (unsigned)(tl.y0 + hy - 1) < (unsigned)H && (unsigned)(tl.x0 + hx - 1) < (unsigned)W;
          if (f >= 0 && f < a.x_T) dma16(rs_x, ok ? xrel[i] + origin + shift : OOB, so_x, dst + DY_BYTES + i * 4096);
          else dma16(rs_f, ok ? fillsel : OOB, 0, dst + DY_BYTES + i * 4096);
        }
      }
    }
  };

  // ---- fragment addresses (transposing reads: lane -> (row q, 8-byte column slot); see conv_wgrad.hip)
  // k-step of this wave: ks = ksi*NKS + my_ks; the my_ks part of every row offset / swizzle parity lives in the bases
  const int hh = lane >> 5, q = (lane & 15) >> 2;
  const int cslot = (lane & 3) * 8 + 32 * ((lane >> 4) & 1);
  const int dsw = (CT == 2) ? 64 * ((q >> 1) & 1) : 0;
  const int dya = (my_ks * 16 + 8 * hh + q) * DROWB + ((ct * 64 + cslot) ^ dsw);          // + ksi*NKS*16*DROWB (+4 rows)
  // halo row of position p0 = ks*16 + 8*hh + q under tap (ky,kx):
  //   PW 16: (ks + ky)*18 + 8*hh + q + kx            swizzle parity = (ks + ky + ((q+kx)>>1)) & 1
  //   PW  8: 100*(ks>>2) + 10*(2*(ks&3) + hh + ky) + q + kx      parity = (hh + ky + ((q+kx)>>1)) & 1
  int xa[3][2];                                            // [kx][parity of the compile-time part]
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int lanepar = (PW == 16) ? my_ks : hh;
      const int xsw = (IT == 2) ? 64 * ((par ^ lanepar ^ ((q + kx) >> 1)) & 1) : 0;
      const int rbase = (PW == 16) ? my_ks * P::HW + 8 * hh + q + kx : 20 * my_ks + 10 * hh + q + kx;
      xa[kx][par] = DY_BYTES + rbase * XROWB + ((it * 64 + cslot) ^ xsw);
    }

  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto trf = [&](const unsigned char* p0, int rowb) __attribute__((always_inline)) {       // rows r..r+3 and r+4..r+7
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * rowb));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };

  // ---- tile loop: K-group kg takes tiles (bx + j*gxg)*NG + kg
  const int tstep = gxg * NG;
  int tile = bx * NG + kg;
  int bsel = 0;
  float sc_next[FT];
#pragma unroll
  for (int f_ = 0; f_ < FT; ++f_) sc_next[f_] = 1.f;
  auto load_scale = [&](const Tile& tl) __attribute__((always_inline)) {
#pragma unroll
    for (int f_ = 0; f_ < FT; ++f_)
      if (a.scale) sc_next[f_] = a.scale[tl.b * a.T + min(tl.t + f_, a.T - 1)];
  };
  Tile cur = decode(tile < g_ntiles ? tile : 0);
  if (tile < g_ntiles) {
    load_scale(cur);
    issue(cur, 0);
  }
#pragma unroll 1
  for (int t0 = bx * NG; t0 < g_ntiles; t0 += tstep) {
    const bool have = tile < g_ntiles;
    dma_wait();
    __syncthreads();                       // this tile has landed for everybody; buffer bsel^1 is free again
    float sc[FT];
#pragma unroll
    for (int f_ = 0; f_ < FT; ++f_) {
      sc[f_] = sc_next[f_];
      asm volatile("" : "+v"(sc[f_]));    // consume the coefficient before the next DMA goes out (see conv_glds.h)
    }
    const int ntile = tile + tstep;
    if (ntile < g_ntiles) {
      const Tile nx = decode(ntile);
      load_scale(nx);
      issue(nx, bsel ^ 1);
    }
    if (have) {
      const unsigned char* buf = gbase + bsel * BUFB;
      constexpr int NK = NKT / NKS, NSTEP = NK * TAPS, LA = 3;
      bf16x8 af[2], bfm[4];
      auto ld_a = [&](int kb, int ksi) __attribute__((always_inline)) {
        bf16x8 v = trf(buf + dya + ksi * NKS * 16 * DROWB, DROWB);
        if (a.scale) {                    // per-frame coefficient folded into dy (bf16 rounding, like a dy2 tensor)
          const float scf = sc[(FT > 1) ? ((ksi * NKS) >> 2) : 0];       // a k-step never straddles frames
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = f2bf(bf2f(v[k]) * scf);
        }
        af[kb] = v;
      };
      auto ld_b = [&](int fb, int st) __attribute__((always_inline)) {
        const int ksi = st / TAPS, tap = st % TAPS, ky = tap / 3, kx = tap % 3;
        constexpr int dummy = 0; (void)dummy;
        const int ksc = ksi * NKS;                                          // compile-time part of the k-step
        const int par = (PW == 16) ? ((ksc + ky) & 1) : (ky & 1);
        const int rimm = (PW == 16) ? (ksc + ky) * P::HW : 100 * (ksc >> 2) + 20 * (ksc & 3) + 10 * ky;
        bfm[fb] = trf(buf + xa[kx][par] + rimm * XROWB, XROWB);
      };
      if constexpr (NKS == 1 && ROWORDER) {
        // HALO-ROW ORDER (a wave that owns every k-step of its 32x32 tile).  The x fragment of (k-step ks, tap (ky, kx)) depends on
        // u = ks + ky (PW 16: halo row u) resp. u = 2 ks + ky (PW 8: halo rows u, u + 1) and kx only: walking (u, kx) instead of
        // (ks, tap) reads each of the 30 (27) distinct fragments ONCE and feeds it to up to three MFMAs -- 76 (70) transposing LDS
        // reads per tile and wave instead of 160 (88), in a loop whose two waves per SIMD issue 2.2 LDS reads per MFMA against
        // ~1 the LDS pipe can serve.  Per tap the k-steps are still added in ascending order: the partial sums are bit-identical.
        constexpr int NU = (PW == 16) ? NK + 2 : 2 * NK + 1, NB = NU * 3, AW = (PW == 16) ? 4 : 2;
        bf16x8 aw[AW];
        auto ld_aw = [&](int ks) __attribute__((always_inline)) { ld_a(0, ks); aw[ks % AW] = af[0]; };
        auto ld_bu = [&](int fb, int j) __attribute__((always_inline)) {
          const int u = j / 3, kx = j % 3;
          const int par = u & 1;
          const int rimm = (PW == 16) ? u * P::HW : 10 * u;
          bfm[fb] = trf(buf + xa[kx][par] + rimm * XROWB, XROWB);
        };
        ld_aw(0);
#pragma unroll
        for (int j = 0; j < LA; ++j) ld_bu(j, j);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * LA, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int u = j / 3, kx = j % 3;
          int nrd = 0, nmf = 0;
          if (j + LA < NB) { ld_bu((j + LA) & 3, j + LA); nrd += 2; }
          // the dy fragment of k-step ks is first used at u = ks (PW 16) / u = 2 ks (PW 8): requested one u earlier, into the slot
          // of the k-step whose last use (ky = 2) lies behind
          const int ksn = (PW == 16) ? u + 1 : (u + 1) / 2;
          if (kx == 0 && ksn < NK && ((PW == 16) || (u & 1))) { ld_aw(ksn); nrd += 2; }
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int num = u - ky, ks = (PW == 16) ? num : num / 2;
            if (num >= 0 && ks < NK && ((PW == 16) || (num & 1) == 0)) {
              acc[ky * 3 + kx] = mfma32(aw[ks % AW], bfm[j & 3], acc[ky * 3 + kx]);
              ++nmf;
            }
          }
          if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // (the builtin takes literals)
          else if (nrd == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          if (nmf == 1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          else if (nmf == 2) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
      } else {
      ld_a(0, 0);
#pragma unroll
      for (int j = 0; j < LA; ++j) ld_b(j, j);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * LA, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int ks = st / TAPS, tap = st % TAPS;
        const int nxs = st + LA;
        if (nxs < NSTEP) {
          if (nxs % TAPS == 0) ld_a((nxs / TAPS) & 1, nxs / TAPS);
          ld_b(nxs & 3, nxs);
        }
        acc[tap] = mfma32(af[ks & 1], bfm[st & 3], acc[tap]);
        if (nxs < NSTEP) {
          if (nxs % TAPS == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      }
    }
    tile = ntile;
    bsel ^= 1;
  }

  // ---- the NKS*NG waves that worked on the same 32x32 tile meet in LDS (tap by tap), then one of them writes the slab
  if (bx == 0 && blockIdx.y == 0 && tid == 0 && a.nsplit_out) *a.nsplit_out = gxg;
  constexpr int NPART = NKS * NG;
  if constexpr (NPART > 1) {
    const int part = kg * NKS + my_ks;                      // 0 = the wave that keeps the sum
    float* red = (float*)smem;                             // [NPART-1][NTILE][16][64] floats
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      __syncthreads();
      if (part > 0) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) red[(((part - 1) * NTILE + my_tile) * 16 + rr) * 64 + lane] = acc[tap][rr];
      }
      __syncthreads();
      if (part == 0) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          float v = acc[tap][rr];
#pragma unroll
          for (int w = 0; w < NPART - 1; ++w) v += red[((w * NTILE + my_tile) * 16 + rr) * 64 + lane];
          acc[tap][rr] = v;
        }
      }
    }
    if (part != 0) return;
  }
  const int cj = ci0 + it * 32 + (lane & 31);
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
#endif
}

// true when every group of the launch can take the LDS-DMA kernel
static inline bool wgrad_glds_ok(const OnirisWgradArgs* args, int ng) {
  for (int g = 0; g < ng; ++g) {
    const OnirisWgradArgs& a = args[g];
    if (a.taps != 9 || !((a.W % 16 == 0 && a.H % 8 == 0) || (a.W == 8 && a.H == 8))) return false;
    if (!(a.fill == 0.f || a.fill == 1.f)) return false;
    if ((long long)a.B * a.T * a.H * a.W * a.Cout * 2 >= (1LL << 31)) return false;
    if ((long long)a.B * a.xb_stride * a.H * a.W * a.Cin * 2 >= (1LL << 31)) return false;
  }
  return true;
}
