// Weight gradient of a GATED causal 3x3 convolution (own-frame weight + both context taps) for 32 x 32-channel weight
// blocks -- the 64x64-pixel level of the UNets, where the tensors are largest and a 32-channel row gives the matrix pipe
// little to do per byte -- as ONE streaming pass over the frames of a sequence.
//
// conv_wgrad_glds_kernel<1,1,2,16> treats the three problems (dW2 += x[s,t]^T ca dout[s,t];  dW3_j += x[0,t-2+j]^T dy3[t])
// as three groups of independent 128-position tiles: every tile copies its own dy tile and x halo (19.7 KB for 18 MFMAs per
// wave), the clean frames are copied three times (own, context of t+1, context of t+2), and a K-group of four waves meets
// at a barrier every 18 MFMAs.  Measured 442 TFLOP/s = 0.18 of the bf16 peak at 2.4 TB/s: bound by neither.
//
// Here a workgroup owns one (sequence, 8x16-pixel tile, 32ci x 32co block) and WALKS THE FRAMES of a segment of the
// sequence: per frame it copies dout[s0,t], dout[s1,t], dy3[t] (3 x 8 KB) and the halos of x[s0,t], x[s1,t] (2 x 11.5 KB);
// the halos of x[s0,t-1], x[s0,t-2] are still in LDS (a four-slot ring).  47 KB per frame instead of 78.8 KB, 36 MFMAs per
// wave and barrier instead of 18, and all 27 taps of the weight block leave the workgroup as ONE slab per weight.
//   waves 0..3: dW2   (slot w >> 1, pixel rows 4 (w & 1) .. +3 of the tile)         144 accumulators each
//   waves 4, 5: dW3[0] (context t-2; pixel rows 4 (w & 1) ..)
//   waves 6, 7: dW3[1] (context t-1)
// Same arithmetic per partial sum as the tile kernel (bf16 operands, the per-frame coefficient folded into the dy
// fragment with one bf16 rounding, fp32 accumulation); the summation ORDER differs (frames of a segment first).
#pragma once
#include "lds_dma.h"

struct WgradStreamDev {
  const void* x;          // [B][2][T][H][W][Cin]
  const void* dout;       // [B][2][T][H][W][Cout]
  const void* dy3;        // [B][T][H][W][Cout]
  const float* scale;     // [B*2*T] (ca) or NULL
  void* dwp2;             // slabs of the own-frame weight   [slab][CoutP][9][CinP]
  void* dwp3;             // slabs of the context weight     [slab][CoutP][18][CinP]
  int32_t* nsplit2;
  int32_t* nsplit3;
  int B, T, H, W, Cin, CinP, Cout, CoutP;
  int ntx, nty, nseg, seglen, ncib;
  float fill;
};

// PH = 8: 8x16-pixel tiles, one 8-wave workgroup per CU (rounds 4-5).  PH = 4: 4x16-pixel tiles, 4 waves (dW2 of slot 0 / slot 1, dW3[0],
// dW3[1]: every wave the whole tile), 66 KB of LDS: TWO independent workgroups per CU that do not share a barrier (the lesson of
// conv_wgrad_glds_kernel's K-groups, profiles/r05_ab_wgrad_groups.txt); +20 % halo rows, twice the slabs.
template <int PH>
__global__ __launch_bounds__(64 * PH, 2) void conv_wgrad_stream_kernel(const WgradStreamDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(PH == 8 || PH == 4, "tile heights");
  constexpr int NTHR = 64 * PH, NPOS = PH * 16;
  constexpr int TAPS = 9, HW_ = 18, HALO = (PH + 2) * HW_, XB = HALO * 64, DB = NPOS * 64;
  constexpr int XS0 = 0, XS1 = 4 * XB, DY0 = 6 * XB;               // x[s0] ring (4) | x[s1] (2) | dy (2 x 3)
  constexpr int LDS_BYTES = 6 * XB + 6 * DB;
  static_assert(LDS_BYTES <= 160 * 1024 && 5 * 3 * 16 * 64 * 4 <= LDS_BYTES, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = d.H, W = d.W, T = d.T, HWp = H * W, Cin = d.Cin, Cout = d.Cout;
  const int cib = blockIdx.y % d.ncib, cob = blockIdx.y / d.ncib;
  const int co0 = cob * 32, ci0 = cib * 32;
  int u = blockIdx.x;
  const int seg = u % d.nseg; u /= d.nseg;
  const int x0 = (u % d.ntx) * 16; u /= d.ntx;
  const int y0 = (u % d.nty) * PH;
  const int b = u / d.nty;
  const int t0 = seg * d.seglen, t1 = min(T, t0 + d.seglen);

  // ---- roles
  const int role = (PH == 8) ? ((wave < 4) ? 0 : (wave < 6 ? 1 : 2)) : ((wave < 2) ? 0 : wave - 1);     // own | ctx t-2 | ctx t-1
  const int slot = (PH == 8) ? ((wave < 4) ? (wave >> 1) : 0) : ((wave < 2) ? wave : 0), half = (PH == 8) ? (wave & 1) : 0;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- DMA descriptors (per lane, frame-invariant)
  constexpr int OOB = (int)0x80000000;
  int dvoff;                                                       // dy tile: NPOS * 4 pieces = one per thread
  {
    const int row = tid >> 2, gp = tid & 3, co = co0 + gp * 8;
    dvoff = (co < Cout) ? (((row >> 4) * W + (row & 15)) * Cout + co) * 2 : OOB;
  }
  int xvoff[2];                                                    // halo: 720 (432) pieces
  bool xok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = i * NTHR + tid, row = e >> 2, gp = e & 3, ci = ci0 + gp * 8;
    const int hy = row / HW_, hx = row % HW_;
    xok[i] = e < HALO * 4 && ci < Cin && (unsigned)(y0 + hy - 1) < (unsigned)H && (unsigned)(x0 + hx - 1) < (unsigned)W;
    xvoff[i] = xok[i] ? (((y0 + hy - 1) * W + (x0 + hx - 1)) * Cin + ci) * 2 : OOB;
  }
  const i32x4 rs_do = make_rsrc(d.dout, d.B * 2 * T * HWp * Cout * 2);
  const i32x4 rs_d3 = make_rsrc(d.dy3, d.B * T * HWp * Cout * 2);
  const i32x4 rs_x = make_rsrc(d.x, d.B * 2 * T * HWp * Cin * 2);
  const i32x4 rs_f = make_rsrc(oniris_fill_rows, 128);
  const int fillsel = (d.fill != 0.f) ? 64 : 0;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  const unsigned wdst = lds0 + wave * 1024;
  const int dy_origin = (y0 * W + x0) * Cout * 2;

  auto issue_halo = [&](int s, int f, unsigned dst) __attribute__((always_inline)) {     // frame f of slot s (f < 0: padding)
    if (f >= 0) {
      const int so = (((b * 2 + s) * T + f) * HWp * Cin) * 2;
      dma16(rs_x, xvoff[0], so, dst + wdst);
      if (wave < (HALO * 4 - NTHR + 63) / 64) { if (NTHR + tid < HALO * 4) dma16(rs_x, xvoff[1], so, dst + wdst + NTHR * 16); }
    } else {
      dma16(rs_f, xok[0] ? fillsel : OOB, 0, dst + wdst);
      if (wave < (HALO * 4 - NTHR + 63) / 64) { if (NTHR + tid < HALO * 4) dma16(rs_f, xok[1] ? fillsel : OOB, 0, dst + wdst + NTHR * 16); }
    }
  };
  auto issue_frame = [&](int f) __attribute__((always_inline)) {    // everything frame f brings: 2 halos + 3 dy tiles
    issue_halo(0, f, XS0 + (f & 3) * XB);
    issue_halo(1, f, XS1 + (f & 1) * XB);
    const unsigned dyb = DY0 + (f & 1) * 3 * DB;
    dma16(rs_do, dvoff, (((b * 2 + 0) * T + f) * HWp * Cout) * 2 + dy_origin, dyb + wdst);
    dma16(rs_do, dvoff, (((b * 2 + 1) * T + f) * HWp * Cout) * 2 + dy_origin, dyb + DB + wdst);
    dma16(rs_d3, dvoff, ((b * T + f) * HWp * Cout) * 2 + dy_origin, dyb + 2 * DB + wdst);
  };

  // ---- fragment addresses (transposing reads; see conv_wgrad.hip): lane -> (row q of its 8-row half, 8-byte column slot)
  const int hh = lane >> 5, q = (lane & 15) >> 2;
  const int cslot = (lane & 3) * 8 + 32 * ((lane >> 4) & 1);
  const int dya = (half * 64 + 8 * hh + q) * 64 + cslot;           // + j*16*64: pixel row j of this wave's four
  int xa[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) xa[kx] = (half * 4 * HW_ + 8 * hh + q + kx) * 64 + cslot;     // + (j + ky) * 18 * 64

  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto trf = [&](const unsigned char* p0) __attribute__((always_inline)) {       // rows r..r+3 and r+4..r+7 (64-byte rows)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 256));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };

  // ---- prologue: the two frames in front of the segment (padding frames in front of the sequence), then frame t0
  const bool scaled = role == 0 && d.scale != nullptr;
  float sc_next = 1.f;
  if (t0 < t1) {
    if (scaled) sc_next = d.scale[(b * 2 + slot) * T + t0];
    asm volatile("" : "+v"(sc_next));
    issue_halo(0, t0 - 2, XS0 + ((t0 - 2) & 3) * XB);
    issue_halo(0, t0 - 1, XS0 + ((t0 - 1) & 3) * XB);
    issue_frame(t0);
  }
#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    dma_wait();
    __syncthreads();                         // frame t has landed for everybody; everybody is done with frame t-1
    float sc = sc_next;
    asm volatile("" : "+v"(sc));             // consume the coefficient before the next DMA goes out (see conv_glds.h)
    if (t + 1 < t1) {
      if (scaled) sc_next = d.scale[(b * 2 + slot) * T + t + 1];
      asm volatile("" : "+v"(sc_next));
      issue_frame(t + 1);
    }
    const unsigned char* abuf = smem + DY0 + (t & 1) * 3 * DB + ((role == 0) ? slot * DB : 2 * DB);
    const unsigned char* bbuf = smem + ((role == 0) ? (slot ? XS1 + (t & 1) * XB : XS0 + (t & 3) * XB)
                                                    : XS0 + ((t - 3 + role) & 3) * XB);
    constexpr int NK = 4, NSTEP = NK * TAPS, LA = 3;
    bf16x8 af[2], bfm[4];
    auto ld_a = [&](int kb, int j) __attribute__((always_inline)) {
      bf16x8 v = trf(abuf + dya + j * 16 * 64);
      if (scaled) {                          // per-frame coefficient folded into dy (bf16 rounding, like a dy2 tensor)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = f2bf(bf2f(v[k]) * sc);
      }
      af[kb] = v;
    };
    auto ld_b = [&](int fb, int st) __attribute__((always_inline)) {
      const int j = st / TAPS, tap = st % TAPS, ky = tap / 3, kx = tap % 3;
      bfm[fb] = trf(bbuf + xa[kx] + (j + ky) * HW_ * 64);
    };
#ifndef WGRAD_STREAM_ROWORDER
#define WGRAD_STREAM_ROWORDER 1            // 0: every (pixel row, tap) reads its own x fragment (round 4's loop; A/B)
#endif
    if constexpr (WGRAD_STREAM_ROWORDER != 0) {
      // halo-row order (see conv_wgrad_glds.h): the x fragment of (pixel row j, tap (ky, kx)) depends on u = j + ky and kx only --
      // 18 distinct fragments per frame and wave instead of 36 reads; per tap the rows are still added in ascending order
      constexpr int NU = NK + 2, NB = NU * 3;
      bf16x8 aw[4];
      auto ld_aw = [&](int j) __attribute__((always_inline)) { ld_a(0, j); aw[j & 3] = af[0]; };
      auto ld_bu = [&](int fb, int i) __attribute__((always_inline)) { bfm[fb] = trf(bbuf + xa[i % 3] + (i / 3) * HW_ * 64); };
      ld_aw(0);
#pragma unroll
      for (int i = 0; i < LA; ++i) ld_bu(i, i);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * LA, 0);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int u = i / 3, kx = i % 3;
        int nrd = 0, nmf = 0;
        if (i + LA < NB) { ld_bu((i + LA) & 3, i + LA); nrd += 2; }
        if (kx == 0 && u + 1 < NK) { ld_aw(u + 1); nrd += 2; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int j = u - ky;
          if (j >= 0 && j < NK) { acc[ky * 3 + kx] = mfma32(aw[j & 3], bfm[i & 3], acc[ky * 3 + kx]); ++nmf; }
        }
        if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
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

  // ---- the waves that worked on the same weight meet in LDS (three taps per round), then waves 0 / 4 / 6 write the slabs
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    if (d.nsplit2) *d.nsplit2 = gridDim.x;
    if (d.nsplit3) *d.nsplit3 = gridDim.x;
  }
  const int nl = (PH == 8) ? ((wave == 1) ? 0 : (wave == 2) ? 1 : (wave == 3) ? 2 : (wave == 5) ? 3 : (wave == 7) ? 4 : -1)
                           : ((wave == 1) ? 0 : -1);             // PH 4: only dW2 has two partial sums (the slots)
  float* red = (float*)smem;                                     // [5][3][16][64]
#pragma unroll
  for (int r3 = 0; r3 < 3; ++r3) {
    __syncthreads();
    if (nl >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) red[((nl * 3 + k) * 16 + rr) * 64 + lane] = acc[r3 * 3 + k][rr];
    }
    __syncthreads();
    if (nl < 0) {
      const int first = (PH == 8) ? ((wave == 0) ? 0 : (wave == 4) ? 3 : 4) : 0;
      const int cnt = (PH == 8) ? ((wave == 0) ? 3 : 1) : ((wave == 0) ? 1 : 0);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          float v = acc[r3 * 3 + k][rr];
          for (int w = 0; w < cnt; ++w) v += red[(((first + w) * 3 + k) * 16 + rr) * 64 + lane];
          acc[r3 * 3 + k][rr] = v;
        }
    }
  }
  if (nl >= 0) return;
  const int taps_total = (role == 0) ? 9 : 18, tap0 = (role == 2) ? 9 : 0;
  const int cj = ci0 + (lane & 31);
  bf16* slab = (bf16*)((role == 0) ? d.dwp2 : d.dwp3) + (size_t)blockIdx.x * taps_total * d.CoutP * d.CinP;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    bf16* base = slab + (size_t)(tap0 + tap) * d.CinP;             // slab layout [co][tap][ci]
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int co = co0 + mfma_row(rr, lane);
      if (co < d.CoutP && cj < d.CinP) base[(size_t)co * taps_total * d.CinP + cj] = f2bf(acc[tap][rr]);
    }
  }
#endif
}

// The launch of oniris_conv_wgrad_group that the streaming kernel can take: the three groups of ONE gated conv in the DART
// training layout (own: all B*2*T frames with the per-frame coefficient; context j: dy3 against the clean frames shifted by
// coff = -2 / -1, padded with `fill`), 9 taps each, 16-pixel-wide tiles, a weight of 32-channel blocks on at least one side.
static inline bool wgrad_stream_ok(const OnirisWgradArgs* a, int ng) {
  if (ng != 3 || a[0].pad_ < 0 || (a[0].pad_ & 2)) return false;
  const OnirisWgradArgs &o = a[0], &c0 = a[1], &c1 = a[2];
  if (o.taps != 9 || o.W % 16 != 0 || o.H % 8 != 0) return false;
  if (o.Cin > 32 && o.Cout > 32) return false;                      // 64x64-channel tiles: conv_wgrad_glds_kernel<2,2>
  if (o.B != 1 || o.coff != 0 || o.tap0 != 0 || o.taps_total != 9 || o.xb_stride != o.T || o.x_T != o.T) return false;
  if (c0.x != o.x || c1.x != o.x || c0.dy != c1.dy || c0.dwp != c1.dwp || c0.scale || c1.scale) return false;
  if (c0.B != c1.B || c0.T != c1.T || o.T != 2 * c0.B * c0.T) return false;
  if (c0.xb_stride != 2 * c0.T || c1.xb_stride != 2 * c0.T || c0.x_T != c0.T || c1.x_T != c0.T) return false;
  if (c0.coff != -2 || c1.coff != -1 || c0.tap0 != 0 || c1.tap0 != 9 || c0.taps_total != 18 || c1.taps_total != 18) return false;
  if (c0.fill != c1.fill || !(c0.fill == 0.f || c0.fill == 1.f)) return false;
  if ((long long)o.T * o.H * o.W * (o.Cout > o.Cin ? o.Cout : o.Cin) * 2 >= (1LL << 31)) return false;
  return true;
}

static int launch_wgrad_stream(const OnirisWgradArgs* a, hipStream_t stream) {
  const OnirisWgradArgs &o = a[0], &c0 = a[1];
  WgradStreamDev d;
  memset(&d, 0, sizeof(d));
  d.x = o.x; d.dout = o.dy; d.dy3 = c0.dy; d.scale = o.scale;
  d.dwp2 = o.dwp; d.dwp3 = c0.dwp; d.nsplit2 = o.nsplit_out; d.nsplit3 = c0.nsplit_out;
  d.B = c0.B; d.T = c0.T; d.H = o.H; d.W = o.W; d.Cin = o.Cin; d.CinP = o.CinP; d.Cout = o.Cout; d.CoutP = o.CoutP;
#ifndef WGRAD_STREAM_PH
#define WGRAD_STREAM_PH 4                  // 8: always the 8x16-pixel form (A/B: make variant VSRC=conv_wgrad VDEF=-DWGRAD_STREAM_PH=8)
#endif
  d.ntx = o.W / 16; d.ncib = cdiv(o.Cin, 32);
  d.fill = c0.fill;
  const int ncob = cdiv(o.Cout, 32), gy = d.ncib * ncob;
  const int cap = o.nsplit_cap < c0.nsplit_cap ? o.nsplit_cap : c0.nsplit_cap;
  // 4x16-pixel tiles (two 4-wave workgroups per CU) where the weights own enough slabs, 8x16 otherwise
  int ph = (WGRAD_STREAM_PH == 4 && d.B * d.ntx * (o.H / 4) <= cap) ? 4 : 8;
  d.nty = o.H / ph;
  const int units = d.B * d.ntx * d.nty;
  if (units > cap) return 1;                                        // more slabs than the weights own: the tile kernel takes it
  // about one (two) workgroup(s) per CU: cut the sequences into segments of >= 8 frames (a segment re-copies two halos at its head)
  int nseg = ((ph == 4 ? 512 : 256) / gy) / units;
  if (nseg > d.T / 8) nseg = d.T / 8;
  if (nseg * units > cap) nseg = cap / units;
  if (nseg < 1) nseg = 1;
  d.seglen = cdiv(d.T, nseg);
  d.nseg = cdiv(d.T, d.seglen);
  if (ph == 4) oniris_launch(conv_wgrad_stream_kernel<4>, dim3(units * d.nseg, gy), dim3(256), stream, d);
  else oniris_launch(conv_wgrad_stream_kernel<8>, dim3(units * d.nseg, gy), dim3(512), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
