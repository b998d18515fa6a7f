// Gated causal 3x3 convolution, DART training layout (S = 2 slots + 2 context frames), LDS-DMA variant for gfx950.
//
// Same math and HBM layout as conv_kernels.h (implicit GEMM, D[co][position], own / ctx0 / ctx1 phases per 32-channel
// chunk), different staging:
//   * the halo image and the weight slab of a phase go global -> LDS with `global_load_lds_dwordx4` (no VGPR staging,
//     no ds_write pass); the two LDS buffers alternate, so the DMA of phase i+1 runs under the MFMAs of phase i and
//     there is ONE barrier per phase;
//   * LDS rows are 64 B (32 channels) with no padding -- the DMA image is lane-linear -- and the four 16-byte parts
//     of a row are XOR-swizzled with row bits 2..3 on the SOURCE side (part p of row R holds channels
//     8*(p ^ ((R>>2)&3))..), which makes every ds_read_b128 of 16 consecutive rows conflict-free;
//   * out-of-image halo pixels and padded context frames read a 64-byte constant row (zeros / ones) from global
//     memory instead of branching;
//   * a wave owns MT position tiles (32 positions each) x NT channel tiles x both slots (MT = 1 is what ships: the
//     MT = 2 / 4-wave form needs 192 accumulator registers and hipcc shuffles them between VGPRs and AGPRs);
//   * workgroups are PERSISTENT: each walks a run of tiles that sit next to each other in its XCD's L2, and issues
//     the first DMA of the next tile before the epilogue of the current one, so the store tail and the cold first
//     load of a tile hide under each other (the epilogue inputs are fetched before that DMA goes out: hipcc drains
//     vmcnt to 0 at the first use of an ordinary load issued while LDS-DMA is in flight).
// Requirements (checked by the dispatcher): S == 2, context path, taps == 9, Cin % 32 == 0, CoutP % (32*NT) == 0,
// H % PH == 0, W % PW == 0, ctx_fill in {0, 1}.
#pragma once
#include <type_traits>
#include "conv_kernels.h"

#include "lds_dma.h"

// Diagnostic build only (make stamp: -DCONV_STAMP): per-phase cycle sums of every wave of workgroup 0, written to the
// buffer passed in OnirisConvArgs.splitk_ws (unused by this kernel): [wave][12] uint64 (slots 7 .. 10: the context phases' share of
// slots 2 .. 5).
#ifdef CONV_STAMP
#define CSTAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CSTAMP(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; }
#define CSTAMP2(c, i, j) { if (c) CSTAMP(i) else CSTAMP(j) }      // (static indices: st_acc stays in registers)
#else
#define CSTAMP_DECL
#define CSTAMP(i)
#define CSTAMP2(c, i, j)
#endif

// NT: 32-channel tiles per wave; WC: channel groups of waves (a workgroup covers 32*NT*WC channels and
// 32*(NW/WC)*MT positions).  PW = 16: one 16x16 tile of one frame; PW = 8: two whole 8x8 frames, and the halo rows are
// 12 entries wide (2 unused) so that the patch rows py and py+2 a 16-lane read group takes are 24 = 8 (mod 16) rows apart.
template <int NT, int PW, int NW, int MT, int WC = 1>
struct GldsCfg {
  static constexpr int NWP = NW / WC;
  static constexpr int NPOS = 32 * NWP * MT, NTHR = 64 * NW, BN = 32 * NT * WC;
  using P = Patch<PW, NPOS>;
  static constexpr int HW = (PW == 8) ? 12 : PW + 2, HH = P::PH + 2, FT = P::FT;
  static constexpr int HALO = FT * HH * HW;
  static constexpr int SROWS = (HALO + 15) / 16 * 16;         // slot stride in rows: a multiple of 16 keeps the swizzle
  static constexpr int AROWS = 2 * SROWS, CROWS = HALO, WROWS = 9 * BN;      // phase of slot 1 equal to slot 0's
  static constexpr int BUF = (AROWS + WROWS) * 64;
  static constexpr int EROW = NT * 32 * 2 + 16;              // a wave stages its own 32 x (32*NT) bf16 tile
  static constexpr int EPI = NW * 32 * EROW;
  static_assert(PW == 16 || PW == 8, "tile geometries of the LDS-DMA kernel");
  static constexpr int LDS_BYTES = (2 * BUF > EPI) ? 2 * BUF : EPI;
  static_assert(LDS_BYTES <= 160 * 1024, "two staging buffers must fit the 160 KB LDS");
  // "resident" layout (RES, one 32-channel chunk = three phases per tile): own | ctx0 | ctx1 regions + a dedicated
  // epilogue staging area, so that every region of tile t+1 is copied while tile t is still being worked on
  static constexpr int BUFC = (SROWS + WROWS) * 64;
  static constexpr int RES_EOFF = BUF + 2 * BUFC;
  static constexpr int RES_BYTES = RES_EOFF + 2 * FT * BN * 4 + EPI;
  static_assert(2 * FT * BN * 4 + EPI <= BUF, "the epilogue stages through ONE of the two buffers");
  static_assert((NWP * 2 * HW) % 16 == 0 || MT == 1, "position tiles of a wave must be 16-row aligned apart");
};

// CTX = false: plain 3x3 convolution of an even number of frames, run as "two slots, no context phases" (the 2-D
// training steps and every non-gated 3x3 conv): same tiles, one phase per 32-channel chunk.
template <int NT, int PW, int NW, int MT, int WC = 1, bool CTX = true, bool RES = false>
__global__ __launch_bounds__(64 * NW, (NW >= 8) ? 2 : 1) void conv_glds_kernel(const ConvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins do not exist in the host pass (the stub needs no body)
  using Cfg = GldsCfg<NT, PW, NW, MT, WC>;
  using P = typename Cfg::P;
  constexpr int S = 2, TAPS = 9, CK = 32, KS = CK / 16, HW_ = Cfg::HW, HH_ = Cfg::HH, FT = Cfg::FT, NWP = Cfg::NWP;
  constexpr int NPH = CTX ? 3 : 1;                         // phases per channel chunk
  constexpr int BN = Cfg::BN, NTHR = Cfg::NTHR, AROWS = Cfg::AROWS, BUF = Cfg::BUF;
  static_assert(!RES || (CTX && FT == 1 && Cfg::RES_BYTES <= 160 * 1024), "resident layout: gated conv, one frame per tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RES ? Cfg::RES_BYTES : Cfg::LDS_BYTES];     // static: see conv_kernels.h

  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wp = wave % NWP, wc = wave / NWP;             // position group / channel group of this wave
  const int H = a.H, W = a.W, T = a.T, Cin = a.Cin, HWp = a.H * a.W;

  // ---- this workgroup's run of tiles.  Workgroup ids go round-robin over the 8 XCDs; XCD k owns the CONTIGUOUS tile
  // range [lo, hi) (channel block fastest, then x, y, frame, batch) and its workgroups stride through it together, so
  // the blocks that share an activation halo (other channel blocks, neighbouring tiles, the next two frames whose
  // context this frame is) run at the same time on the same L2.
  const int ntiles = d.ntx * d.nty * d.ntt * a.B * d.ncob;
  int tl, tl_hi, tl_step;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int lo = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    tl_hi = lo + q + ((xcd < rr) ? 1 : 0);
    tl_step = (nwg - xcd + 7) >> 3;
    tl = lo + (blockIdx.x >> 3);
  }
  if (tl >= tl_hi) return;
  struct Tile { int co0, x0, y0, t0, b; };
  auto decode = [&](int id) __attribute__((always_inline)) {
    Tile t;
    t.co0 = (id % d.ncob) * BN; id /= d.ncob;
    t.x0 = (id % d.ntx) * P::PW; id /= d.ntx;
    t.y0 = (id % d.nty) * P::PH; id /= d.nty;
    t.t0 = (id % d.ntt) * P::FT; id /= d.ntt;
    t.b = id;
    return t;
  };

  // lane -> position inside a 32-position tile (see conv_kernels.h: 16-lane read groups get 16 consecutive halo rows)
  int pr = r;
  if constexpr (PW == 16) {
    const bool ga = (r < 4) || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int k = ga ? ((r < 4) ? r : (r < 16) ? r - 8 : r - 12) : ((r < 12) ? r - 4 : (r < 20) ? r - 8 : r - 16);
    pr = (ga ? 0 : 16) + k;
  } else {                                                // PW == 8: a read group takes patch rows (0,2) resp. (1,3)
    const bool ga = (r < 4) || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int k = ga ? ((r < 4) ? r : (r < 16) ? r - 8 : r - 12) : ((r < 12) ? r - 4 : (r < 20) ? r - 8 : r - 16);
    pr = ((k >> 3) * 2 + (ga ? 0 : 1)) * 8 + (k & 7);
  }
  int arow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = (wp + NWP * m) * 32 + pr;      // tile m of a wave sits NWP*32 positions (a multiple of 16 halo rows) on
    const int ft = p / (P::PH * P::PW), py = (p / P::PW) % P::PH, px = p % P::PW;
    arow[m] = (ft * HH_ + py) * HW_ + px;
  }

  // ---- DMA descriptors (one 16-byte piece per lane per instruction; piece e of a region lands at region + 16*e).
  // All per-lane byte offsets are phase-invariant (`buffer_load ... offen lds`: the phase / chunk / frame part of the
  // address is the uniform soffset), and out-of-image halo pixels use an offset beyond num_records: the buffer range
  // check then writes zeros, so spatial padding costs neither branches nor address math.
  constexpr int TOTA = AROWS * 4, TOTC = Cfg::CROWS * 4, TOTW = Cfg::WROWS * 4;
  constexpr int NIA = (TOTA + NTHR - 1) / NTHR, NIC = (TOTC + NTHR - 1) / NTHR, NIW = (TOTW + NTHR - 1) / NTHR;
  constexpr int OOB = (int)0x80000000;
  const int frame_elems = HWp * Cin;
  int adesc[NIA], wdesc[NIW];
  auto set_adesc = [&](const Tile& t) __attribute__((always_inline)) {
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));      // opaque: keeps hipcc from hoisting (and then spilling) the per-piece constants
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int e = i * NTHR + tid_;
      const int row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
      const int s = row / Cfg::SROWS, rem = row % Cfg::SROWS;
      const int f_ = rem / (HH_ * HW_), hr = rem % (HH_ * HW_);          // frame inside the tile, halo pixel
      const int y = t.y0 + hr / HW_ - 1, x = t.x0 + hr % HW_ - 1;
      adesc[i] = OOB;                                                    // (for FT > 1 the frame is kept in bits 28..30
      if (e < TOTA && rem < Cfg::HALO && hr % HW_ < P::PW + 2 && t.t0 + f_ < T && (unsigned)y < (unsigned)H &&
          (unsigned)x < (unsigned)W)                                     //  of nothing: it is re-derived where needed)
        adesc[i] = ((s * T + f_) * frame_elems + (y * W + x) * Cin + gp * 8) * 2;
    }
  };
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    const int e = (i * NW + wave) * 64 + lane;
    const int row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
    const int tap = row / BN, co = row % BN;
    wdesc[i] = ((tap * a.CoutP + co) * a.CinP + gp * 8) * 2;
  }
  const int wbytes = TAPS * a.CoutP * a.CinP * 2;
  const i32x4 rs_f = make_rsrc(oniris_fill_rows, 128);
  const i32x4 rs_wo = make_rsrc(a.w_own, wbytes), rs_wc = make_rsrc(CTX ? a.w_ctx : a.w_own, CTX ? 2 * wbytes : wbytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  const int fillsel = (a.ctx_fill != 0.f) ? 64 : 0;

  auto issue = [&](const Tile& t, int ch, int ph, int bsel) __attribute__((always_inline)) {
    const int c0 = ch * CK;
    // two alternating buffers, or (RES) the phase's own region: own at 0, ctx0 / ctx1 behind it (weights follow the
    // SROWS context rows there)
    const int boff = RES ? ((ph == 0) ? 0 : BUF + (ph - 1) * Cfg::BUFC) : bsel * BUF;
    const int woff = (RES && ph != 0) ? Cfg::SROWS * 64 : AROWS * 64;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + boff + wave * 1024);   // + i * NW * 1024 per piece
    if (ph == 0) {
      const bf16* xb = (const bf16*)a.x + (size_t)t.b * S * T * frame_elems;
      const i32x4 rs_x = make_rsrc(xb, S * T * frame_elems * 2);
      const int so = (t.t0 * frame_elems + c0) * 2;
#pragma unroll
      for (int i = 0; i < NIA; ++i)
        if ((i * NW + wave) * 64 + lane < TOTA) dma16(rs_x, adesc[i], so, dst + i * NW * 1024);
    } else if constexpr (FT == 1) {
      const int f = t.t0 + ((ph == 1) ? a.coff0 : a.coff1);
      if (f >= 0 && f < a.ctx_T) {
        const bf16* cb_ = (const bf16*)a.ctx + (size_t)t.b * a.ctx_bstride * frame_elems;
        const i32x4 rs_c = make_rsrc(cb_, a.ctx_T * frame_elems * 2);
        const int so = (f * frame_elems + c0) * 2;
#pragma unroll
        for (int i = 0; i < NIC; ++i)                        // rows < HALO: the slot-0 rows of the halo image
          if ((i * NW + wave) * 64 + lane < TOTC) dma16(rs_c, adesc[i], so, dst + i * NW * 1024);
      } else {                                               // padded frame: constant row (spatial padding stays 0)
#pragma unroll
        for (int i = 0; i < NIC; ++i)
          if ((i * NW + wave) * 64 + lane < TOTC) dma16(rs_f, (adesc[i] < 0) ? OOB : fillsel, 0, dst + i * NW * 1024);
      }
    } else {                                                 // several frames per tile: real / padded is per row
      const int coff = (ph == 1) ? a.coff0 : a.coff1;
      const bf16* cb_ = (const bf16*)a.ctx + (size_t)t.b * a.ctx_bstride * frame_elems;
      const i32x4 rs_c = make_rsrc(cb_, a.ctx_T * frame_elems * 2);
      const int shift = (t.t0 + coff) * frame_elems * 2;     // may be negative; voffset + shift >= 0 on the real rows
#pragma unroll
      for (int i = 0; i < NIC; ++i) {
        const int e = (i * NW + wave) * 64 + lane;
        if (e < TOTC) {
          const int f = t.t0 + coff + ((e >> 2) % Cfg::SROWS) / (HH_ * HW_);
          if (f >= 0 && f < a.ctx_T) dma16(rs_c, (adesc[i] < 0) ? OOB : adesc[i] + shift, c0 * 2, dst + i * NW * 1024);
          else dma16(rs_f, (adesc[i] < 0) ? OOB : fillsel, 0, dst + i * NW * 1024);
        }
      }
    }
    const int sw = ((t.co0 * a.CinP + c0) + ((ph == 2) ? TAPS * a.CoutP * a.CinP : 0)) * 2;
#ifdef CONV_STAMP
    if (a.big_tile & 64) return;          // diagnostic: no weight copies at all (results are garbage): what would resident weights buy?
#endif
    if (ph == 0) {
#pragma unroll
      for (int i = 0; i < NIW; ++i)
        if ((i * NW + wave) * 64 + lane < TOTW) dma16(rs_wo, wdesc[i], sw, dst + woff + i * NW * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < NIW; ++i)
        if ((i * NW + wave) * 64 + lane < TOTW) dma16(rs_wc, wdesc[i], sw, dst + woff + i * NW * 1024);
    }
  };

  // ---- fragment addresses (k-step 0; k-step 1 is the same address ^ 32)
  const int wa0 = (wc * NT * 32 + r) * 64 + ((h ^ ((r >> 2) & 3)) << 4);          // + offset of the weight rows in the buffer
  // slot 1 and the wave's further position tiles are whole multiples of 16 rows away: same swizzle, constant offset
  int xa0[TAPS];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int R = arow[0] + (tap / 3) * HW_ + (tap % 3);
    xa0[tap] = R * 64 + ((h ^ ((R >> 2) & 3)) << 4);
  }

  f32x16 acc[S][MT][NT];
  f32x16 accc[MT][NT];

  // `mid`: called once, in front of step MID_STEP (wave-uniform; the younger half of the workgroup issues its share of
  // the next phase's DMA there, see the phase loop)
  constexpr int MID_STEP = 4;
  auto mfma_steps = [&](auto own_, int boff, int woff, auto mid) __attribute__((always_inline)) {
    constexpr bool OWN = decltype(own_)::value;
    constexpr int NX = OWN ? S : 1;
    constexpr int NSTEP = TAPS * KS;
    const unsigned char* base = smem + boff;
    const unsigned char* wbase = base + woff;
    bf16x8 wf[2][NT], xf[2][NX][MT];
    auto ld = [&](int fb, int st) __attribute__((always_inline)) {
      const int tap = st / KS, ks = st % KS;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        wf[fb][n] = *(const bf16x8*)(wbase + ((wa0 ^ (ks * 32)) + (tap * BN + n * 32) * 64));
#pragma unroll
      for (int s = 0; s < NX; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          xf[fb][s][m] = *(const bf16x8*)(base + ((xa0[tap] ^ (ks * 32)) + (s * Cfg::SROWS + m * (NWP * 2 * HW_)) * 64));
    };
    ld(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      mid(st);
      if (st + 1 < NSTEP) ld((st + 1) & 1, st + 1);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if constexpr (OWN) {
#pragma unroll
            for (int s = 0; s < S; ++s) acc[s][m][n] = mfma32(wf[st & 1][n], xf[st & 1][s][m], acc[s][m][n]);
          } else {
            accc[m][n] = mfma32(wf[st & 1][n], xf[st & 1][0][m], accc[m][n]);
          }
        }
      // pin the pipeline: the fragment reads of step st+1 go out BEFORE the MFMAs of step st (T19)
      if (st + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NT + NX * MT, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NX * MT * NT, 0);
    }
  };
  auto no_mid = [](int) __attribute__((always_inline)) {};

  constexpr int EROW = Cfg::EROW;
  constexpr int ESC = 2 * FT * BN * 4;                  // emb-scale vectors [slot][frame][BN], in front of the staging tiles
  const int nphase = (Cin / CK) * NPH;
  Tile cur = decode(tl);
  set_adesc(cur);
  int bsel = 0;
  issue(cur, 0, 0, 0);
  if constexpr (RES) { issue(cur, 0, 1, 0); issue(cur, 0, 2, 0); }
  Tile nxt = cur;
  bool more = false;
  int tl_next = tl;
  CSTAMP_DECL
#pragma unroll 1
  for (;;) {
    dma_wait();
    __syncthreads();     // phase 0 of `cur` has landed for everybody; the previous epilogue is over
    CSTAMP(0)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][m][n][i] = 0.f; acc[1][m][n][i] = 0.f; accc[m][n][i] = 0.f; }
      }
    float esc_v = 0.f;
    // gate coefficients of this lane's frame: fetched now (nothing in flight) so that no wait on them can fall behind
    // a later LDS-DMA
    const int lft = ((wp * 32 + pr) / (P::PH * P::PW)) % FT;     // frame of this lane's position inside the tile (MT == 1 when FT > 1)
    const bool lvalid = cur.t0 + lft < T;
    const int n0 = (cur.b * S) * T + (lvalid ? cur.t0 + lft : 0), n1 = n0 + T;       // slot 0 / slot 1
    const float cown0 = a.coef_own ? a.coef_own[n0] : 1.f, cown1 = a.coef_own ? a.coef_own[n1] : 1.f;
    const float cctx0 = a.coef_ctx ? a.coef_ctx[n0] : 1.f, cctx1 = a.coef_ctx ? a.coef_ctx[n1] : 1.f;
    if (a.epi == ONIRIS_EPI_EMB_SILU && tid < 2 * FT * BN) {     // emb-scale element [slot][frame][co] of this thread (parked in LDS later)
      const int co = cur.co0 + tid % BN, f_ = (tid / BN) % FT, s_ = tid / (FT * BN);
      if (co < a.Cout && cur.t0 + f_ < T)
        esc_v = ((const float*)a.escale)[(size_t)((cur.b * S + s_) * T + cur.t0 + f_) * (a.escale_pitch ? a.escale_pitch : a.Cout) + co];
    }
    CSTAMP(1)
    if constexpr (!RES) {
#pragma unroll 1
      for (int itp = 0; itp < nphase; ++itp) {
        const int ph = itp % NPH;
        // The copy of the next phase (buffer bsel^1: last read in phase itp-1) is issued by the older half of the waves
        // BEFORE their MFMAs and by the younger half a few steps INTO theirs.  A buffer_load..lds costs its wave ~100
        // cycles of issue; with all eight waves issuing at the phase top (they leave the barrier together) the matrix
        // pipe of every SIMD sat idle for the ~1000 cycles its two waves spent on their ~10 pieces each (stamped: 30 %
        // of the kernel).  Now one wave of a SIMD feeds the matrix pipe while the other one issues.
        const bool have_next = itp + 1 < nphase;
        // (Which half goes first, and `s_setprio` on either half for part or all of a phase, measured nothing: round 4.)
        const bool early = __builtin_amdgcn_readfirstlane(wave) < NW / 2 || (a.big_tile & 32);      // (bit 5: A/B knob, all waves issue at the phase top)
        if (have_next && early) issue(cur, (itp + 1) / NPH, (itp + 1) % NPH, bsel ^ 1);
        auto mid = [&](int st) __attribute__((always_inline)) {
          if (st == MID_STEP && have_next && !early) issue(cur, (itp + 1) / NPH, (itp + 1) % NPH, bsel ^ 1);
        };
        CSTAMP2(ph == 0, 2, 7)
        if (!CTX || ph == 0) mfma_steps(std::true_type{}, bsel * BUF, AROWS * 64, mid);
        else mfma_steps(std::false_type{}, bsel * BUF, AROWS * 64, mid);
#ifdef CONV_STAMP
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][0][0]), "v"(accc[0][0]), "v"(acc[0][0][NT - 1]), "v"(acc[1][0][NT - 1]), "v"(accc[0][NT - 1]));
#endif
        CSTAMP2(ph == 0, 3, 8)
        dma_wait();                        // this wave's share of the next phase has landed ...
        CSTAMP2(ph == 0, 4, 9)
        __syncthreads();                   // ... everybody's has; and everybody is done reading buffer bsel
        CSTAMP2(ph == 0, 5, 10)
        bsel ^= 1;
      }
    }
    // Both buffers are free now.  The epilogue stages through buffer bsel^1 (the one just consumed); the first DMA of
    // the next tile goes to buffer bsel, continuing the alternation.  (RES: a dedicated staging area.)
    unsigned char* stg = smem + (RES ? Cfg::RES_EOFF : (bsel ^ 1) * BUF);
    unsigned char* ep = stg + ESC + wave * 32 * EROW;
    if constexpr (RES) mfma_steps(std::true_type{}, 0, AROWS * 64, no_mid);       // own phase (all three regions landed at the tile top)

    // -- epilogue inputs must not wait behind the next LDS-DMA (vmcnt is in order, and hipcc waits vmcnt(0) at the
    // first use of an ordinary load issued while DMA is in flight): gate coefficients and emb-scale were fetched at
    // the top of the tile; the emb-scale vectors are parked in LDS now
    if (a.epi == ONIRIS_EPI_EMB_SILU) {
      if (tid < 2 * FT * BN) *(float*)(stg + tid * 4) = esc_v;
      __syncthreads();
    }
    asm volatile("" ::"v"(cown0), "v"(cown1), "v"(cctx0), "v"(cctx1));   // consume here: nothing is in flight at this point
    // -- first DMA of the next tile (runs under the epilogue below)
    tl_next = tl + tl_step;
    more = tl_next < tl_hi;
    nxt = cur;
    if constexpr (RES) {
      // Resident layout: a region is refilled for the NEXT tile as soon as its phase of this tile is done, so every
      // copy has the rest of the tile (one or two phases + the epilogue) to land, and the one wait at the tile top is
      // the only one.  (Ordinary loads were consumed above: from here on DMA is always in flight.)
      __syncthreads();                                   // everybody is done reading the own region
      if (more) { nxt = decode(tl_next); set_adesc(nxt); issue(nxt, 0, 0, 0); }
      mfma_steps(std::false_type{}, BUF, Cfg::SROWS * 64, no_mid);
      __syncthreads();
      if (more) issue(nxt, 0, 1, 0);
      mfma_steps(std::false_type{}, BUF + Cfg::BUFC, Cfg::SROWS * 64, no_mid);
      __syncthreads();
      if (more) issue(nxt, 0, 2, 0);
    } else if (more) {
      nxt = decode(tl_next);
      set_adesc(nxt);
      issue(nxt, 0, 0, bsel);
    }

    // -- epilogue: lane-local math (lane = position), bf16 results transposed through a wave-private LDS tile so that
    // global stores are 16 B per lane over whole channel rows
    bf16* og = (bf16*)a.out;
    bool clip_hit = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ptile = (wp + NWP * m) * 32;              // first position of this tile inside the workgroup tile
      const int cow = cur.co0 + wc * NT * 32;              // first channel of this wave
      auto flush = [&](bf16* dst, size_t blk) __attribute__((always_inline)) {
        constexpr int PO = NT * 32 / 8;
#pragma unroll
        for (int it = 0; it < 32 * PO / 64; ++it) {
          const int id = it * 64 + lane;
          const int row = id / PO, part = id % PO;
          const int q = ptile + row;
          const int f_ = q / (P::PH * P::PW), yy = (q / P::PW) % P::PH, xx = q % P::PW;
          const size_t px_ = (size_t)(cur.t0 + f_) * HWp + (cur.y0 + yy) * W + (cur.x0 + xx);
          const int co = cow + part * 8;
          if (co < a.Cout && cur.t0 + f_ < T) {
            const u32x4 v_ = *(const u32x4*)(ep + row * EROW + part * 16);
            u32x4* o_ = (u32x4*)(dst + (blk + px_) * a.Cout + co);
            if (d.nt) __builtin_nontemporal_store(v_, o_); else *o_ = v_;
          }
        }
      };
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float cown = s ? cown1 : cown0, cctx = s ? cctx1 : cctx0;
        const size_t blk = (size_t)(cur.b * S + s) * T * HWp;
        // one channel tile (16 values per lane) at a time keeps the epilogue's register footprint small; the gated
        // sum is simply recomputed from the accumulators for every output that needs it
        auto raw = [&](int nt, float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = CTX ? __builtin_fmaf(cctx, accc[m][nt][i], cown * acc[s][m][nt][i]) : cown * acc[s][m][nt][i];
        };
        auto put = [&](int nt, const float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = f2bf(v[4 * g + k]);
            *(bf16x4*)(ep + pr * EROW + (nt * 32 + 8 * g + 4 * h) * 2) = o;
          }
        };
        float v[16];
        if (a.epi == ONIRIS_EPI_MPSUM) {     // (the res loads queue behind the DMA just issued: vmcnt is in order)
          if (a.out2) {                      // raw conv output (needed for d gate)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { raw(nt, v); put(nt, v); }
            flush((bf16*)a.out2, blk);
          }
          const int p = ptile + pr;
          const size_t obase = (blk + (size_t)(cur.t0 + lft) * HWp + (cur.y0 + (p / P::PW) % P::PH) * W + cur.x0 + p % P::PW) * a.Cout;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            raw(nt, v);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int co = cow + nt * 32 + 8 * g + 4 * h;
              bf16x4 rv;
#pragma unroll
              for (int k = 0; k < 4; ++k) rv[k] = f2bf(0.f);
              if (co < a.Cout && lvalid) rv = *(const bf16x4*)((const bf16*)a.res + obase + co);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                float o = a.ta * bf2f(rv[k]) + a.tb * v[4 * g + k];
                if (a.clip > 0.f) {
                  o = fminf(fmaxf(o, -a.clip), a.clip);
                  clip_hit |= !(fabsf(bf2f(f2bf(o))) < a.clip);     // (what the backward's mask tests: the STORED value)
                }
                v[4 * g + k] = o;
              }
            }
            put(nt, v);
          }
          flush(og, blk);
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { raw(nt, v); put(nt, v); }
          flush(og, blk);
          if (a.epi == ONIRIS_EPI_EMB_SILU) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              raw(nt, v);
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 ev = *(const float4*)(stg + ((s * FT + lft) * BN + wc * NT * 32 + nt * 32 + 8 * g + 4 * h) * 4);
                const float cvv[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const float z = bf2f(f2bf(v[4 * g + k])) * cvv[k];     // the activation sees the bf16-rounded y
                  v[4 * g + k] = z * sigmoid_fast(z) * (1.f / 0.596f);
                }
              }
              put(nt, v);
            }
            flush((bf16*)a.out2, blk);
          }
        }
        if (CTX && a.ctx_out && s == 0) {   // unscaled context product y3 (shared by both slots), kept for d(gate)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = accc[m][nt][i];
            put(nt, v);
          }
          flush((bf16*)a.ctx_out, (size_t)cur.b * T * HWp);
        }
      }
    }
    if (a.clip_flag && __builtin_amdgcn_ballot_w64(clip_hit) != 0ull) {      // (practically never: OnirisConvArgs.clip_flag)
      if (lane == 0) atomicOr(a.clip_flag, 1);
    }
    CSTAMP(6)
    if (!more) break;
    cur = nxt;
    tl = tl_next;
  }
#ifdef CONV_STAMP
  if (blockIdx.x == 0 && lane == 0 && a.splitk_ws) {
    unsigned long long* dst = (unsigned long long*)a.splitk_ws + wave * 12;
    for (int i = 0; i < 12; ++i) dst[i] = st_acc[i];
  }
#endif
#endif
}

template <int NT, int PW, int NW, int MT, int WC = 1, bool CTX = true, bool RES = false>
static int launch_conv_glds(const OnirisConvArgs& a, hipStream_t stream) {
  using Cfg = GldsCfg<NT, PW, NW, MT, WC>;
  using P = typename Cfg::P;
  ConvDev d;
  d.a = a;
  d.ksplit = 1; d.reduce = 0;
  d.nt = (long long)a.B * a.S * a.T * a.H * a.W * a.Cout * 2 >= oniris_ew_nt_bytes();
  d.ncob = a.CoutP / Cfg::BN;
  d.ntx = a.W / P::PW; d.nty = a.H / P::PH; d.ntt = cdiv(a.T, P::FT);
  const long long ntiles = (long long)d.ntx * d.nty * d.ntt * a.B * d.ncob;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) { oniris_set_error("conv_fwd: bad grid %lld", ntiles); return ONIRIS_EINVAL; }
  const int ncu = oniris_persistent_wgs();       // one workgroup per CU (minus the CUs reserved for a gradient exchange in flight)
  const long long nblk = ntiles < ncu ? ntiles : ncu;
  auto kern = conv_glds_kernel<NT, PW, NW, MT, WC, CTX, RES>;
  oniris_launch_tagged(d.nt ? "nt-stores" : nullptr, kern, dim3((unsigned)nblk), dim3(Cfg::NTHR), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

// true when the LDS-DMA variant can run this problem (plain: S == 1 without context, an even number of frames)
static inline bool conv_glds_ok(const OnirisConvArgs& a, int PH, int PW, int BN, bool plain = false) {
  return (plain ? (a.S == 1 && !a.ctx && a.T % 2 == 0) : (a.S == 2 && a.ctx != nullptr)) && a.taps == 9 && a.Cin % 32 == 0 && a.CoutP % BN == 0 && a.H % PH == 0 && a.W % PW == 0 &&
         (a.ctx_fill == 0.f || a.ctx_fill == 1.f) &&
         2LL * a.T * a.H * a.W * a.Cin * 2 < (1LL << 31) && (long long)a.ctx_T * a.H * a.W * a.Cin * 2 < (1LL << 31) &&
         18LL * a.CoutP * a.CinP * 2 < (1LL << 31);
}
