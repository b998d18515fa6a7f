// Gated causal 3x3 convolution, DART training layout, Cin = 32 -> Cout <= 32 (the 64x64-pixel level of the UNets and its
// dgrads): STREAMING variant of conv_glds_kernel<NT=1, RES>.
//
// At 32 channels a 16x16-pixel tile gives the matrix pipe 72 MFMAs per wave against 138 KB of LDS-DMA (halo images of
// both slots and both context frames + three weight slabs, all re-copied for every tile) and an epilogue that stops all
// eight waves: the launches sit at 0.21-0.24 of the bf16 peak and at half of their HBM roofline (the level is HBM-bound:
// 235 MB / 15 us of MFMA work per launch at B = 2).  Here
//   * a workgroup (4 waves, 4x16 pixels, TWO per CU so that one's epilogue runs under the other's MFMAs) owns one
//     (sequence, tile) and WALKS THE FRAMES of a segment: per frame it copies the halos of x[s0,t], x[s1,t] (2 x 6.9 KB;
//     dgrad: + dy3[t]); the context frames t-2, t-1 are the slot-0 halos of earlier steps, still in an LDS ring;
//   * the weights (27 taps x 32 x 32) live in REGISTERS for the whole walk, 18 fragments per wave: waves 0,1 hold the
//     own-frame slab and compute both slots of their 32 positions, wave 2 / 3 holds the slab of context frame t-2 / t-1 and
//     computes its product for both position tiles -- 36 MFMAs per wave and frame, one LDS fragment read per MFMA;
//   * the partial products meet through LDS (six arrays of 16 floats per lane), then wave w finishes slot w >> 1 of tile
//     w & 1 (waves 2,3 also the un-gated context product y3): same epilogue arithmetic as conv_glds.h.
// Bit-compatible INPUT rounding (bf16 operands, fp32 accumulation); the summation order over (tap, k) equals the tile
// kernel's per product, the own + context combination is the same fma.
#pragma once
#include "conv_kernels.h"
#include "lds_dma.h"

#ifdef CONV_STAMP
#define SSTAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SSTAMP(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; }
#else
#define SSTAMP_DECL
#define SSTAMP(i)
#endif

struct ConvStreamDev {
  OnirisConvArgs a;
  int ntx, nty, nseg, seglen;
  int dir;                 // +1: frames ascending (context = earlier frames: coff = -2, -1); -1: descending (coff = 2, 1)
  int nt;                  // outputs are streamed with non-temporal stores (tensors beyond oniris_ew_nt_bytes(): nothing re-reads them from a cache)
};

template <bool ALIAS>      // ALIAS: the context frames are the slot-0 frames of x itself (forward); else a separate tensor (dgrad)
__global__ __launch_bounds__(256, 2) void conv_stream_kernel(const ConvStreamDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TAPS = 9, KS = 2, NST = TAPS * KS, HW_ = 18, HALO = 6 * 18, HBUF = 7168;     // 108 rows x 64 B, padded to 112
  constexpr int NXS0 = ALIAS ? 4 : 2, NCTX = ALIAS ? 0 : 3;
  constexpr int EROW = 80, STGB = 32 * EROW;
  constexpr int XS0 = 0, XS1 = NXS0 * HBUF, CTXR = XS1 + 2 * HBUF, EXCH = CTXR + NCTX * HBUF;
  // six exchange arrays of 16 x 64 floats; the forward kernel has room for a separate staging tile per wave, the dgrad
  // kernel (one more halo ring) stages through the exchange arrays behind a third barrier
  constexpr int STG = ALIAS ? EXCH + 6 * 4096 : EXCH, ESC = EXCH + 6 * 4096 + (ALIAS ? 4 * STGB : 0);
  constexpr int LDS_BYTES = ESC + 512;                // emb-scale rows [frame parity][slot][32] fp32
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  // roles: wave 0 / 1 = own-frame product of position tile 0 / 1 (both slots); wave 2 / 3 = context product of frame
  // t + coff0 / t + coff1 (both position tiles).  Everybody: 18 weight fragments, 2 accumulators, 36 MFMAs per frame.
  // Epilogue: wave w finishes slot w >> 1 of position tile w & 1.
  const int uw = __builtin_amdgcn_readfirstlane(wave);     // (an SGPR: role branches are scalar branches)
  const bool ctxw = uw >= 2;
  const int sl = uw >> 1, pt = uw & 1;
  const int H = a.H, W = a.W, T = a.T, HWp = H * W, Cout = a.Cout;
  constexpr int Cin = 32;
  const int frame_elems = HWp * Cin;

  // ---- this workgroup's (sequence, tile, segment); workgroup ids go round-robin over the XCDs: XCD k takes a contiguous
  // range of units, so tiles that share halo rows sit behind the same L2
  int u;
  {
    const int n = gridDim.x, xcd = blockIdx.x & 7, q = n >> 3, rr = n & 7;
    u = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blockIdx.x >> 3);
  }
  const int seg = u % d.nseg; u /= d.nseg;
  const int x0 = (u % d.ntx) * 16; u /= d.ntx;
  const int y0 = (u % d.nty) * 4;
  const int b = u / d.nty;
  const int t_lo = seg * d.seglen, t_hi = min(T, t_lo + d.seglen), nfr = t_hi - t_lo;
  const int dir = d.dir, ts = (dir > 0) ? t_lo : t_hi - 1;

  // ---- lane -> position inside a 32-position tile (conv_kernels.h: 16-lane read groups take 16 consecutive halo rows)
  int pr;
  {
    const bool ga = (r < 4) || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int k = ga ? ((r < 4) ? r : (r < 16) ? r - 8 : r - 12) : ((r < 12) ? r - 4 : (r < 20) ? r - 8 : r - 16);
    pr = (ga ? 0 : 16) + k;
  }
  // fragment addresses of the two MFMA streams of this wave: own waves read the SAME tile from two buffers (slot 0 / 1),
  // context waves read BOTH tiles from one buffer
  int xaddrA[TAPS], xaddrB[TAPS];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int tA = ctxw ? 0 : pt, tB = ctxw ? 1 : pt;
    const int RA = (2 * tA + (pr >> 4) + tap / 3) * HW_ + (pr & 15) + tap % 3;
    const int RB = (2 * tB + (pr >> 4) + tap / 3) * HW_ + (pr & 15) + tap % 3;
    xaddrA[tap] = RA * 64 + ((h ^ ((RA >> 2) & 3)) << 4);            // k-step 1: ^ 32
    xaddrB[tap] = RB * 64 + ((h ^ ((RB >> 2) & 3)) << 4);
  }

  // ---- DMA descriptors: 432 pieces of 16 B per halo image
  constexpr int OOB = (int)0x80000000;
  int hv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = i * 256 + tid, row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
    const int y = y0 + row / HW_ - 1, x = x0 + row % HW_ - 1;
    hv[i] = (e < HALO * 4 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? ((y * W + x) * Cin + gp * 8) * 2 : OOB;
  }
  const i32x4 rs_x = make_rsrc((const bf16*)a.x + (size_t)b * 2 * T * frame_elems, 2 * T * frame_elems * 2);
  const i32x4 rs_c = make_rsrc((const bf16*)a.ctx + (size_t)b * a.ctx_bstride * frame_elems, a.ctx_T * frame_elems * 2);
  const i32x4 rs_f = make_rsrc(oniris_fill_rows, 128);
  const int fillsel = (a.ctx_fill != 0.f) ? 64 : 0;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem, wdst = lds0 + wave * 1024;
  auto halo = [&](const i32x4& rs, int so, unsigned dst) __attribute__((always_inline)) {
    dma16(rs, hv[0], so, dst + wdst);
    if (256 + tid < HALO * 4) dma16(rs, hv[1], so, dst + wdst + 4096);
  };
  auto halo_fill = [&](unsigned dst) __attribute__((always_inline)) {
    dma16(rs_f, (hv[0] < 0) ? OOB : fillsel, 0, dst + wdst);
    if (256 + tid < HALO * 4) dma16(rs_f, (hv[1] < 0) ? OOB : fillsel, 0, dst + wdst + 4096);
  };
  auto ctx_slot = [&](int f) __attribute__((always_inline)) {        // LDS offset of context frame f
    return ALIAS ? XS0 + (f & 3) * HBUF : CTXR + ((f + 3) % 3) * HBUF;
  };
  auto issue_ctx = [&](int f) __attribute__((always_inline)) {       // context frame f (padding outside [0, ctx_T))
    if (f >= 0 && f < a.ctx_T) halo(rs_c, f * frame_elems * 2, ctx_slot(f));
    else halo_fill(ctx_slot(f));
  };
  auto issue_x = [&](int f) __attribute__((always_inline)) {         // both slots of frame f
    halo(rs_x, f * frame_elems * 2, XS0 + (ALIAS ? (f & 3) : (f & 1)) * HBUF);
    halo(rs_x, (T + f) * frame_elems * 2, XS1 + (f & 1) * HBUF);
  };

  // ---- weights -> registers (once): lane (r = co row, h = 8-channel group of the k-step)
  bf16x8 wreg[NST];
  {
    const bf16* wsrc = ctxw ? (const bf16*)a.w_ctx + (size_t)(uw - 2) * TAPS * a.CoutP * a.CinP : (const bf16*)a.w_own;
#pragma unroll
    for (int i = 0; i < NST; ++i)
      wreg[i] = *(const bf16x8*)(wsrc + ((size_t)(i / KS) * a.CoutP + r) * a.CinP + (i % KS) * 16 + h * 8);
  }
#pragma unroll
  for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(wreg[i]));    // consumed before any LDS-DMA is in flight

  // ---- per-frame epilogue inputs.  NOTHING in the frame loop is an ordinary vector load: the vmcnt stream of a wave
  // holds its LDS-DMA copies and its epilogue stores only, and the wait at the top of a step is COUNTED -- it lets the stores
  // of the previous epilogue drain while the next frame is worked on (vmcnt returns in issue order; with a full wait every
  // step paid the round trip of its own stores: the first version of this kernel ran at the tile kernel's speed).
  //   gate coefficients: wave-uniform -> scalar loads (lgkmcnt);  emb-scale rows: one LDS-DMA instruction of wave 0 per
  //   frame (2 x 128 B), with the halos;  residual: buffer loads through inline asm at the top of the step (hipcc cannot see
  //   them, so it inserts no wait of its own), waited for by count right before the epilogue uses them.
  const int ppy = 2 * pt + (pr >> 4), ppx = pr & 15;                 // pixel of this lane's OUTPUT position inside the 4x16 tile
  const int pix = (y0 + ppy) * W + x0 + ppx;
  const int uslot = sl;
  const bool emb = a.epi == ONIRIS_EPI_EMB_SILU, mps = a.epi == ONIRIS_EPI_MPSUM;
  const int epitch = a.escale_pitch ? a.escale_pitch : Cout;
  const i32x4 rs_e = make_rsrc(a.escale ? a.escale : (const void*)oniris_fill_rows, emb ? (int)(((size_t)a.B * 2 * T - 1) * epitch + Cout) * 4 : 0);
  const i32x4 rs_r = make_rsrc(mps ? (const bf16*)a.res + (size_t)b * 2 * T * HWp * Cout : (const bf16*)oniris_fill_rows,
                               mps ? 2 * T * HWp * Cout * 2 : 0);
  // piece (lane & 7) of slot (lane >> 3)'s row: the slot goes into the per-lane offset, the frame into the uniform one
  const int evoff = (lane < 16 && (lane & 7) * 4 < Cout) ? ((lane >> 3) * T * epitch) * 4 + (lane & 7) * 16 : OOB;
  auto issue_esc = [&](int f) __attribute__((always_inline)) {       // wave 0, lanes 0..15: [slot][32] floats of frame f
    if (lane < 16) dma16(rs_e, evoff, ((b * 2 * T + f) * epitch) * 4, lds0 + ESC + (f & 1) * 256);
  };
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 resq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { resq[g][0] = 0u; resq[g][1] = 0u; }
  int rvoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rvoff[g] = (8 * g + 4 * h < Cout) ? (pix * Cout + 8 * g + 4 * h) * 2 : OOB;
  auto load_res = [&](int f) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(((uslot * T + f) * HWp * Cout) * 2);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(resq[g]) : "v"(rvoff[g]), "s"(rs_r), "s"(so) : "memory");
  };
  auto wait_vm = [&](int n) __attribute__((always_inline)) {         // s_waitcnt vmcnt(n), n wave-uniform, 0..9
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
  };
  // instruction counts of this wave (wave-uniform): copies per frame, stores per epilogue
  const int per_halo = (uw < 3) ? 2 : 1;                             // the second piece of a halo image ends inside wave 2
  const int n_dma = (ALIAS ? 2 : 3) * per_halo + ((emb && uw == 0) ? 1 : 0);
  const int n_st = (a.ctx_out ? 1 : 0) + 2 * (mps ? ((a.out2 ? 1 : 0) + 1) : emb ? 2 : 1);
  if (nfr <= 0) return;
  // ---- prologue copies: the two context frames of the first step, then its own frames
  issue_ctx(ts + a.coff0);
  issue_ctx(ts + a.coff1);
  issue_x(ts);
  if (emb && uw == 0) issue_esc(ts);

  // exchange arrays: E0/E1 = slot 1's own product of tile 0/1 (from waves 0/1); E2/E3 = context frame 0's product of
  // tile 0/1 (from wave 2); E4/E5 = context frame 1's (from wave 3)
  float* const E = (float*)(smem + EXCH);
  unsigned char* const ep = smem + STG + wave * (ALIAS ? STGB : 4096);          // this wave's staging tile

  SSTAMP_DECL
#pragma unroll 1
  for (int i = 0; i < nfr; ++i) {
    const int t = ts + i * dir, tn = t + dir;
    SSTAMP(7)
    wait_vm(i ? n_st : 0);                 // this wave's copies of frame t have landed (its last epilogue's stores may still drain)
    SSTAMP(0)
    __syncthreads();                       // ... everybody's have; and everybody is done with the previous step
    SSTAMP(1)
    const bool has_next = i + 1 < nfr;
    const int nfrm = __builtin_amdgcn_readfirstlane((b * 2 + uslot) * T + t);
    // (read through the constant address space: hipcc then issues scalar loads -- lgkmcnt, not vmcnt -- for these
    // wave-uniform addresses; nothing in this kernel writes the coefficient vectors)
    typedef const __attribute__((address_space(4))) float cfloat_t;
    const float cown = a.coef_own ? ((cfloat_t*)(size_t)a.coef_own)[nfrm] : 1.f;
    const float cctx = a.coef_ctx ? ((cfloat_t*)(size_t)a.coef_ctx)[nfrm] : 1.f;
    if (mps) load_res(t);                  // (in front of the copies: waited for by count, see the epilogue)
    if (has_next) {
      issue_x(tn);
      if (!ALIAS) issue_ctx(tn + a.coff1);
      if (emb && uw == 0) issue_esc(tn);
    }

    SSTAMP(2)
    f32x16 acc0, acc1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
    {
      const unsigned char* bA = smem + (ctxw ? ctx_slot(t + (uw == 2 ? a.coff0 : a.coff1)) : XS0 + (ALIAS ? (t & 3) : (t & 1)) * HBUF);
      const unsigned char* bB = ctxw ? bA : smem + XS1 + (t & 1) * HBUF;
      bf16x8 xf[2][2];
      xf[0][0] = *(const bf16x8*)(bA + xaddrA[0]);
      xf[0][1] = *(const bf16x8*)(bB + xaddrB[0]);
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        if (st + 1 < NST) {
          const int tap = (st + 1) / KS, ks = (st + 1) % KS;
          xf[(st + 1) & 1][0] = *(const bf16x8*)(bA + (xaddrA[tap] ^ (ks * 32)));
          xf[(st + 1) & 1][1] = *(const bf16x8*)(bB + (xaddrB[tap] ^ (ks * 32)));
        }
        acc0 = mfma32(wreg[st], xf[st & 1][0], acc0);
        acc1 = mfma32(wreg[st], xf[st & 1][1], acc1);
        if (st + 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
    }
#ifdef CONV_STAMP
    asm volatile("" ::"v"(acc0), "v"(acc1));
#endif
    SSTAMP(3)
    // ---- the partial products meet
    if (!ctxw) {
#pragma unroll
      for (int k = 0; k < 16; ++k) E[(pt * 16 + k) * 64 + lane] = acc1[k];
    } else {
      const int e0 = (uw == 2) ? 2 : 4;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        E[((e0 + 0) * 16 + k) * 64 + lane] = acc0[k];
        E[((e0 + 1) * 16 + k) * 64 + lane] = acc1[k];
      }
    }
    SSTAMP(4)
    __syncthreads();
    SSTAMP(5)
    // v = cctx * (context frame 0's + context frame 1's product) + cown * (own product of slot sl), tile pt
    float v[16], cx[16];
    if (!ctxw) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        cx[k] = E[((2 + pt) * 16 + k) * 64 + lane] + E[((4 + pt) * 16 + k) * 64 + lane];
        v[k] = __builtin_fmaf(cctx, cx[k], cown * acc0[k]);
      }
    } else {
      const float* Eo = E + ((((uw == 2) ? 4 : 2) + pt) * 16) * 64 + lane;       // the other context frame's product, tile pt
      const float* Ew = E + (pt * 16) * 64 + lane;                                // slot 1's own product, tile pt
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float mine = pt ? acc1[k] : acc0[k], other = Eo[k * 64];
        cx[k] = (uw == 2) ? mine + other : other + mine;                          // (frame 0's + frame 1's, in that order)
        v[k] = __builtin_fmaf(cctx, cx[k], cown * Ew[k * 64]);
      }
    }
    if (!ALIAS) __syncthreads();           // dgrad: the staging tiles ARE the exchange arrays
    // ---- epilogue of slot `sl`, tile `pt` (lane = position; bf16 results transposed through the wave's LDS tile)
    const size_t blk = ((size_t)(b * 2 + sl) * T + t) * HWp;
    auto put = [&](const float (&vv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f2bf(vv[4 * g + k]);
        *(bf16x4*)(ep + pr * EROW + (8 * g + 4 * h) * 2) = o;
      }
    };
    auto flush_row = [&](bf16* dst, size_t fblk, int it) __attribute__((always_inline)) {      // pixel row `it` of the tile
      const int id = it * 64 + lane, row = id >> 2, part = id & 3;
      const size_t px_ = (size_t)(y0 + 2 * pt + (row >> 4)) * W + x0 + (row & 15);
      if (part * 8 < Cout) {
        const u32x4 v_ = *(const u32x4*)(ep + row * EROW + part * 16);
        u32x4* o_ = (u32x4*)(dst + (fblk + px_) * Cout + part * 8);
        if (d.nt) __builtin_nontemporal_store(v_, o_); else *o_ = v_;
      }
    };
    auto flush = [&](bf16* dst, size_t fblk) __attribute__((always_inline)) {
      flush_row(dst, fblk, 0);
      flush_row(dst, fblk, 1);
    };
    // the un-gated context product y3 (shared by both slots, kept for d(gate)): every wave has it for its tile -- the wave
    // that finishes slot 0 stores the tile's first pixel row, the one that finishes slot 1 the second (the stores are what
    // an epilogue waits for: three tensors on one pair of waves and two on the other cost 10 % of the launch)
    auto store_y3 = [&]() __attribute__((always_inline)) {
      put(cx);
      flush_row((bf16*)a.ctx_out, ((size_t)b * T + t) * HWp, sl);
    };
    if (mps) {
      // the residual loads of this step are older than its copies: n_dma younger instructions may still be in flight
      wait_vm(has_next ? n_dma : 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(resq[g]));
      float o[16];
      bool clip_hit = false;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x4 rv = __builtin_bit_cast(bf16x4, resq[g]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float q = a.ta * bf2f(rv[k]) + a.tb * v[4 * g + k];
          if (a.clip > 0.f) {
            q = fminf(fmaxf(q, -a.clip), a.clip);
            clip_hit |= !(fabsf(bf2f(f2bf(q))) < a.clip);         // (what the backward's mask tests: the STORED value)
          }
          o[4 * g + k] = q;
        }
      }
      if (a.ctx_out) store_y3();
      if (a.out2) { put(v); flush((bf16*)a.out2, blk); }
      put(o);
      flush((bf16*)a.out, blk);
      if (a.clip_flag && __builtin_amdgcn_ballot_w64(clip_hit) != 0ull) {    // (practically never: OnirisConvArgs.clip_flag)
        if (lane == 0) atomicOr(a.clip_flag, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // one more op in the vmcnt stream than the counted waits know
      }
      SSTAMP(6)
      continue;
    }
    if (a.ctx_out) store_y3();
    {
      put(v);
      flush((bf16*)a.out, blk);
      if (a.epi == ONIRIS_EPI_EMB_SILU) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 ev = *(const float4*)(smem + ESC + (t & 1) * 256 + (sl * 32 + 8 * g + 4 * h) * 4);
          const float cvv[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float z = bf2f(f2bf(v[4 * g + k])) * cvv[k];     // the activation sees the bf16-rounded y
            v[4 * g + k] = z * sigmoid_fast(z) * (1.f / 0.596f);
          }
        }
        put(v);
        flush((bf16*)a.out2, blk);
      }
    }
    SSTAMP(6)
  }
#ifdef CONV_STAMP
  if (blockIdx.x == 0 && lane == 0 && a.splitk_ws) {
    unsigned long long* dst = (unsigned long long*)a.splitk_ws + wave * 8;
    for (int i = 0; i < 8; ++i) dst[i] = st_acc[i];
  }
#endif
#endif
}

// Shapes the streaming kernel takes: DART training layout with the context path, 9 taps, exactly 32 input channels, at most
// 32 output channels, 16-pixel-wide tiles of 4 rows, context offsets (-2, -1) [forward: walk up] or (2, 1) [dgrad: walk down].
static inline bool conv_stream_ok(const OnirisConvArgs& a) {
  if (!(a.S == 2 && a.ctx && a.taps == 9 && a.Cin == 32 && a.CinP == 64 && a.CoutP == 32 && a.W % 16 == 0 && a.H % 4 == 0))
    return false;
  if (!(a.ctx_fill == 0.f || a.ctx_fill == 1.f) || a.ctx_T != a.T) return false;
  if (!((a.coff0 == -2 && a.coff1 == -1) || (a.coff0 == 2 && a.coff1 == 1))) return false;
  if (2LL * a.T * a.H * a.W * 32 * 2 >= (1LL << 31)) return false;
  return true;
}

static int launch_conv_stream(const OnirisConvArgs& a, hipStream_t stream) {
  ConvStreamDev d;
  d.a = a;
  d.ntx = a.W / 16; d.nty = a.H / 4;
  d.dir = (a.coff0 < 0) ? 1 : -1;
  const int units = a.B * d.ntx * d.nty;
  int nseg = 512 / units;                  // two workgroups per CU; a segment re-copies two context halos at its head
  if (nseg > a.T / 8) nseg = a.T / 8;
  if (nseg < 1) nseg = 1;
  d.nt = 2LL * a.B * a.T * a.H * a.W * a.Cout * 2 >= oniris_ew_nt_bytes();
  d.seglen = cdiv(a.T, nseg);
  d.nseg = cdiv(a.T, d.seglen);
  const bool alias = a.ctx == a.x && a.ctx_bstride == 2 * a.T && d.dir == 1;
  if (alias) oniris_launch_tagged(d.nt ? "nt-stores" : nullptr, conv_stream_kernel<true>, dim3(units * d.nseg), dim3(256), stream, d);
  else oniris_launch_tagged(d.nt ? "nt-stores" : nullptr, conv_stream_kernel<false>, dim3(units * d.nseg), dim3(256), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
