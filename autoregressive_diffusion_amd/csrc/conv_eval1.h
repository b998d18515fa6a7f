// Gated causal 3x3 convolution of ONE generated frame per sequence (the KV-cached sampler, edm2/sampler.py:12-85: 31 UNet
// evaluations per frame, each a chain of 58 of these) -- weight-streaming kernel for gfx950.
//
// A one-frame conv has 64 .. 4096 output positions and K = 27 * Cin: a handful of tiles, each with a long reduction whose
// weights (up to 3.5 MB per layer, 92 MB per evaluation) must come from L2 / HBM.  conv_fwd_kernel serves it as a split-K
// pair of launches (partials through a workspace, ~14 + ~14 us, bound by the serial latency of its register-staged phases
// and by the second launch).  Here ONE launch does it: a workgroup owns an 8x8-pixel x 32-channel output tile and STREAMS
// its whole K through a four-slot LDS ring:
//   * waves 4..7 = loaders: per phase (32 input channels x {own frame | cached frame 0 | cached frame 1}) one 10x12 halo
//     image (7.5 KB) and the 9-tap weight slab of the tile's 32 output channels (18 KB) by LDS-DMA, up to three phases ahead
//     of the consumers (counted vmcnt, one barrier per phase; phase 0 goes out alone so that the first MFMA starts as soon as
//     it has landed) -- the reduction runs at the CU's L2 -> LDS copy rate instead of at one round trip per phase;
//   * waves 0..3 = compute: they split the 18 (tap, k-step) products of a phase (wave w: k-step w & 1 of taps (w >> 1) + 2 t),
//     both 32-position halves of the tile each: D[co][position] += W[tap][co][ci] . X[position + tap][ci], MFMA 32x32x16,
//     every fragment read of the phase in front of its MFMAs; own and context products keep separate accumulators (the gate
//     mixes them in the epilogue);
//   * the four partial tiles meet in LDS (the ring is free then) and ALL eight waves run the epilogue on fp32 values:
//     gate combine, emb-scale + SiLU or mp_sum + clip, 8-byte coalesced bf16 stores.
// LDS images as in conv_glds.h: 64-byte rows (32 channels), the four 16-byte parts XOR-swizzled with row bits 2..3 on the
// source side, 12-entry halo rows (2 unused) so that the 16-lane groups of a ds_read_b128 take 16 consecutive rows.
// Context product kept between evaluations (OnirisConvArgs.ctx_prod, ABI 11): the 31 evaluations of one generated frame
// convolve the SAME cached pair with the same context weights -- only the gate coefficient in front of that product
// changes.  Mode 3 computes the product once per frame (context phases only, fp32 store), mode 2 reads it back and walks the
// own phases only: a third of the phases and of the weight stream per evaluation.  The stored value is the fp32 sum the
// epilogue would have formed itself, so mode 2 is bit-identical to the all-phases launch.
// Requirements (conv_eval1_ok): S == 1, T == 1, context = the cached pair (ctx_T == 2, coff = 0 / 1), taps == 9,
// Cin % 32 == 0, H % 8 == 0, W % 8 == 0.  Same results as the split-K path up to fp32 summation order.
#pragma once
#include "conv_kernels.h"
#include "lds_dma.h"

// CO = output channels per workgroup: 32, or 16 for the launches of a handful of workgroups (round 6).  A loader
// wave's LDS-DMA instruction moves 1 KB and the four of them get one out per ~50 cycles (the CU's address path: 20 B / clock, the
// per-CU streaming rate of MI355X_MICROARCH.md), so a phase of 8 halo + 18 weight instructions is 1.3-1.5 K cycles however the MFMAs
// are arranged (scratch/r06_eval1_stamp.py); with 16 channels it is 8 + 9 on twice the CUs.  The MFMA keeps its 32 rows: lanes 16..31
// read the weight rows of lanes 0..15 again and their outputs are dropped (the matrix pipe is idle most of the time here anyway).
template <int CO>
struct Eval1Cfg {
  static constexpr int HW = 12, HH = 10, HALO = HH * HW;           // halo rows of an 8x8 tile (12-wide: see conv_glds.h)
  static constexpr int WROWS = 9 * CO;
  static constexpr int SLOTB = (HALO + WROWS) * 64;               // 26112 B / 16896 B
  static constexpr int NSLOT = 4;
  static constexpr int NIA = 2;                                    // halo DMA instructions per loader wave and phase (4 waves x 64 lanes)
  static constexpr int NJW = WROWS / 16;                           // weight DMA instructions per phase (16 slab rows each): 18 / 9
  static constexpr int NIW = (NJW + 3) / 4;                        // ... per loader wave, at most
  static constexpr int RSTR = CO + 1;                              // floats per position of the reduction area (+ 1: the
                                                                   // lanes of a store hit different banks)
  static constexpr int RED = 4 * 2 * 64 * RSTR * 4;                // [wave][own | ctx][position][co] fp32 = 66 KB (aliases the ring)
  static_assert(NSLOT * SLOTB >= RED && NSLOT * SLOTB <= 160 * 1024, "ring holds the reduction area");
};

template <int CO>
__global__ __launch_bounds__(512, 2) void conv_eval1_kernel(const ConvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = Eval1Cfg<CO>;
  constexpr int CG = CO / 4;                       // 4-channel groups per position in the epilogue: 64 * CG threads take part
  constexpr int HW_ = Cfg::HW, SLOTB = Cfg::SLOTB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::NSLOT * SLOTB];
  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Cin = a.Cin, HWp = H * W;
  // tile: channel block fastest (the blocks of one pixel tile share its halo in L2), then x, y, sequence
  // (a three-dimensional grid: one division by a run-time value instead of three in front of the first address of the launch)
  const int co0 = (int)blockIdx.x * CO, x0 = (int)blockIdx.y * 8;
  const int b = (int)blockIdx.z / d.nty, y0 = ((int)blockIdx.z - b * d.nty) * 8;
  // phases per 32-channel chunk: own | cached frame 0 | cached frame 1, or (ctx_prod_mode 2) the own one only, or (3) the two
  // context ones only
  const int mode = a.ctx_prod_mode;
  const int phs = (mode == 2) ? 1 : (mode == 3) ? 2 : 3, ph0 = (mode == 3) ? 1 : 0;
  const int NP = (Cin / 32) * phs;
  // (mode 2) this thread's four context sums, requested now: nothing else of this wave is in flight yet, and the loader waves'
  // counted waits only ever leave YOUNGER requests outstanding
  float4 y3v = make_float4(0.f, 0.f, 0.f, 0.f);
  // ... and the epilogue's other operands -- the residual (EPI_MPSUM: 4 bf16) or the emb-scale row (EPI_EMB_SILU: 4 floats) of this
  // thread's four outputs and the two gate coefficients: behind the last barrier each was a round trip of its own (round 6: 2.1 K
  // cycles from the last barrier to the end of a 19 K-cycle launch)
  // Every wave requests them behind its last phase (the two barriers and the hand-over of the partial tiles cover the round trip): the
  // compute waves' registers are full during the phases, and in a loader wave an older request would sit in front of every counted
  // wait for a phase (vmcnt retires in order).  Four scalars, not a vector: hipcc (ROCm 7.2) let the gate coefficients' defaults
  // overwrite elements of a 4-vector that was live across the branch.
  unsigned epre0 = 0u, epre1 = 0u, epre2 = 0u, epre3 = 0u;
  float cown_pre = 1.f, cctx_pre = 1.f;
  if (mode == 2 && tid < 64 * CG) {
    const int pos = tid / CG, cq = (tid % CG) * 4;
    if (co0 + cq < a.Cout)
      y3v = *(const float4*)(a.ctx_prod + ((size_t)b * HWp + (size_t)(y0 + (pos >> 3)) * W + (x0 + (pos & 7))) * a.Cout + co0 + cq);
  }
  // (E1 / E2 below: LDS traffic only, so the barrier waits for LDS only -- __syncthreads() would also wait for these requests, gfx9's
  // vmcnt counting loads and stores alike)
#define EVAL1_RAW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define EVAL1_LOAD_EPILOGUE_OPERANDS()                                                                                          \
  do {                                                                                                                          \
    if (mode != 3) {                                                                                                            \
      const int pos_ = tid / CG, cq_ = (tid % CG) * 4;                                                                          \
      if (tid < 64 * CG && co0 + cq_ < a.Cout) {                                                                                               \
        const size_t o_ = ((size_t)b * HWp + (size_t)(y0 + (pos_ >> 3)) * W + (x0 + (pos_ & 7))) * a.Cout + co0 + cq_;          \
        if (a.epi == ONIRIS_EPI_MPSUM) {                                                                                        \
          const uint2 t2_ = *(const uint2*)((const bf16*)a.res + o_);                                                           \
          epre0 = t2_.x; epre1 = t2_.y;                                                                                         \
        } else if (a.epi == ONIRIS_EPI_EMB_SILU) {                                                                              \
          const float4 t4_ = *(const float4*)((const float*)a.escale + (size_t)b * (a.escale_pitch ? a.escale_pitch : a.Cout) + co0 + cq_); \
          epre0 = __builtin_bit_cast(unsigned, t4_.x); epre1 = __builtin_bit_cast(unsigned, t4_.y);                             \
          epre2 = __builtin_bit_cast(unsigned, t4_.z); epre3 = __builtin_bit_cast(unsigned, t4_.w);                             \
        }                                                                                                                       \
      }                                                                                                                         \
      if (a.coef_own) cown_pre = a.coef_own[b];                                                                                 \
      if (a.coef_ctx) cctx_pre = a.coef_ctx[b];                                                                                 \
    }                                                                                                                           \
  } while (0)
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
#ifdef EVAL1_STAMP
  // diagnostic build (make variant VSRC=conv_fwd_s1ctx VNAME=e1stamp VDEF=-DEVAL1_STAMP; scratch/r06_eval1_stamp.py): shader-clock stamps of
  // compute wave 0 ([0..15]) and loader wave 4 ([16..31]) of every workgroup into the (reserved, unused) emb_gain pointer
  long long* st_ = a.emb_gain ? (long long*)a.emb_gain + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 40 : nullptr;
#define E1STAMP(i) do { if (st_ && (tid & 255) == 0) st_[(wave >> 2) * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  if (st_ && tid == 0) st_[32] = (long long)__builtin_amdgcn_s_memrealtime();
#else
#define E1STAMP(i) do { } while (0)
#endif
  E1STAMP(0);

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    constexpr int OOB = (int)0x80000000;
    constexpr int TOTA = Cfg::HALO * 4, TOTW = Cfg::WROWS * 4;
    int adesc[Cfg::NIA];
#pragma unroll
    for (int i = 0; i < Cfg::NIA; ++i) {
      const int e = (i * 4 + lw) * 64 + lane;
      const int row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
      const int y = y0 + row / HW_ - 1, x = x0 + row % HW_ - 1;
      adesc[i] = OOB;
      if (e < TOTA && row % HW_ < 10 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
        adesc[i] = ((y * W + x) * Cin + gp * 8) * 2;
    }
    // weight pieces: instruction j = i * 4 + lw takes slab rows 16 j .. 16 j + 15 = tap j / 2, output channels 16 (j & 1) + lane / 4,
    // piece (lane & 3) ^ ((lane >> 4) & 3) -- the lane's part is the same for every instruction, the rest is wave-uniform and goes
    // into the scalar offset (one descriptor instead of five; round 6: the descriptors were 1.4 K cycles of a 19 K-cycle launch)
    const int wlane = ((lane >> 2) * a.CinP + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2;
    static_assert(TOTW == Cfg::NJW * 64, "sixteen slab rows per weight instruction");
    E1STAMP(12);                                   // (descriptors computed)
    const int frame_bytes = HWp * Cin * 2;
    const int wbytes = 9 * a.CoutP * a.CinP * 2;
    const i32x4 rs_x = make_rsrc((const bf16*)a.x + (size_t)b * HWp * Cin, frame_bytes);
    const i32x4 rs_c = make_rsrc((const bf16*)a.ctx + (size_t)b * a.ctx_bstride * HWp * Cin, 2 * frame_bytes);
    const i32x4 rs_wo = make_rsrc(a.w_own, wbytes), rs_wc = make_rsrc(a.w_ctx, 2 * wbytes);
    auto issue = [&](int p) __attribute__((always_inline)) {
      const int ch = p / phs, ph = ph0 + p - phs * ch, c0 = ch * 32;
      const unsigned dst = lds0 + (p % Cfg::NSLOT) * SLOTB + lw * 1024;
      if (ph == 0) {
#pragma unroll
        for (int i = 0; i < Cfg::NIA; ++i)
          if ((i * 4 + lw) * 64 + lane < TOTA) dma16(rs_x, adesc[i], c0 * 2, dst + i * 4096);
      } else {
#pragma unroll
        for (int i = 0; i < Cfg::NIA; ++i)
          if ((i * 4 + lw) * 64 + lane < TOTA) dma16(rs_c, adesc[i], (ph - 1) * frame_bytes + c0 * 2, dst + i * 4096);
      }
      const int sw = ((co0 * a.CinP + c0) + ((ph == 2) ? 9 * a.CoutP * a.CinP : 0)) * 2;
      const unsigned wdst = dst + Cfg::HALO * 64;
#pragma unroll
      for (int i = 0; i < Cfg::NIW; ++i) {
        const int j = i * 4 + lw;                  // (wave-uniform)
        if (j < Cfg::NJW) {
          const int so = sw + ((CO == 32) ? (((j >> 1) * a.CoutP + (j & 1) * 16) * a.CinP) * 2 : (j * a.CoutP * a.CinP) * 2);
          if (ph == 0) dma16(rs_wo, wlane, so, wdst + i * 4096);
          else dma16(rs_wc, wlane, so, wdst + i * 4096);
        }
      }
    };
    constexpr int AHEAD = Cfg::NSLOT - 1;
    E1STAMP(13);                                   // (resources built)
    // Phase 0 goes out alone and the compute waves start on it as soon as it has landed; the ring is topped up behind barrier_0.
    // (Rounds 4-5 issued three phases first: issuing a phase takes a loader wave 0.8-1 K cycles -- the four waves' 28 instructions
    // share one address path -- so the first MFMA waited 4.5 K cycles for bytes that had landed after 3.4 K: scratch/r06_eval1_stamp.py)
    int next = 0;
    issue(next++);
    E1STAMP(1);
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
      // requests so far: phases 0 .. next - 1; barrier_p needs phase p landed and leaves the k younger ones in flight
      const int k = next - 1 - p;
      // this wave's DMA instructions per phase: 2 halo + its share of the NJW weight instructions (7 / 6 with 32 channels, 5 / 4 with 16)
      const int per = Cfg::NIA + (Cfg::NJW - lw + 3) / 4;
      switch (k * per) {                           // (k <= AHEAD - 1 = 2; any other value over-waits)
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      if (p < 8) E1STAMP(2 + p);                   // (this phase's bytes have landed for this loader wave)
      __syncthreads();                             // barrier_p: phase p landed, phase p - 1 released
      // top-up: two phases behind barrier_0 (the first compute phase is the long one: cold instruction cache), one per barrier
      // after that -- the compute waves need phase p + 1 one phase time (~1.3 K cycles) from here, and issuing takes 0.9 K per phase:
      // a second issue in front of the wait made them wait for it (slot of phase next: phase next - NSLOT <= p - 1 is consumed)
#pragma unroll 1
      for (int q = 0; q < (p == 0 ? 2 : 1) && next < NP && next <= p + AHEAD; ++q) issue(next++);
    }
    EVAL1_LOAD_EPILOGUE_OPERANDS();
  } else {
    // ------------------------------------------------------------------------------------------------ compute waves
    const int r = lane & 31, h = lane >> 5;
    // lane -> position inside a 32-position half (4 patch rows x 8): a 16-lane read group takes patch rows (0,2) / (1,3),
    // i.e. halo rows 24 = 8 (mod 16) apart (conv_glds.h, PW == 8)
    const bool ga = (r < 4) || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int k = ga ? ((r < 4) ? r : (r < 16) ? r - 8 : r - 12) : ((r < 12) ? r - 4 : (r < 20) ? r - 8 : r - 16);
    const int pr = ((k >> 3) * 2 + (ga ? 0 : 1)) * 8 + (k & 7);
    int xa[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) xa[m] = (4 * m + (pr >> 3)) * HW_ + (pr & 7);          // halo row of tap (0, 0)
    const int rw_ = r & (CO - 1);                  // (CO == 16: lanes 16..31 read the rows of lanes 0..15, their MFMA rows are dropped)
    const int wa0 = rw_ * 64 + ((h ^ ((rw_ >> 2) & 3)) << 4);
    f32x16 acc[2][2];                              // [own | ctx][position half]
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }
    // the 18 (tap, k-step) products of a phase dealt to the four waves: wave w takes k-step w & 1 of taps (w >> 1) + 2 t -- five taps
    // for waves 0 and 1, four for waves 2 and 3 (10 / 8 MFMAs; rounds 4-5 dealt whole taps, 3 / 2 / 2 / 2 = 12 / 8 / 8 / 8)
    const int ksw = wave & 1, tap0 = wave >> 1, ntap = (wave < 2) ? 5 : 4;
#ifdef EVAL1_STAMP
    bool stamp_now = false;
#endif
    // the fifteen fragment offsets inside a slot, once (they were ~6 address instructions in front of every read of every phase)
    int woff[5], xoff[5][2];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int tap = tap0 + 2 * t;                // (t = 4 of waves 2, 3: tap 9 / 10, never read)
      const int toff = (tap / 3) * HW_ + (tap % 3);
      woff[t] = (wa0 ^ (ksw * 32)) + tap * CO * 64;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int R = xa[m] + toff;
        xoff[t][m] = R * 64 + (((2 * ksw + h) ^ ((R >> 2) & 3)) << 4);
      }
    }
    auto phase = [&](auto own_, const unsigned char* base) __attribute__((always_inline)) {
      constexpr int WHICH = decltype(own_)::value ? 0 : 1;
      const unsigned char* wbase = base + Cfg::HALO * 64;
      // every fragment read of the phase goes out first (15 / 12 ds_read_b128 per wave), then the MFMAs: read -> wait -> MFMA per
      // pair, as the compiler orders the plain loop nest, exposes the LDS latency (~130 cycles, more under the DMA writes) in front
      // of every pair (scratch/r06_eval1_stamp.py: 750 cycles for 12 reads + 8 MFMAs, 540 for 6 + 4)
      bf16x8 wf[5], xf[5][2];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
          wf[t] = *(const bf16x8*)(wbase + woff[t]);
#pragma unroll
          for (int m = 0; m < 2; ++m) xf[t][m] = *(const bf16x8*)(base + xoff[t][m]);
        }
      }
#ifdef EVAL1_STAMP
      if (stamp_now) E1STAMP(13);
#endif
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[WHICH][m] = mfma32(wf[t], xf[t][m], acc[WHICH][m]);
        }
      }
    };
    __syncthreads();                               // barrier_0: phase 0 landed
    E1STAMP(1);
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
      if (p > 0) __syncthreads();                  // barrier_p
#ifdef EVAL1_STAMP
      stamp_now = (p == 2) || (NP < 3 && p == NP - 1);
      if (stamp_now) E1STAMP(12);
#endif
      const unsigned char* base = smem + (p % Cfg::NSLOT) * SLOTB;
      if (ph0 + p % phs == 0) phase(std::true_type{}, base);
      else phase(std::false_type{}, base);
      if (p < 8) E1STAMP(2 + p);
    }
    EVAL1_LOAD_EPILOGUE_OPERANDS();
    // partial tiles -> LDS as [wave][own | ctx][position][co] fp32 (after E1: every wave is done with the ring)
    EVAL1_RAW_BARRIER();                                 // E1
    float* red = (float*)smem + wave * (2 * 64 * Cfg::RSTR);
#pragma unroll
    for (int wch = 0; wch < 2; ++wch)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int rr = 0; rr < CO / 2; ++rr)       // (MFMA rows 8 (rr / 4) + 4 h + rr % 4 < CO)
          red[(wch * 64 + m * 32 + pr) * Cfg::RSTR + mfma_row(rr, lane)] = acc[wch][m][rr];
  }
  if (wave >= 4) EVAL1_RAW_BARRIER();                    // E1 (loader side)
  EVAL1_RAW_BARRIER();                                   // E2: the partial tiles are in LDS
  E1STAMP(10);

  // ---------------------------------------------------------------------------------------------------- epilogue
  // thread -> (position, 4 consecutive output channels): 64 positions x CG channel groups = 512 / 256 threads
  if (tid < 64 * CG) {
    const int pos = tid / CG, cq = (tid % CG) * 4;
    const int py = pos >> 3, px = pos & 7;         // (position index m*32 + pr: patch row 4m + (pr >> 3), column pr & 7)
    const float* red = (const float*)smem;
    float own[4] = {0.f, 0.f, 0.f, 0.f}, ctx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* po = red + (size_t)w * (2 * 64 * Cfg::RSTR) + (0 * 64 + pos) * Cfg::RSTR + cq;
      const float* pc = red + (size_t)w * (2 * 64 * Cfg::RSTR) + (1 * 64 + pos) * Cfg::RSTR + cq;
#pragma unroll
      for (int i = 0; i < 4; ++i) { own[i] += po[i]; ctx[i] += pc[i]; }
    }
    const int co = co0 + cq;
    if (co < a.Cout) {
      const int n = b;                             // frame-slot index (S == 1, T == 1)
      const size_t o = ((size_t)n * HWp + (size_t)(y0 + py) * W + (x0 + px)) * a.Cout + co;
      if (mode == 2) { ctx[0] = y3v.x; ctx[1] = y3v.y; ctx[2] = y3v.z; ctx[3] = y3v.w; }
      else if (mode != 0) *(float4*)(a.ctx_prod + o) = make_float4(ctx[0], ctx[1], ctx[2], ctx[3]);
      if (mode == 3) return;
      const float cown = cown_pre, cctx = cctx_pre;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(cctx, ctx[i], cown * own[i]);
      bf16x4 ov;
      if (a.ctx_out) {                             // un-gated context product (kept for d gate; NULL in the sampler)
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = f2bf(ctx[i]);
        *(bf16x4*)((bf16*)a.ctx_out + o) = ov;
      }
      if (a.epi == ONIRIS_EPI_MPSUM) {
        if (a.out2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ov[i] = f2bf(v[i]);
          *(bf16x4*)((bf16*)a.out2 + o) = ov;
        }
        const uint2 t2 = make_uint2(epre0, epre1);
        const bf16x4 rv = __builtin_bit_cast(bf16x4, t2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t = a.ta * bf2f(rv[i]) + a.tb * v[i];
          if (a.clip > 0.f) t = fminf(fmaxf(t, -a.clip), a.clip);
          ov[i] = f2bf(t);
        }
        *(bf16x4*)((bf16*)a.out + o) = ov;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = f2bf(v[i]);
        *(bf16x4*)((bf16*)a.out + o) = ov;
        if (a.epi == ONIRIS_EPI_EMB_SILU) {
          const float cv[4] = {__builtin_bit_cast(float, epre0), __builtin_bit_cast(float, epre1), __builtin_bit_cast(float, epre2),
                               __builtin_bit_cast(float, epre3)};
          bf16x4 o2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float z = bf2f(ov[i]) * cv[i];             // the activation sees the bf16-rounded y
            o2[i] = f2bf(z * sigmoid_fast(z) * (1.f / 0.596f));
          }
          *(bf16x4*)((bf16*)a.out2 + o) = o2;
        }
      }
    }
  }
  E1STAMP(11);
#ifdef EVAL1_STAMP
  if (st_ && tid == 0) st_[33] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#endif
}

// true when the weight-streaming kernel can run this problem (one new frame per sequence against the cached pair)
static inline bool conv_eval1_ok(const OnirisConvArgs& a) {
  return a.S == 1 && a.T == 1 && a.ctx && a.w_ctx && a.taps == 9 && a.ctx_T == 2 && a.coff0 == 0 && a.coff1 == 1 &&
         a.Cin % 32 == 0 && a.Cin >= 32 && a.H % 8 == 0 && a.W % 8 == 0 && a.Cout % 8 == 0 &&
         a.ctx_prod_mode >= 0 && a.ctx_prod_mode <= 3 && (a.ctx_prod_mode == 0 || a.ctx_prod != nullptr) &&
         2LL * a.H * a.W * a.Cin * 2 < (1LL << 31) && 18LL * a.CoutP * a.CinP * 2 < (1LL << 31) &&
         (a.escale_pitch == 0 || a.escale_pitch % 4 == 0) && (long long)(a.H / 8) * a.B <= 65535 && a.W / 8 <= 65535;
}

template <int CO>
static int launch_conv_eval1_co(const OnirisConvArgs& a, hipStream_t stream) {
  ConvDev d;
  memset(&d, 0, sizeof(d));
  d.a = a;
  d.ncob = a.CoutP / CO;
  d.ntx = a.W / 8; d.nty = a.H / 8; d.ntt = 1;
  const long long nz = (long long)d.nty * a.B;
  if (d.ncob <= 0 || d.ntx <= 0 || d.ntx > 65535 || nz <= 0 || nz > 65535) { oniris_set_error("conv_fwd: bad grid %d x %d x %lld", d.ncob, d.ntx, nz); return ONIRIS_EINVAL; }
  oniris_launch(conv_eval1_kernel<CO>, dim3((unsigned)d.ncob, (unsigned)d.ntx, (unsigned)nz), dim3(512), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

static int launch_conv_eval1(const OnirisConvArgs& a, hipStream_t stream) {
  // 16 output channels per workgroup wherever 32 would leave the launch on <= 128 CUs (every level of one sequence: 8 .. 64
  // workgroups; 39.2 -> 40.4 frames/s against doing it from 128 input channels on only); big_tile bit 256 = always 32 (A/B, tests)
  const long long wg32 = (long long)(a.CoutP / 32) * (a.W / 8) * (a.H / 8) * a.B;
  if (wg32 <= 128 && !(a.big_tile & 256)) return launch_conv_eval1_co<16>(a, stream);
  return launch_conv_eval1_co<32>(a, stream);
}
