// Gated causal 3x3 convolution of ONE generated frame per sequence (the KV-cached sampler, edm2/sampler.py:12-85: 31 UNet
// evaluations per frame, each a chain of 58 of these) -- weight-streaming kernel for gfx950.
//
// A one-frame conv has 64 .. 4096 output positions and K = 27 * Cin: a handful of tiles, each with a long reduction whose
// weights (up to 3.5 MB per layer, 92 MB per evaluation) must come from L2 / HBM.  conv_fwd_kernel serves it as a split-K
// pair of launches (partials through a workspace, ~14 + ~14 us, bound by the serial latency of its register-staged phases
// and by the second launch).  Here ONE launch does it: a workgroup owns an 8x8-pixel x 32-channel output tile and STREAMS
// its whole K through a four-slot LDS ring:
//   * waves 4..7 = loaders: per phase (32 input channels x {own frame | cached frame 0 | cached frame 1}) one 10x12 halo
//     image (7.5 KB) and the 9-tap weight slab of the tile's 32 output channels (18 KB) by LDS-DMA, three phases ahead of
//     the consumers (counted vmcnt, one barrier per phase) -- the reduction runs at the CU's L2 -> LDS copy rate instead of
//     at one round trip per phase;
//   * waves 0..3 = compute: they split the nine TAPS of a phase (wave 0: taps 0, 4, 8; wave w: taps w, w + 4), both
//     32-position halves of the tile each: D[co][position] += W[tap][co][ci] . X[position + tap][ci], MFMA 32x32x16;
//     own and context products keep separate accumulators (the gate mixes them in the epilogue);
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

struct Eval1Cfg {
  static constexpr int HW = 12, HH = 10, HALO = HH * HW;           // halo rows of an 8x8 tile (12-wide: see conv_glds.h)
  static constexpr int WROWS = 9 * 32;
  static constexpr int SLOTB = (HALO + WROWS) * 64;               // 26112 B
  static constexpr int NSLOT = 4;
  static constexpr int NIA = 2, NIW = 5;                           // DMA instructions per loader wave and phase (4 waves x 64 lanes)
  static constexpr int RSTR = 33;                                  // floats per position of the reduction area (32 + 1: the
                                                                   // 32 lanes of a store hit 32 banks)
  static constexpr int RED = 4 * 2 * 64 * RSTR * 4;                // [wave][own | ctx][position][co] fp32 = 66 KB (aliases the ring)
  static_assert(NSLOT * SLOTB >= RED && NSLOT * SLOTB <= 160 * 1024, "ring holds the reduction area");
};

__global__ __launch_bounds__(512, 2) void conv_eval1_kernel(const ConvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = Eval1Cfg;
  constexpr int HW_ = Cfg::HW, SLOTB = Cfg::SLOTB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::NSLOT * SLOTB];
  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Cin = a.Cin, HWp = H * W;
  // tile: channel block fastest (the blocks of one pixel tile share its halo in L2), then x, y, sequence
  int bid = blockIdx.x;
  const int co0 = (bid % d.ncob) * 32; bid /= d.ncob;
  const int x0 = (bid % d.ntx) * 8; bid /= d.ntx;
  const int y0 = (bid % d.nty) * 8; bid /= d.nty;
  const int b = bid;
  // phases per 32-channel chunk: own | cached frame 0 | cached frame 1, or (ctx_prod_mode 2) the own one only, or (3) the two
  // context ones only
  const int mode = a.ctx_prod_mode;
  const int phs = (mode == 2) ? 1 : (mode == 3) ? 2 : 3, ph0 = (mode == 3) ? 1 : 0;
  const int NP = (Cin / 32) * phs;
  // (mode 2) this thread's four context sums, requested now: nothing else of this wave is in flight yet, and the loader waves'
  // counted waits only ever leave YOUNGER requests outstanding
  float4 y3v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (mode == 2) {
    const int pos = tid >> 3, cq = (tid & 7) * 4;
    if (co0 + cq < a.Cout)
      y3v = *(const float4*)(a.ctx_prod + ((size_t)b * HWp + (size_t)(y0 + (pos >> 3)) * W + (x0 + (pos & 7))) * a.Cout + co0 + cq);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    constexpr int OOB = (int)0x80000000;
    constexpr int TOTA = Cfg::HALO * 4, TOTW = Cfg::WROWS * 4;
    int adesc[Cfg::NIA], wdesc[Cfg::NIW];
#pragma unroll
    for (int i = 0; i < Cfg::NIA; ++i) {
      const int e = (i * 4 + lw) * 64 + lane;
      const int row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
      const int y = y0 + row / HW_ - 1, x = x0 + row % HW_ - 1;
      adesc[i] = OOB;
      if (e < TOTA && row % HW_ < 10 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
        adesc[i] = ((y * W + x) * Cin + gp * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < Cfg::NIW; ++i) {
      const int e = (i * 4 + lw) * 64 + lane;
      const int row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
      const int tap = row / 32, co = row % 32;
      wdesc[i] = (e < TOTW) ? ((tap * a.CoutP + co) * a.CinP + gp * 8) * 2 : OOB;
    }
    const int frame_bytes = HWp * Cin * 2;
    const int wbytes = 9 * a.CoutP * a.CinP * 2;
    const i32x4 rs_x = make_rsrc((const bf16*)a.x + (size_t)b * HWp * Cin, frame_bytes);
    const i32x4 rs_c = make_rsrc((const bf16*)a.ctx + (size_t)b * a.ctx_bstride * HWp * Cin, 2 * frame_bytes);
    const i32x4 rs_wo = make_rsrc(a.w_own, wbytes), rs_wc = make_rsrc(a.w_ctx, 2 * wbytes);
    auto issue = [&](int p) __attribute__((always_inline)) {
      const int ch = p / phs, ph = ph0 + p - phs * ch, c0 = ch * 32;
      const unsigned dst = lds0 + (p & 3) * SLOTB + lw * 1024;
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
      if (ph == 0) {
#pragma unroll
        for (int i = 0; i < Cfg::NIW; ++i)
          if ((i * 4 + lw) * 64 + lane < TOTW) dma16(rs_wo, wdesc[i], sw, wdst + i * 4096);
      } else {
#pragma unroll
        for (int i = 0; i < Cfg::NIW; ++i)
          if ((i * 4 + lw) * 64 + lane < TOTW) dma16(rs_wc, wdesc[i], sw, wdst + i * 4096);
      }
    };
#pragma unroll 1
    for (int p = 0; p < 3 && p < NP; ++p) issue(p);
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
      // requests so far: phases 0 .. min(NP, p + 3) - 1; barrier_p needs phases <= p + 1 landed.  A phase is 480 halo +
      // 1152 weight pieces = 2 + 5 DMA instructions of loader waves 0, 1 and 2 + 4 of waves 2, 3 (their fifth weight
      // instruction would lie wholly beyond the slab and is never issued)
      if (p + 2 < NP) {
        if (lw < 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                             // barrier_p: phase p + 1 landed, phase p - 1 released
      if (p + 3 < NP) issue(p + 3);
    }
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
    const int wa0 = r * 64 + ((h ^ ((r >> 2) & 3)) << 4);
    f32x16 acc[2][2];                              // [own | ctx][position half]
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }
    const int ntap = (wave == 0) ? 3 : 2;
    auto phase = [&](auto own_, const unsigned char* base) __attribute__((always_inline)) {
      constexpr int WHICH = decltype(own_)::value ? 0 : 1;
      const unsigned char* wbase = base + Cfg::HALO * 64;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t < ntap) {
          const int tap = wave + 4 * t;
          const int toff = (tap / 3) * HW_ + (tap % 3);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 wf = *(const bf16x8*)(wbase + ((wa0 ^ (ks * 32)) + tap * 32 * 64));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const int R = xa[m] + toff;
              const bf16x8 xf = *(const bf16x8*)(base + R * 64 + (((2 * ks + h) ^ ((R >> 2) & 3)) << 4));
              acc[WHICH][m] = mfma32(wf, xf, acc[WHICH][m]);
            }
          }
        }
      }
    };
    __syncthreads();                               // barrier_0: phases 0 and 1 landed
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
      if (p > 0) __syncthreads();                  // barrier_p
      const unsigned char* base = smem + (p & 3) * SLOTB;
      if (ph0 + p % phs == 0) phase(std::true_type{}, base);
      else phase(std::false_type{}, base);
    }
    // partial tiles -> LDS as [wave][own | ctx][position][co] fp32 (after E1: every wave is done with the ring)
    __syncthreads();                               // E1
    float* red = (float*)smem + wave * (2 * 64 * Cfg::RSTR);
#pragma unroll
    for (int wch = 0; wch < 2; ++wch)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
          red[(wch * 64 + m * 32 + pr) * Cfg::RSTR + mfma_row(rr, lane)] = acc[wch][m][rr];
  }
  if (wave >= 4) __syncthreads();                  // E1 (loader side)
  __syncthreads();                                 // E2: the partial tiles are in LDS

  // ---------------------------------------------------------------------------------------------------- epilogue
  // thread -> (position, 4 consecutive output channels): 64 positions x 8 channel groups = 512 threads
  {
    const int pos = tid >> 3, cq = (tid & 7) * 4;
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
      const float cown = a.coef_own ? a.coef_own[n] : 1.f, cctx = a.coef_ctx ? a.coef_ctx[n] : 1.f;
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
        const bf16x4 rv = *(const bf16x4*)((const bf16*)a.res + o);
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
          const float4 ev = *(const float4*)((const float*)a.escale + (size_t)n * (a.escale_pitch ? a.escale_pitch : a.Cout) + co);
          const float cv[4] = {ev.x, ev.y, ev.z, ev.w};
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
#endif
}

// true when the weight-streaming kernel can run this problem (one new frame per sequence against the cached pair)
static inline bool conv_eval1_ok(const OnirisConvArgs& a) {
  return a.S == 1 && a.T == 1 && a.ctx && a.w_ctx && a.taps == 9 && a.ctx_T == 2 && a.coff0 == 0 && a.coff1 == 1 &&
         a.Cin % 32 == 0 && a.Cin >= 32 && a.H % 8 == 0 && a.W % 8 == 0 && a.Cout % 8 == 0 &&
         a.ctx_prod_mode >= 0 && a.ctx_prod_mode <= 3 && (a.ctx_prod_mode == 0 || a.ctx_prod != nullptr) &&
         2LL * a.H * a.W * a.Cin * 2 < (1LL << 31) && 18LL * a.CoutP * a.CinP * 2 < (1LL << 31) &&
         (a.escale_pitch == 0 || a.escale_pitch % 4 == 0);
}

static int launch_conv_eval1(const OnirisConvArgs& a, hipStream_t stream) {
  ConvDev d;
  memset(&d, 0, sizeof(d));
  d.a = a;
  d.ncob = a.CoutP / 32;
  d.ntx = a.W / 8; d.nty = a.H / 8; d.ntt = 1;
  const long long nblk = (long long)d.ntx * d.nty * a.B * d.ncob;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { oniris_set_error("conv_fwd: bad grid %lld", nblk); return ONIRIS_EINVAL; }
  oniris_launch(conv_eval1_kernel, dim3((unsigned)nblk), dim3(512), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
