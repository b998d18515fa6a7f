// 1x1 convolution (MPConv with a 1x1 / linear weight, edm2/conv.py:36-42: attn_qkv, attn_proj, conv_skip and their
// data gradients) as a persistent LDS-DMA GEMM for gfx950:  out[pos][co] = sum_ci x[pos][ci] * W[co][ci].
//
// HBM-bound (K = Cin is 64..768): what matters is that activations are read once, outputs written once, and that
// enough bytes are in flight per CU.  One workgroup = 8 waves = a tile of 256 positions x 128 output channels
// (4 position waves x 2 channel waves, each wave 64 positions x 64 channels = 4 MFMA 32x32x16 accumulators: every
// fragment read from LDS feeds two MFMAs); K is walked in 64-channel chunks through THREE staging buffers
// (48 KB each: 256 x 128-byte activation rows + 128 x 128-byte weight rows), so the copies of chunks j+1 and j+2 are
// in flight while chunk j is multiplied -- the chunk sequence runs on across tile boundaries, i.e. the first chunks of
// the next tile land under the epilogue of the current one.  Tiles that share their positions (the other channel
// blocks) are consecutive in a workgroup's run, so the activation chunk is re-read from L2, not HBM.
// Rows are 128 bytes, unpadded (the DMA image is lane-linear); the 16-byte pieces are XOR-swizzled on the source side
// with (row >> 1) & 7, which makes the 16-row groups of a ds_read_b128 conflict-free (same scheme as the attention
// K tile).  Epilogues: none / mp_sum(+clip) with optional raw output, as conv_kernels.h.
#pragma once
#include "conv_kernels.h"
#include "lds_dma.h"

struct C1Cfg {
  static constexpr int BM = 256, BN = 128, CK = 64, NW = 8, NTHR = 512, NSTG = 3;
  static constexpr int STG = (BM + BN) * 128;                       // bytes per stage
  static constexpr int NPA = BM * 8 / NTHR, NPW = BN * 8 / NTHR;    // 16-byte pieces per thread and chunk: 4 + 2
  static constexpr int EROW = 64 * 2 + 16;                          // epilogue staging row (64 channels of one position)
  static_assert(NW * 32 * EROW <= STG, "the epilogue stages through one (free) stage buffer");
  static_assert(NSTG * STG <= 160 * 1024, "stages must fit the LDS");
};

__global__ __launch_bounds__(512, 1) void conv1x1_glds_kernel(const ConvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = C1Cfg;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, CK = Cfg::CK, NTHR = Cfg::NTHR, STG = Cfg::STG, NSTG = Cfg::NSTG;
  constexpr int NPA = Cfg::NPA, NPW = Cfg::NPW, EROW = Cfg::EROW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTG * STG];
  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int wp = wave & 3, wc = wave >> 2;                   // position group (64 positions) / channel group (64 channels)
  const int Cin = a.Cin, Cout = a.Cout;
  const long long M = (long long)a.B * a.S * a.T * a.H * a.W;
  const int nchunk = Cin / CK, ncb = d.ncob, ntiles = d.ntt * ncb;

  // this workgroup's contiguous run of tiles inside its XCD's range (tile = position tile * ncb + channel block)
  int tl, tl_hi, tl_step;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int lo = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    tl_hi = lo + q + ((xcd < rr) ? 1 : 0);
    tl_step = (nwg - xcd + 7) >> 3;
    const int cnt = tl_hi - lo, per = (cnt + tl_step - 1) / tl_step;        // contiguous sub-runs: channel blocks of
    tl = lo + (blockIdx.x >> 3) * per;                                      // one position tile stay together
    tl_hi = (tl + per < tl_hi) ? tl + per : tl_hi;
  }
  if (tl >= tl_hi) return;
  const int nitem = (tl_hi - tl) * nchunk;

  // DMA descriptors (tile-invariant per-lane offsets; tile / chunk go into the uniform soffset)
  int adesc[NPA], wdesc[NPW];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int e = i * NTHR + tid, row = e >> 3, pc = (e & 7) ^ ((row >> 1) & 7);
    adesc[i] = (row * Cin + pc * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int e = i * NTHR + tid, row = e >> 3, pc = (e & 7) ^ ((row >> 1) & 7);
    wdesc[i] = (row * a.CinP + pc * 8) * 2;
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  const size_t xbytes = (size_t)M * Cin * 2;
  auto issue = [&](int item) __attribute__((always_inline)) {
    const int t = tl + item / nchunk, ch = item % nchunk, stg = item % NSTG;
    const int mt = t / ncb, nb = t % ncb;
    const size_t m0 = (size_t)mt * BM;
    // activation rows m0 .. m0+255 (rows beyond M: the buffer range check returns zeros)
    const size_t left = xbytes - m0 * Cin * 2;
    const i32x4 rs_x = make_rsrc((const bf16*)a.x + m0 * Cin, (int)(left < (size_t)BM * Cin * 2 ? left : (size_t)BM * Cin * 2));
    const int co0 = nb * BN;
    const int wleft = (a.CoutP - co0) * a.CinP * 2;
    const i32x4 rs_w = make_rsrc((const bf16*)a.w_own + (size_t)co0 * a.CinP, wleft < BN * a.CinP * 2 ? wleft : BN * a.CinP * 2);
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + stg * STG + wave * 1024);
#pragma unroll
    for (int i = 0; i < NPA; ++i) dma16(rs_x, adesc[i], ch * CK * 2, dst + i * (NTHR * 16));
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma16(rs_w, wdesc[i], ch * CK * 2, dst + BM * 128 + i * (NTHR * 16));
  };

  // fragment addresses inside a stage (k-step ks: piece index 2*ks + h before the swizzle)
  int xa[2], wa[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int row = wp * 64 + m * 32 + r;
    xa[m] = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int row = wc * 64 + n * 32 + r;
    wa[n] = BM * 128 + row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
  }

  f32x16 acc[2][2];
  issue(0);
  if (nitem > 1) issue(1);
  bool drained = false;              // stores were issued since the last full wait (vmcnt counts them too)
#pragma unroll 1
  for (int item = 0; item < nitem; ++item) {
    const int ch = item % nchunk, stg = item % NSTG;
    // chunk `item` must have landed; chunk item+1 (NPA + NPW newer loads of this wave) may still be in flight
    if (item + 1 < nitem && !drained) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA + NPW) : "memory");
    else dma_wait();
    drained = false;
    __syncthreads();                 // everybody's share has landed; stage (item+2) % 3 was consumed in item-1
    if (item + 2 < nitem) issue(item + 2);
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    }
    const unsigned char* base = smem + stg * STG;
    bf16x8 xf[4][2], wf[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int m = 0; m < 2; ++m) xf[ks][m] = *(const bf16x8*)(base + (xa[m] ^ (ks * 32)));
#pragma unroll
      for (int n = 0; n < 2; ++n) wf[ks][n] = *(const bf16x8*)(base + (wa[n] ^ (ks * 32)));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(wf[ks][n], xf[ks][m], acc[m][n]);
    if (ch + 1 < nchunk) continue;

    // ---------------------------------------------------------------- epilogue of the tile (lane = position)
    const int t = tl + item / nchunk;
    const long long m0 = (long long)(t / ncb) * BM;
    const int cow = (t % ncb) * BN + wc * 64;                  // first channel of this wave
    __syncthreads();                                           // every wave is done reading stage `stg`
    unsigned char* ep = smem + stg * STG + wave * 32 * EROW;
    bf16* og = (bf16*)a.out;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const long long prow = m0 + wp * 64 + m * 32;            // first position of this 32-position block
      auto put = [&](int nt, const float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = f2bf(v[4 * g + k]);
          *(bf16x4*)(ep + r * EROW + (nt * 32 + 8 * g + 4 * h) * 2) = o;
        }
      };
      auto flush = [&](bf16* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int id = it * 64 + lane, row = id >> 3, part = id & 7;
          const int co = cow + part * 8;
          if (prow + row < M && co < Cout) {
            const u32x4 v_ = *(const u32x4*)(ep + row * EROW + part * 16);
            u32x4* o_ = (u32x4*)(dst + (size_t)(prow + row) * Cout + co);
            if (d.nt) __builtin_nontemporal_store(v_, o_); else *o_ = v_;
          }
        }
      };
      float v[16];
      if (a.epi == ONIRIS_EPI_MPSUM) {
        if (a.out2) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[m][nt][i];
            put(nt, v);
          }
          flush((bf16*)a.out2);
        }
        const bool valid = prow + r < M;
        const size_t obase = (size_t)(prow + r) * Cout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co = cow + nt * 32 + 8 * g + 4 * h;
            bf16x4 rv;
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[k] = f2bf(0.f);
            if (valid && co < Cout) rv = *(const bf16x4*)((const bf16*)a.res + obase + co);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float o = a.ta * bf2f(rv[k]) + a.tb * acc[m][nt][4 * g + k];
              if (a.clip > 0.f) o = fminf(fmaxf(o, -a.clip), a.clip);
              v[4 * g + k] = o;
            }
          }
          put(nt, v);
        }
        flush(og);
      } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[m][nt][i];
          put(nt, v);
        }
        flush(og);
      }
    }
    drained = true;                  // global stores (and res loads) are in the vmcnt stream now: next wait is a full one
  }
#endif
}

static inline bool conv1x1_glds_ok(const OnirisConvArgs& a) {
  const long long M = (long long)a.B * a.S * a.T * a.H * a.W;
  return a.taps == 1 && !a.ctx && a.Cin % 64 == 0 && a.CinP == a.Cin && a.Cin <= 1024 && a.Cout % 8 == 0 &&
         (a.epi == ONIRIS_EPI_NONE || a.epi == ONIRIS_EPI_MPSUM) && M >= 8192 && M * a.Cin * 2 < (1LL << 31) &&
         (long long)a.CoutP * a.CinP * 2 < (1LL << 31);
}

static int launch_conv1x1_glds(const OnirisConvArgs& a, hipStream_t stream) {
  ConvDev d;
  d.a = a;
  const long long M = (long long)a.B * a.S * a.T * a.H * a.W;
  d.ntt = (int)((M + C1Cfg::BM - 1) / C1Cfg::BM);
  d.ncob = cdiv(a.CoutP, C1Cfg::BN);
  d.ntx = d.nty = 1; d.ksplit = 1; d.reduce = 0;
  d.nt = M * a.Cout * 2 >= oniris_ew_nt_bytes();
  const long long ntiles = (long long)d.ntt * d.ncob;
  const int ncu = oniris_persistent_wgs();       // one workgroup per CU (minus the CUs reserved for a gradient exchange in flight)
  const long long nblk = ntiles < ncu ? ntiles : ncu;
  oniris_launch_tagged(d.nt ? "nt-stores" : nullptr, conv1x1_glds_kernel, dim3((unsigned)nblk), dim3(C1Cfg::NTHR), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
