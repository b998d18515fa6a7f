// Implicit-GEMM gated causal convolution for gfx950 (MFMA 32x32x16 bf16 -> fp32).
//
// Data layout in HBM: activations [frame][H][W][C] bf16 (channels-last, C % 16 == 0), packed weights
// [tap][CoutP][CinP] bf16 (ci contiguous).  One 256-thread workgroup (4 waves) computes a tile of
//   128 positions (FT frames x PH x PW patch)  x  S slots (clean|noised)  x  BN = 32*NT output channels.
// Wave w owns positions [32w, 32w+32): MFMA D[i = co][j = position], so every lane is ONE position and the
// epilogue (gate combine, emb-scale+SiLU, mp_sum+clip) is lane-local.
// Per 32-channel K chunk:  phase OWN: the S*HALO input pixels of the patch (+1 halo) are staged ONCE in LDS and
// reused by the 9 taps (each tap is a constant row offset into the halo image); phases CTX0/CTX1 do the same for
// the two previous clean frames whose product is SHARED by the clean and the noised slot (one accumulator).
// LDS rows are padded by 16 B (row = CK*2+16 bytes) which makes the 16-byte fragment reads conflict-free.
// Roofline: MFMA-bound for C >= 64 (intensity ~ 27*C FLOP/B of activation), HBM-bound below.
#pragma once
#include <type_traits>
#include <cstdlib>
#include "common.h"
#include "../../include/oniris.h"

template <int PW_, int NPOS_ = 128>
struct Patch {
  static constexpr int PW = PW_;
  static constexpr int NPOS = NPOS_;                       // positions per workgroup tile (32 per wave)
  static constexpr int PH = (PW_ == 16) ? (NPOS_ / 16) : PW_;
  static constexpr int FT = NPOS_ / (PW * PH);
  static constexpr int HW = PW + 2, HH = PH + 2;
  static constexpr int HALO = FT * HH * HW;
};

struct ConvDev {
  OnirisConvArgs a;
  int ntx, nty, ntt, ncob;
  int grid3d;            // conv_fwd_kernel, 1x1 without split-K: the grid is (position tile, batch element, channel block) -- no division
                         // by a run-time value in front of the first address (a few-tile launch is a latency chain; round 6)
  int ksplit;            // > 1: the (chunk, phase) rounds of a tile are dealt to `ksplit` workgroups (conv_fwd_kernel)
  int reduce;            // 1: second launch of a split-K conv -- sum the slices' partials and run the epilogue
  int nt;                // conv_glds_kernel: outputs go out with non-temporal stores (tensors beyond oniris_ew_nt_bytes())
};

template <int S, int TAPS, int CK, int NT, bool HAS_CTX, int PW, int NW = 4>
struct ConvCfg {
  using P = Patch<PW, 32 * NW>;
  static constexpr int NPOS = 32 * NW, NTHR = 64 * NW;
  static constexpr int BN = 32 * NT;
  static constexpr int ROWB = CK * 2 + 16;
  static constexpr int PARTS = CK / 8;
  static constexpr int AROWS = (TAPS == 9) ? S * P::HALO : S * NPOS;
  static constexpr int CROWS = (TAPS == 9) ? P::HALO : NPOS;
  static constexpr int WROWS = TAPS * BN;
  static constexpr int EPI_BYTES = NW * 32 * (BN * 2 + 16);        // wave-private transpose tiles of the epilogue
  static constexpr int LDS_BYTES = ((AROWS + WROWS) * ROWB > EPI_BYTES) ? (AROWS + WROWS) * ROWB : EPI_BYTES;
};

template <int S, int TAPS, int CK, int NT, bool HAS_CTX, int PW, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8) ? 2 : 2) void conv_fwd_kernel(const ConvDev d) {
  using Cfg = ConvCfg<S, TAPS, CK, NT, HAS_CTX, PW, NW>;
  using P = typename Cfg::P;
  constexpr int BN = Cfg::BN, ROWB = Cfg::ROWB, PARTS = Cfg::PARTS;
  constexpr int NTHR = Cfg::NTHR, NPOS = Cfg::NPOS;
  // static LDS (up to 160 KB on gfx950): no hipFuncSetAttribute, and hipGraph kernel nodes carry the size themselves
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::LDS_BYTES];
  unsigned char* A_lds = smem;
  unsigned char* W_lds = smem + Cfg::AROWS * ROWB;

  const OnirisConvArgs& a = d.a;
  // (reduce launch of a split-K conv: one 64-thread block per wave of the tile, see below)
  const bool red = d.reduce != 0;
  const int tid = red ? (int)(blockIdx.x % (unsigned)NW) * 64 + (int)threadIdx.x : (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, T = a.T, Cin = a.Cin, HWp = a.H * a.W;

  // ---- tile decode (split-K: consecutive workgroups are the K-slices of one tile)
  const int ks = d.ksplit, kz = (ks > 1 && !red) ? (int)(blockIdx.x % (unsigned)ks) : 0;
  const int tile_id = red ? (int)(blockIdx.x / (unsigned)NW) : (ks > 1) ? (int)(blockIdx.x / (unsigned)ks) : (int)blockIdx.x;
  int bid = tile_id;
  int t0 = 0, y0 = 0, x0 = 0, q0 = 0, b, co0;
  if (TAPS == 1 && d.grid3d) {
    q0 = (int)blockIdx.x * NPOS; b = (int)blockIdx.y; co0 = (int)blockIdx.z * BN;
  } else {
    if constexpr (TAPS == 9) {
      const int tx = bid % d.ntx; bid /= d.ntx;
      const int ty = bid % d.nty; bid /= d.nty;
      const int tc = bid % d.ntt; bid /= d.ntt;
      t0 = tc * P::FT; y0 = ty * P::PH; x0 = tx * P::PW;
    } else {
      const int tq = bid % d.ntt; bid /= d.ntt;      // ntt = number of 128-position tiles per (b,s)
      q0 = tq * NPOS;
    }
    b = bid % a.B;
    co0 = (bid / a.B) * BN;
  }

  // ---- this lane's position.  For 16-wide patches the 32 positions of a wave (2 patch rows x 16 px) are dealt to
  // the lanes so that each 16-lane group of a ds_read_b128 ({0-3,12-15,20-27} / {4-11,16-19,28-31}) reads 16
  // CONSECUTIVE halo rows: with 80-byte rows that is conflict-free (the natural order is 2-way on every read).
  int pr = r;
  if constexpr (TAPS == 9 && PW == 16) {
    const bool ga = (r < 4) || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int k = ga ? ((r < 4) ? r : (r < 16) ? r - 8 : r - 12) : ((r < 12) ? r - 4 : (r < 20) ? r - 8 : r - 16);
    pr = (ga ? 0 : 16) + k;
  }
  const int p = wave * 32 + pr;
  int ft = 0, py = 0, px = 0, arow = p;
  if constexpr (TAPS == 9) {
    ft = p / (P::PH * P::PW); py = (p / P::PW) % P::PH; px = p % P::PW;
    arow = (ft * P::HH + py) * P::HW + px;
  }

  f32x16 acc[S][NT];
  f32x16 accc[NT];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[s][n][i] = 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) accc[n][i] = 0.f;

  const bf16* xg = (const bf16*)a.x;
  const bf16* cg = (const bf16*)a.ctx;
  const int nchunk = (Cin + CK - 1) / CK;
  constexpr int NPH = HAS_CTX ? 3 : 1;

  // Software pipeline: the global loads of phase i+1 (activation halo + weight slab, 16 B per lane) are issued into
  // registers BEFORE the MFMA section of phase i and written to LDS after it, so HBM/L2 latency hides under the
  // matrix work of the same workgroup (T14 "issue early / write late").
  constexpr int TOTA = Cfg::AROWS * PARTS, TOTC = Cfg::CROWS * PARTS, TOTW = Cfg::WROWS * PARTS;
  constexpr int NIA = (TOTA + NTHR - 1) / NTHR, NIW = (TOTW + NTHR - 1) / NTHR;
  u32x4 ra[NIA], rw[NIW];
  const unsigned short fb = __builtin_bit_cast(unsigned short, f2bf(a.ctx_fill));
  const unsigned fill2 = (unsigned)fb | ((unsigned)fb << 16);

  // per-thread load descriptors, computed ONCE (the halo decode needs divisions by (PW+2), (PH+2)):
  //   bit 31 valid | bit 30 slot | bits 25..29 frame-in-tile | bits 0..24 (y*W+x)*Cin + part*8
  unsigned adesc[NIA];
  if constexpr (TAPS == 9) {
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int e = tid + i * NTHR;
      adesc[i] = 0u;
      if (e < TOTA) {
        const int row = e / PARTS, part = e % PARTS;
        const int s = row / P::HALO, hr = row % P::HALO;
        const int f_ = hr / (P::HH * P::HW), rem = hr % (P::HH * P::HW);
        const int y = y0 + rem / P::HW - 1, x = x0 + rem % P::HW - 1;
        if (t0 + f_ < T && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
          adesc[i] = 0x80000000u | ((unsigned)s << 30) | ((unsigned)f_ << 25) | (unsigned)((y * W + x) * Cin + part * 8);
      }
    }
  }
  const int part8 = (tid % PARTS) * 8;
  const int frame_elems = HWp * Cin;
  const bf16* xb = xg + (size_t)b * S * T * frame_elems;                       // this batch element's frames
  const bf16* cb_ = HAS_CTX ? cg + (size_t)b * a.ctx_bstride * frame_elems : nullptr;
  const int wrow0 = tid / PARTS;

  auto load_phase = [&](int ch, int ph) __attribute__((always_inline)) {
    const int c0 = ch * CK;
    const bool cok = (c0 + part8) < Cin;
    if constexpr (TAPS == 9) {
      if (ph == 0) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          const unsigned dsc = adesc[i];
          ra[i] = u32x4{0u, 0u, 0u, 0u};
          if ((dsc >> 31) && cok) {
            const int fr = (int)((dsc >> 30) & 1u) * T + t0 + (int)((dsc >> 25) & 31u);
            ra[i] = *(const u32x4*)(xb + (fr * frame_elems + (int)(dsc & 0x1ffffffu) + c0));
          }
        }
      } else {
        const int coff = (ph == 1) ? a.coff0 : a.coff1;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          const unsigned dsc = adesc[i];
          ra[i] = u32x4{0u, 0u, 0u, 0u};
          if (tid + i * NTHR < TOTC && (dsc >> 31) && cok) {     // rows < HALO are the slot-0 rows of the own image
            const int f = t0 + (int)((dsc >> 25) & 31u) + coff;
            if (f >= 0 && f < a.ctx_T) ra[i] = *(const u32x4*)(cb_ + (f * frame_elems + (int)(dsc & 0x1ffffffu) + c0));
            else ra[i] = u32x4{fill2, fill2, fill2, fill2};
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const int e = tid + i * NTHR;
        ra[i] = u32x4{0u, 0u, 0u, 0u};
        if (e < TOTA) {
          const int row = e / PARTS;
          const int s = row / NPOS, q = q0 + (row % NPOS);
          if (q < T * HWp && cok) {
            const size_t pos = (size_t)(b * S + s) * T * HWp + q;
            if (a.x2 == nullptr) {
              ra[i] = *(const u32x4*)(xg + pos * Cin + c0 + part8);
            } else {
              // two-source input (OnirisConvArgs.x2): the decoder's mp_cat; the raw piece is loaded here, scaled / rounded /
              // activated when it is written to LDS (store_phase): the load stays in flight under the MFMAs of the phase before
              const int c = c0 + part8, C1 = a.x_split;
              ra[i] = (c < C1) ? *(const u32x4*)(xg + pos * C1 + c)
                               : *(const u32x4*)((const bf16*)a.x2 + pos * (Cin - C1) + (c - C1));
            }
          }
        }
      }
    }
    const bf16* wg = ((ph == 0) ? (const bf16*)a.w_own
                                : (const bf16*)a.w_ctx + (size_t)(ph - 1) * TAPS * a.CoutP * a.CinP) +
                     (size_t)co0 * a.CinP + c0 + part8;
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int row = wrow0 + i * (NTHR / PARTS);
      if (tid + i * NTHR < TOTW) {
        const int tap = row / BN, co = row % BN;
        rw[i] = *(const u32x4*)(wg + (tap * a.CoutP + co) * a.CinP);
      }
    }
  };
  auto store_phase = [&](int ph, int ch) __attribute__((always_inline)) {
    const int tot = (ph == 0) ? TOTA : TOTC;
    if constexpr (TAPS == 1) {
      if (a.x2 != nullptr) {
        // xo = bf16(w * source) is what the MFMAs consume; its mp_silu goes out to act_out from the workgroups of the first
        // output-channel block (the rounding conventions of oniris_act_fwd: the activation sees the ROUNDED xo)
        const int c = ch * CK + part8;
        const float w = (c < a.x_split) ? a.cat_w1 : a.cat_w2;
        // the activation of channel round `ch` is written by ONE of the tile's output-channel blocks, the rounds dealt round-robin:
        // 32 sigmoids per lane and round are ~1 us, and with all of them on the first block the launch lasted as long as that
        // block's 8 rounds (15 us at 512 input channels against 6.4 us for a plain 256-channel launch; round 6)
        const bool act_mine = (ch % d.ncob) == (co0 / BN);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          const int e = tid + i * NTHR;
          if (e < TOTA) {
            const int row = e / PARTS;
            const int s = row / NPOS, q = q0 + (row % NPOS);
            const bf16x8 in = __builtin_bit_cast(bf16x8, ra[i]);
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(in[k]) * w);
            ra[i] = __builtin_bit_cast(u32x4, o);
            if (act_mine && a.act_out && q < T * HWp && c < Cin) {
              bf16x8 av;
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float z = bf2f(o[k]);
                av[k] = f2bf(z * sigmoid_fast(z) * (1.0f / 0.596f));
              }
              *(bf16x8*)((bf16*)a.act_out + ((size_t)(b * S + s) * T * HWp + q) * Cin + c) = av;
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int e = tid + i * NTHR;
      if (e < tot) *(u32x4*)(A_lds + (e / PARTS) * ROWB + (e % PARTS) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int e = tid + i * NTHR;
      if (e < TOTW) *(u32x4*)(W_lds + (e / PARTS) * ROWB + (e % PARTS) * 16) = rw[i];
    }
  };

  auto mfma_steps = [&](auto own_) __attribute__((always_inline)) {
    constexpr bool OWN = decltype(own_)::value;
    constexpr int NX = OWN ? S : 1;
    constexpr int KS = CK / 16, NSTEP = TAPS * KS;
    bf16x8 wf[2][NT], xf[2][NX];
    auto ld = [&](int buf, int st) __attribute__((always_inline)) {
      const int tap = st / KS, ks = st % KS;
      const int off = (TAPS == 9) ? ((tap / 3) * P::HW + (tap % 3)) : 0;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        wf[buf][n] = *(const bf16x8*)(W_lds + (tap * BN + n * 32 + r) * ROWB + ks * 32 + h * 16);
#pragma unroll
      for (int s = 0; s < NX; ++s) {
        const int srow = (TAPS == 9) ? s * P::HALO : s * NPOS;
        xf[buf][s] = *(const bf16x8*)(A_lds + (srow + arow + off) * ROWB + ks * 32 + h * 16);
      }
    };
    ld(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      if (st + 1 < NSTEP) ld((st + 1) & 1, st + 1);
      if constexpr (OWN) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[s][n] = mfma32(wf[st & 1][n], xf[st & 1][s], acc[s][n]);
      } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) accc[n] = mfma32(wf[st & 1][n], xf[st & 1][0], accc[n]);
      }
    }
  };

  const int nphase_all = nchunk * NPH;
  const int it0 = (ks > 1) ? (int)((long long)nphase_all * kz / ks) : 0;
  const int nphase = red ? 0 : (ks > 1) ? (int)((long long)nphase_all * (kz + 1) / ks) : nphase_all;
  if (it0 < nphase) load_phase(it0 / NPH, it0 % NPH);
#pragma unroll 1
  for (int itp = it0; itp < nphase; ++itp) {
    const int ph = itp % NPH;
    store_phase(ph, itp / NPH);
    __syncthreads();
    if (itp + 1 < nphase) load_phase((itp + 1) / NPH, (itp + 1) % NPH);
    // ------------------------------------------------------------------ MFMA over taps x k-steps
    // The own / context phases are separate straight-line sequences (a branch per step would pin every LDS read
    // right in front of its MFMA); fragments are double-buffered in registers: the reads of step i+1 are in flight
    // while the MFMAs of step i run.
    if (ph == 0) mfma_steps(std::true_type{});
    else mfma_steps(std::false_type{});
    __syncthreads();
  }

  // -------------------------------------------------------------------- split-K: partial sums meet in the workspace
  // Tiny grids (one generated frame in the rollout: 4..32 tiles) leave the chip idle while every workgroup walks
  // its whole K = 27*Cin serially at HBM latency.  With ksplit > 1 each slice writes its accumulators to
  // splitk_ws[tile][slice] and exits; a second launch of this kernel (d.reduce, one 64-thread block per wave of a
  // tile: the reduction is latency/bandwidth-bound per block, so it is spread as wide as the epilogue allows) adds
  // the slices in order (deterministic) and runs the epilogue.  The kernel boundary is the only synchronisation:
  // an in-kernel "last slice reduces" needs agent-scope release/acquire = L2 writeback + invalidate per workgroup
  // on a multi-XCD part, measured slower than not splitting at all.
  if (ks > 1) {
    constexpr int NACC = (S + (HAS_CTX ? 1 : 0)) * NT;
    float* wsp = a.splitk_ws + ((size_t)tile_id * ks) * (size_t)(NACC * NTHR * 16);
    if (!red) {
      float* mine = wsp + (size_t)kz * (NACC * NTHR * 16) + tid * 16;
      auto put = [&](int j, const f32x16& v) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(float4*)(mine + (size_t)j * NTHR * 16 + q * 4) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      };
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) put(s * NT + n, acc[s][n]);
      if constexpr (HAS_CTX) {
#pragma unroll
        for (int n = 0; n < NT; ++n) put(S * NT + n, accc[n]);
      }
      return;
    }
    auto get = [&](int j, f32x16& v) __attribute__((always_inline)) {
      const float* src = wsp + (size_t)j * NTHR * 16 + tid * 16;
      for (int z0 = 0; z0 < ks; z0 += 4) {                  // 16 loads of 16 B in flight per lane
        float4 t[4][4];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            t[zz][q] = (z0 + zz < ks) ? *(const float4*)(src + (size_t)(z0 + zz) * (NACC * NTHR * 16) + q * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[4 * q] += t[zz][q].x; v[4 * q + 1] += t[zz][q].y; v[4 * q + 2] += t[zz][q].z; v[4 * q + 3] += t[zz][q].w;
          }
      }
    };
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int n = 0; n < NT; ++n) get(s * NT + n, acc[s][n]);
    if constexpr (HAS_CTX) {
#pragma unroll
      for (int n = 0; n < NT; ++n) get(S * NT + n, accc[n]);
    }
  }

  // -------------------------------------------------------------------- epilogue
  // The math is lane-local (lane = position): gate combine, emb-scale+SiLU, mp_sum+clip.  The bf16 results are then
  // transposed through a wave-private LDS tile ([32 positions][BN channels]) so that global stores are 16 B per lane
  // and cover whole 128-byte channel rows (8-byte scattered stores cost ~10x write traffic: measured WRITE_SIZE).
  bool valid;
  size_t pix;       // pixel index inside the (b,s) block of T frames
  int tloc;
  if constexpr (TAPS == 9) {
    tloc = t0 + ft;
    valid = tloc < T;
    pix = (size_t)tloc * HWp + (y0 + py) * W + (x0 + px);
  } else {
    const int q = q0 + p;
    valid = q < T * HWp;
    tloc = (valid && q >= HWp) ? q / HWp : 0;
    pix = (size_t)q;
  }
  constexpr int EROW = BN * 2 + 16;
  unsigned char* ep = smem + wave * 32 * EROW;
  // coalesced write-out of the wave's LDS tile to tensor `dst` ((b,s)-block element offset `blk`)
  auto flush = [&](bf16* dst, size_t blk) __attribute__((always_inline)) {
    constexpr int PO = BN / 8;
#pragma unroll
    for (int it = 0; it < 32 * PO / 64; ++it) {
      const int id = it * 64 + lane;
      const int row = id / PO, part = id % PO;
      const int pr = wave * 32 + row;
      bool ok;
      size_t px_;
      if constexpr (TAPS == 9) {
        const int f_ = pr / (P::PH * P::PW), yy = (pr / P::PW) % P::PH, xx = pr % P::PW;
        ok = (t0 + f_) < T;
        px_ = (size_t)(t0 + f_) * HWp + (y0 + yy) * W + (x0 + xx);
      } else {
        ok = (q0 + pr) < T * HWp;
        px_ = (size_t)(q0 + pr);
      }
      const int co = co0 + part * 8;
      if (ok && co < a.Cout) *(uint4*)(dst + (blk + px_) * a.Cout + co) = *(const uint4*)(ep + row * EROW + part * 16);
    }
  };
  bf16* og = (bf16*)a.out;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int n = (b * S + s) * T + tloc;
    const float cown = (a.coef_own && valid) ? a.coef_own[n] : 1.f;
    float cctx = 0.f;
    if constexpr (HAS_CTX) cctx = (a.coef_ctx && valid) ? a.coef_ctx[n] : 1.f;
    const size_t blk = (size_t)(b * S + s) * T * HWp;
    const size_t obase = (blk + pix) * a.Cout;
    float v[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[nt][i] = cown * acc[s][nt][i];
        if constexpr (HAS_CTX) v[nt][i] = __builtin_fmaf(cctx, accc[nt][i], v[nt][i]);   // explicit: same rounding in every variant
      }
    // helper: write v (as bf16) into the wave tile
    auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = f2bf(v[nt][4 * g + k]);
          *(bf16x4*)(ep + pr * EROW + (nt * 32 + 8 * g + 4 * h) * 2) = o;
        }
    };
    if (a.epi == ONIRIS_EPI_MPSUM) {
      if (a.out2) { stage(); flush((bf16*)a.out2, blk); }       // raw conv output (needed for d gate)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + nt * 32 + 8 * g + 4 * h;
          bf16x4 rv;
#pragma unroll
          for (int k = 0; k < 4; ++k) rv[k] = f2bf(0.f);
          if (valid && co < a.Cout) rv = *(const bf16x4*)((const bf16*)a.res + obase + co);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float o = a.ta * bf2f(rv[k]) + a.tb * v[nt][4 * g + k];
            if (a.clip > 0.f) o = fminf(fmaxf(o, -a.clip), a.clip);
            v[nt][4 * g + k] = o;
          }
        }
    }
    stage();
    flush(og, blk);
    if (a.epi == ONIRIS_EPI_EMB_SILU) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + nt * 32 + 8 * g + 4 * h;
          float4 ev = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid && co < a.Cout) ev = *(const float4*)((const float*)a.escale + (size_t)n * (a.escale_pitch ? a.escale_pitch : a.Cout) + co);
          const float cvv[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float z = bf2f(f2bf(v[nt][4 * g + k])) * cvv[k];     // the activation sees the bf16-rounded y
            v[nt][4 * g + k] = z * sigmoid_fast(z) * (1.f / 0.596f);
          }
        }
      stage();
      flush((bf16*)a.out2, blk);
    }
    if constexpr (HAS_CTX) {
      if (a.ctx_out && s == 0) {          // unscaled context product y3 (shared by both slots), kept for d(gate)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[nt][i] = accc[nt][i];
        stage();
        flush((bf16*)a.ctx_out, (size_t)b * T * HWp);
      }
    }
  }
}

template <int S, int TAPS, int CK, int NT, bool HAS_CTX, int PW, int NW = 4>
static int launch_conv_fwd(const OnirisConvArgs& a, hipStream_t stream) {
  using Cfg = ConvCfg<S, TAPS, CK, NT, HAS_CTX, PW, NW>;
  using P = typename Cfg::P;
  ConvDev d;
  d.a = a;
  d.ncob = a.CoutP / Cfg::BN;
  if (TAPS == 9) {
    d.ntx = a.W / P::PW; d.nty = a.H / P::PH; d.ntt = cdiv(a.T, P::FT);
  } else {
    d.ntx = 1; d.nty = 1; d.ntt = cdiv(a.T * a.H * a.W, Cfg::NPOS);
  }
  long long nblk = (long long)d.ntx * d.nty * d.ntt * a.B * d.ncob;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { oniris_set_error("conv_fwd: bad grid %lld", nblk); return ONIRIS_EINVAL; }
  // split-K when the caller lent a workspace and the tiles alone leave most of the chip idle
  d.ksplit = 1; d.reduce = 0; d.grid3d = 0;
  const long long ntile = nblk;
  if (a.splitk_ws && nblk <= 64) {
    const int nphase = cdiv(a.Cin, CK) * (HAS_CTX ? 3 : 1);
    constexpr size_t per = (size_t)(S + (HAS_CTX ? 1 : 0)) * NT * Cfg::NTHR * 16 * sizeof(float);
    static const int ks_cap = getenv("ONIRIS_KSCAP") ? atoi(getenv("ONIRIS_KSCAP")) : 12;
    long long ks = 256 / nblk;
    if (ks > nphase) ks = nphase;
    if (ks > ks_cap) ks = ks_cap;
    // Cin = 32: three rounds, the second launch costs more than it saves; 1x1 convs: a round is one 64-channel MFMA
    // sweep (~1 us), and in the sampler's graphs a second node costs ~5 us by itself: split from Cin = 768 on
    if (nphase < (TAPS == 1 ? 12 : 6)) ks = 1;
    if ((size_t)nblk * ks * per > a.splitk_ws_bytes) ks = (long long)(a.splitk_ws_bytes / (nblk * per));
    if (ks > 1) { d.ksplit = (int)ks; nblk *= ks; }
  }
  auto kern = conv_fwd_kernel<S, TAPS, CK, NT, HAS_CTX, PW, NW>;
  if (TAPS == 1 && d.ksplit == 1 && a.B <= 65535 && d.ncob <= 65535) {
    d.grid3d = 1;
    ONIRIS_KLAUNCH(kern, dim3((unsigned)d.ntt, (unsigned)a.B, (unsigned)d.ncob), dim3(Cfg::NTHR), 0, stream, d);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  ONIRIS_KLAUNCH(kern, dim3((unsigned)nblk), dim3(Cfg::NTHR), 0, stream, d);
  if (d.ksplit > 1) {
    ONIRIS_LAUNCH_CHECK();
    d.reduce = 1;
    ONIRIS_KLAUNCH(kern, dim3((unsigned)(ntile * NW)), dim3(64), 0, stream, d);
  }
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
