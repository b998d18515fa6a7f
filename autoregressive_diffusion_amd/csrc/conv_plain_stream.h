// Plain 3x3 convolution, 32 -> <= 32 channels (the 64x64-pixel level of the UNets in the 2-D training steps -- just_2d,
// edm2/conv.py:60 -- and the data gradients of those layers): STREAMING variant of conv_glds_kernel<NT=1, CTX=false>.
//
// At 32 channels a 16x16-pixel tile of the tile kernel is ONE phase (one 32-channel chunk, no context phases): 42 KB of
// LDS-DMA (both frames' halos + the weight slab, re-copied for every tile) in front of 36 MFMAs per wave and an epilogue
// that stops all eight waves -- 0.44 of its HBM roofline.  The level is HBM-bound by a factor of three (0.5-0.8 GB per
// launch against 31 us of MFMA work at B = 8), so what matters is bytes in flight and that nothing waits for anything else.
// Here (the forward half of conv_stream.h without its context ring):
//   * a workgroup (4 waves, an 8x16-pixel tile, TWO per CU: they share no barrier and drift apart) owns one spatial tile
//     and WALKS the frames of a segment; per frame it copies ONE halo image (10 x 18 pixels, 11.25 KB) into a four-slot
//     ring, three frames ahead (counted vmcnt: the stores of the previous epilogues drain under the next frames);
//   * the weights (9 taps x 32 x 32) live in REGISTERS for the whole walk, 18 fragments per wave; wave w computes pixel rows
//     2w, 2w + 1 of the tile: 18 MFMAs per frame in two independent chains, one LDS fragment read per MFMA, no exchange
//     between the waves;
//   * epilogues as conv_stream.h / conv_glds.h (none | emb-scale + SiLU | mp_sum + clip with the clip report), lane =
//     position, bf16 results transposed through the wave's own LDS tile, 16-byte stores (non-temporal on big tensors).
// Same operand rounding (bf16 operands, fp32 accumulation); the summation order over (tap, k) is the tile kernel's with the
// two k-steps of a tap in separate chains, joined at the end.
#pragma once
#include "conv_kernels.h"
#include "lds_dma.h"

struct ConvPlainStreamDev {
  OnirisConvArgs a;
  int ntx, nty, nseg, seglen, nfr;     // nfr = B * S * T frames
  int nt;                              // non-temporal output stores (see conv_stream.h)
};

__global__ __launch_bounds__(256, 2) void conv_plain_stream_kernel(const ConvPlainStreamDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TAPS = 9, KS = 2, NST = TAPS * KS, HW_ = 18, HALO = 10 * HW_, HBUF = 12288, NRING = 4, DEPTH = 3;
  constexpr int EROW = 80, STGB = 32 * EROW;
  constexpr int STG = NRING * HBUF, ESC = STG + 4 * STGB, LDS_BYTES = ESC + NRING * 128;     // emb-scale rows [ring slot][32] fp32
  static_assert(HALO * 64 <= HBUF && 2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const OnirisConvArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int H = a.H, W = a.W, HWp = H * W, Cout = a.Cout;
  constexpr int Cin = 32;
  const int frame_elems = HWp * Cin;

  // ---- this workgroup's (tile, segment); workgroup ids go round-robin over the XCDs: XCD k takes a contiguous range of units
  int u;
  {
    const int n = gridDim.x, xcd = blockIdx.x & 7, q = n >> 3, rr = n & 7;
    u = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blockIdx.x >> 3);
  }
  const int x0 = (u % d.ntx) * 16; u /= d.ntx;
  const int y0 = (u % d.nty) * 8;
  const int seg = u / d.nty;
  const int n_lo = seg * d.seglen, n_hi = min(d.nfr, n_lo + d.seglen), nfr = n_hi - n_lo;
  if (nfr <= 0) return;

  // ---- lane -> position inside the wave's 32-position tile (2 pixel rows x 16; conv_kernels.h: 16-lane read groups take 16
  // consecutive halo rows)
  int pr;
  {
    const bool ga = (r < 4) || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int k = ga ? ((r < 4) ? r : (r < 16) ? r - 8 : r - 12) : ((r < 12) ? r - 4 : (r < 20) ? r - 8 : r - 16);
    pr = (ga ? 0 : 16) + k;
  }
  int xaddr[TAPS];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int R = (2 * uw + (pr >> 4) + tap / 3) * HW_ + (pr & 15) + tap % 3;
    xaddr[tap] = R * 64 + ((h ^ ((R >> 2) & 3)) << 4);                   // k-step 1: ^ 32
  }

  // ---- DMA descriptors: 720 pieces of 16 B per halo image, three per thread
  constexpr int OOB = (int)0x80000000;
  int hv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = i * 256 + tid, row = e >> 2, gp = (e & 3) ^ ((row >> 2) & 3);
    const int y = y0 + row / HW_ - 1, x = x0 + row % HW_ - 1;
    hv[i] = (e < HALO * 4 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? ((y * W + x) * Cin + gp * 8) * 2 : OOB;
  }
  const i32x4 rs_x = make_rsrc(a.x, d.nfr * frame_elems * 2);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem, wdst = lds0 + wave * 1024;
  const bool emb = a.epi == ONIRIS_EPI_EMB_SILU, mps = a.epi == ONIRIS_EPI_MPSUM;
  const int epitch = a.escale_pitch ? a.escale_pitch : Cout;
  const i32x4 rs_e = make_rsrc(a.escale ? a.escale : (const void*)oniris_fill_rows, emb ? (int)(((size_t)d.nfr - 1) * epitch + Cout) * 4 : 0);
  const int evoff = (lane < 8 && lane * 4 < Cout) ? lane * 16 : OOB;    // piece `lane` of the frame's row
  auto issue = [&](int f) __attribute__((always_inline)) {              // halo of frame f (+ its emb-scale row: wave 0)
    const int so = f * frame_elems * 2;
    const unsigned dst = (f & (NRING - 1)) * HBUF + wdst;
    dma16(rs_x, hv[0], so, dst);
    dma16(rs_x, hv[1], so, dst + 4096);
    if (512 + tid < HALO * 4) dma16(rs_x, hv[2], so, dst + 8192);
    if (emb && uw == 0 && lane < 8) dma16(rs_e, evoff, (f * epitch) * 4, lds0 + ESC + (f & (NRING - 1)) * 128);
  };
  // instruction counts of this wave (wave-uniform): copies per frame, residual loads, stores per epilogue
  const int n_dma = 3 + ((emb && uw == 0) ? 1 : 0);                      // (the third halo piece ends inside wave 3: 720 - 512 = 208 lanes)
  const int n_res = mps ? 4 : 0;
  const int n_st = 2 * (mps ? ((a.out2 ? 1 : 0) + 1) : emb ? 2 : 1);

  // ---- weights -> registers (once): lane (r = co row, h = 8-channel group of the k-step)
  bf16x8 wreg[NST];
  {
    const bf16* wsrc = (const bf16*)a.w_own;
#pragma unroll
    for (int i = 0; i < NST; ++i)
      wreg[i] = *(const bf16x8*)(wsrc + ((size_t)(i / KS) * a.CoutP + r) * a.CinP + (i % KS) * 16 + h * 8);
  }
#pragma unroll
  for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(wreg[i]));        // consumed before any LDS-DMA is in flight

  // ---- epilogue inputs (see conv_stream.h: nothing in the frame loop is an ordinary vector load)
  const int ppy = 2 * uw + (pr >> 4), ppx = pr & 15;                     // pixel of this lane's output position inside the 8x16 tile
  const int pix = (y0 + ppy) * W + x0 + ppx;
  const i32x4 rs_r = make_rsrc(mps ? a.res : (const void*)oniris_fill_rows, mps ? d.nfr * HWp * Cout * 2 : 0);
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 resq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { resq[g][0] = 0u; resq[g][1] = 0u; }
  int rvoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rvoff[g] = (8 * g + 4 * h < Cout) ? (pix * Cout + 8 * g + 4 * h) * 2 : OOB;
  auto load_res = [&](int f) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane((f * HWp * Cout) * 2);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(resq[g]) : "v"(rvoff[g]), "s"(rs_r), "s"(so) : "memory");
  };
  auto wait_vm = [&](int n) __attribute__((always_inline)) {             // s_waitcnt vmcnt(n), n wave-uniform
    if (n > 40) n = 40;                                                   // (waiting for more than asked is always correct)
    switch (n) {
#define PSW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
      PSW(0) PSW(1) PSW(2) PSW(3) PSW(4) PSW(5) PSW(6) PSW(7) PSW(8) PSW(9) PSW(10) PSW(11) PSW(12) PSW(13) PSW(14) PSW(15) PSW(16) PSW(17)
      PSW(18) PSW(19) PSW(20) PSW(21) PSW(22) PSW(23) PSW(24) PSW(25) PSW(26) PSW(27) PSW(28) PSW(29) PSW(30) PSW(31) PSW(32) PSW(33)
      PSW(34) PSW(35) PSW(36) PSW(37) PSW(38) PSW(39)
      default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
#undef PSW
    }
  };

  // ---- prologue copies: the first DEPTH frames of the segment
#pragma unroll 1
  for (int i = 0; i < DEPTH && i < nfr; ++i) issue(n_lo + i);
  unsigned char* const ep = smem + STG + wave * STGB;                    // this wave's staging tile

#pragma unroll 1
  for (int i = 0; i < nfr; ++i) {
    const int f = n_lo + i;
    // this wave's copies of frame f have landed.  Younger instructions of this wave that may still be in flight: the copies of
    // the frames issued after f's (up to DEPTH - 1 of them, each issued in a step behind that step's residual loads and in
    // front of its stores) and the stores / residual loads of the steps in between
    {
      const int ahead = min(DEPTH - 1, nfr - 1 - i);                     // frames f+1 .. f+ahead are requested
      // steps i-DEPTH+1 .. i-1 issued: [res][dma (if any)][stores]; the copies of frame f+j went out in step i+j-DEPTH (or the prologue)
      int younger = 0;
      for (int j = 1; j <= ahead; ++j) younger += n_dma;
      const int steps_behind = min(i, DEPTH - 1);                         // full steps issued after frame f's copies ... at most
      younger += steps_behind * (n_res + n_st);
      if (i >= DEPTH) younger += n_st;                                    // the stores of step i-DEPTH, issued right after f's copies
      wait_vm(younger);
    }
    __syncthreads();                       // ... everybody's have; and everybody is done with frame f - 1 (its ring slot is free)
    typedef const __attribute__((address_space(4))) float cfloat_t;
    const float cown = a.coef_own ? ((cfloat_t*)(size_t)a.coef_own)[f] : 1.f;
    if (mps) load_res(f);                  // (in front of the copies: waited for by count, see the epilogue)
    const bool has_next = i + DEPTH < nfr;
    if (has_next) issue(f + DEPTH);

    f32x16 acc0, acc1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
    {
      const unsigned char* bA = smem + (f & (NRING - 1)) * HBUF;
      bf16x8 xf[2];
      xf[0] = *(const bf16x8*)(bA + xaddr[0]);
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        if (st + 1 < NST) {
          const int tap = (st + 1) / KS, ks = (st + 1) % KS;
          xf[(st + 1) & 1] = *(const bf16x8*)(bA + (xaddr[tap] ^ (ks * 32)));
        }
        if (st & 1) acc1 = mfma32(wreg[st], xf[st & 1], acc1);
        else acc0 = mfma32(wreg[st], xf[st & 1], acc0);
        if (st + 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = cown * (acc0[k] + acc1[k]);

    // ---- epilogue (lane = position; bf16 results transposed through the wave's LDS tile)
    const size_t blk = (size_t)f * HWp;
    auto put = [&](const float (&vv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f2bf(vv[4 * g + k]);
        *(bf16x4*)(ep + pr * EROW + (8 * g + 4 * h) * 2) = o;
      }
    };
    auto flush = [&](bf16* dst) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int id = it * 64 + lane, row = id >> 2, part = id & 3;
        const size_t px_ = (size_t)(y0 + 2 * uw + (row >> 4)) * W + x0 + (row & 15);
        if (part * 8 < Cout) {
          const u32x4 v_ = *(const u32x4*)(ep + row * EROW + part * 16);
          u32x4* o_ = (u32x4*)(dst + (blk + px_) * Cout + part * 8);
          if (d.nt) __builtin_nontemporal_store(v_, o_); else *o_ = v_;
        }
      }
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
      if (a.out2) { put(v); flush((bf16*)a.out2); }
      put(o);
      flush((bf16*)a.out);
      if (a.clip_flag && __builtin_amdgcn_ballot_w64(clip_hit) != 0ull) {    // (practically never: OnirisConvArgs.clip_flag)
        if (lane == 0) atomicOr(a.clip_flag, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // one more op in the vmcnt stream than the counted waits know
      }
      continue;
    }
    put(v);
    flush((bf16*)a.out);
    if (emb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ev = *(const float4*)(smem + ESC + (f & (NRING - 1)) * 128 + (8 * g + 4 * h) * 4);
        const float cvv[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float z = bf2f(f2bf(v[4 * g + k])) * cvv[k];     // the activation sees the bf16-rounded y
          v[4 * g + k] = z * sigmoid_fast(z) * (1.f / 0.596f);
        }
      }
      put(v);
      flush((bf16*)a.out2);
    }
  }
#endif
}

// Shapes the plain streaming kernel takes: S == 1, no context path, 9 taps, exactly 32 input channels, at most 32 output
// channels, images of 8-row x 16-column tiles, the epilogues of the 3x3 layers.
static inline bool conv_plain_stream_ok(const OnirisConvArgs& a) {
  if (!(a.S == 1 && !a.ctx && a.taps == 9 && a.Cin == 32 && a.CinP == 64 && a.CoutP == 32 && a.W % 16 == 0 && a.H % 8 == 0)) return false;
  if (a.ctx_out || a.x2 || a.coef_ctx || a.ctx_prod_mode != 0) return false;
  if (!(a.epi == ONIRIS_EPI_NONE || a.epi == ONIRIS_EPI_EMB_SILU || a.epi == ONIRIS_EPI_MPSUM)) return false;
  if (a.escale_pitch != 0 && a.escale_pitch % 4 != 0) return false;
  const long long nfr = (long long)a.B * a.T;
  if (nfr * a.H * a.W * 32 * 2 >= (1LL << 31) || nfr * (a.escale_pitch ? a.escale_pitch : a.Cout) * 4 >= (1LL << 31)) return false;
  return nfr * (a.H / 8) * (a.W / 16) >= 512;          // (small launches: the tile kernel's single wave of workgroups is as good)
}

static int launch_conv_plain_stream(const OnirisConvArgs& a, hipStream_t stream) {
  ConvPlainStreamDev d;
  d.a = a;
  d.ntx = a.W / 16; d.nty = a.H / 8;
  d.nfr = a.B * a.T;
  const int tiles = d.ntx * d.nty;
  int nseg = 512 / tiles;                  // two workgroups per CU
  if (nseg > d.nfr / 4) nseg = d.nfr / 4;  // (a segment pays its pipeline fill: three frames)
  if (nseg < 1) nseg = 1;
  d.seglen = cdiv(d.nfr, nseg);
  d.nseg = cdiv(d.nfr, d.seglen);
  d.nt = (long long)d.nfr * a.H * a.W * a.Cout * 2 >= oniris_ew_nt_bytes();
  oniris_launch_tagged(d.nt ? "nt-stores" : nullptr, conv_plain_stream_kernel, dim3(tiles * d.nseg), dim3(256), stream, d);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
