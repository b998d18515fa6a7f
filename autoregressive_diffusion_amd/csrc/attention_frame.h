// Dense attention INSIDE frames of P = 64 / 128 / 256 tokens, head width 64: FrameAttention (reference
// edm2/attention/attention_modules.py:105-119) and VideoAttention's just_2d branch (:36-45) -- 16x16 latents in the Lunar-Lander
// net, 8x8 in the Counter-Strike net.
//
// Why kernels of their own (round 6).  The grid kernels (attention.hip) give a workgroup 128 query rows and stream the keys in
// 64-row tiles behind one barrier + one LDS-DMA wait each; with 256 keys that is four waits in front of 16 MFMAs per wave each,
// two workgroups per (frame, head) that both copy all of K and V: 0.14-0.19 of the bf16 MFMA peak (profiles/r05_conv_shapes.txt).
// The persistent VideoAttention kernels on a block-diagonal table were measured too (profiles/r06_ab_frame_ws.txt): slower.
// Here a workgroup owns a SUPER-BLOCK of 256 consecutive tokens of one head -- 256 / P whole frames -- whose K | V (forward, dQ)
// or Q | dO (dK / dV) images, 64 KB, land in LDS by ONE burst of LDS-DMA behind ONE barrier; wave w owns rows 64 w .. 64 w + 63
// (two 32-row MFMA blocks that share every fragment read from LDS) and walks the P / 64 tiles of ITS frame.  No mask exists
// (everything inside a frame is allowed), so the hot loop is the unmasked tile body of the grid kernels.
//
// Operand layouts, swizzles and the no-running-maximum softmax are those of attention.hip (helpers reused); q arrives carrying
// log2(e) / 8 (qkv_norm_kernel), lse is log2-domain.
#pragma once

struct FrameAttnDev {
  OnirisAttnArgs a;
  long long ntok;          // B * P tokens per head
  int tiles;               // P / 64: key tiles per frame
};

// one burst: tile t (64 rows x 64 channels of head `head`) of tensors g0 / g1 -> LDS [t][g0 | g1], swizzled on the source side:
// SW0 / SW1 pick the piece permutation of attention.hip's images -- 0: (row >> 1) & 7 (row reads), 1: 4 * bit1(row) (transposing
// reads; row reads of it are 4-way conflicted), 2: bit1 << 2 | bit3 << 1 | bit2 (the dual-use image: conflict-free for both)
template <int SW0, int SW1>
__device__ __forceinline__ void frame_burst(const bf16* g0, const bf16* g1, long long tok0, long long ntok, int C /* row pitch */, int head,
                                            unsigned lds0, int tid) {
  constexpr int TB = 64 * 128, OOB = (int)0x80000000;
  const int wave = tid >> 6;
  const long long left = ntok - tok0;
  const int rows = left >= 256 ? 256 : (int)left;
  const i32x4 r0 = make_rsrc(g0 + (size_t)tok0 * C, rows * C * 2);
  const i32x4 r1 = make_rsrc(g1 + (size_t)tok0 * C, rows * C * 2);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = i * 256 + tid, row = e >> 3, pp = e & 7;
      const int dual = (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
      const int sw0 = SW0 == 0 ? ((row >> 1) & 7) : SW0 == 1 ? 4 * ((row >> 1) & 1) : dual;
      const int sw1 = SW1 == 0 ? ((row >> 1) & 7) : SW1 == 1 ? 4 * ((row >> 1) & 1) : dual;
      const bool ok = t * 64 + row < rows;
      const unsigned dst = lds0 + t * 2 * TB + i * 4096 + wave * 1024;
      dma16(r0, ok ? (row * C + head * 64 + (pp ^ sw0) * 8) * 2 : OOB, t * 64 * C * 2, dst);
      dma16(r1, ok ? (row * C + head * 64 + (pp ^ sw1) * 8) * 2 : OOB, t * 64 * C * 2, dst + TB);
    }
  }
}

// J = 32-row query blocks per wave: 2 = a workgroup per 256 tokens (training); 1 = a workgroup per 128 tokens of a 256-token frame
// (round 6: one frame per sequence in the cached sampler is heads workgroups -- two query halves per frame put the launch on twice the
// CUs with half the MFMA / softmax chain each; K | V of the frame are staged by both)
template <int J>
__global__ __launch_bounds__(256, 2) void frame_attn_fwd_kernel(const FrameAttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 2 * TB];     // [tile][K | V]
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, C = a.C, P = a.Lq;
  const long long tok0 = (long long)(blockIdx.x / (2 / J)) * 256;            // the 256-token block whose K | V are staged
  const int qoff = (J == 2) ? 0 : (int)(blockIdx.x & 1) * 128;               // ... and this workgroup's query rows inside it
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  // Q fragments first (ordinary loads), consumed before any LDS-DMA is in flight
  const bf16* qg = (const bf16*)a.q + head * 64;
  bf16x8 qf[J][4];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const long long qrow = tok0 + qoff + wave * (32 * J) + j * 32 + r;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (qrow < d.ntok) v = *(const u32x4*)(qg + (size_t)qrow * C + ks * 16 + h * 8);
      qf[j][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  frame_burst<0, 1>((const bf16*)a.k, (const bf16*)a.v, tok0, d.ntok, C, head, lds0, tid);
  f32x16 o[J][2];
  float l2[J][2];
#pragma unroll
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[j][0][i] = 0.f; o[j][1][i] = 0.f; }
    l2[j][0] = 0.f; l2[j][1] = 0.f;
  }
  const int kb0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);
  const int vb0 = (4 * hh + q4) * 128 + pcol * 2, vsw = (q4 >> 1) & 1;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto vtr = [&](const unsigned char* vt, int tokbase, int dt) __attribute__((always_inline)) {
    const unsigned char* p0 = vt + vb0 + tokbase * 128 + ((dt ^ vsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 128));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  dma_wait();
  __syncthreads();
  const int t0 = ((qoff + wave * (32 * J)) / P) * d.tiles;  // first key tile of this wave's frame inside the super-block
#pragma unroll 1
  for (int t = t0; t < t0 + d.tiles; ++t) {
    const unsigned char* Kt = smem + t * 2 * TB;
    const unsigned char* Vt = Kt + TB;
    bf16x8 kf[2][4], vf[2][2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *(const bf16x8*)(Kt + ((kb0 ^ (ks * 32)) + kt * 4096));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[kt][s2][dt] = vtr(Vt, kt * 32 + 16 * s2, dt);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      f32x16 s[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[kt] = mfma32(kf[kt][ks], qf[j][ks], s[kt]);
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const float p = __builtin_amdgcn_exp2f(s[kt][rr] - SOFTMAX_OFF);
          s[kt][rr] = p;
          l2[j][rr & 1] += p;
        }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack8(s[kt], s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) o[j][dt] = mfma32(vf[kt][s2][dt], pb, o[j][dt]);
        }
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const long long qrow = tok0 + qoff + wave * (32 * J) + j * 32 + r;
    float l = l2[j][0] + l2[j][1];
    l += __shfl_xor(l, 32);
    if (qrow >= d.ntok) continue;
    const float inv = (l > 0.f) ? 1.f / l : 0.f;
    bf16* og = (bf16*)a.out + (size_t)qrow * C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 ov;
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = f2bf(o[j][dt][4 * g + k] * inv);
        *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h) = ov;
      }
    if (a.lse && h == 0) {
      const long long b = qrow / P;
      a.lse[(size_t)(b * a.heads + head) * P + (qrow - b * P)] = SOFTMAX_OFF + log2f(fmaxf(l, 1e-30f));
    }
  }
#endif
}

// dQ: the forward's structure (lane = query row); K in the dual-use image (row reads for S^T, transposing reads for dQ^T), V row-read
__global__ __launch_bounds__(256, 2) void frame_attn_dq_kernel(const FrameAttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 2 * TB];     // [tile][K | V]
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, C = a.C, P = a.Lq;
  const long long tok0 = (long long)blockIdx.x * 256;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  const bf16* qg = (const bf16*)a.q + head * 64;
  const bf16* dog = (const bf16*)a.dout + head * 64;
  bf16x8 qf[2][4], dof[2][4];
  float lse[2] = {0.f, 0.f}, delta[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long qrow = tok0 + wave * 64 + j * 32 + r;
    const bool in = qrow < d.ntok;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u}, w = u32x4{0u, 0u, 0u, 0u};
      if (in) {
        v = *(const u32x4*)(qg + (size_t)qrow * C + ks * 16 + h * 8);
        w = *(const u32x4*)(dog + (size_t)qrow * C + ks * 16 + h * 8);
      }
      qf[j][ks] = __builtin_bit_cast(bf16x8, v);
      dof[j][ks] = __builtin_bit_cast(bf16x8, w);
    }
    if (in) {
      const long long b = qrow / P;
      const size_t li = (size_t)(b * a.heads + head) * P + (qrow - b * P);
      lse[j] = a.lse[li];
      delta[j] = a.delta[li];
    }
  }
  frame_burst<2, 0>((const bf16*)a.k, (const bf16*)a.v, tok0, d.ntok, C, head, lds0, tid);
  f32x16 dq[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[j][0][i] = 0.f; dq[j][1][i] = 0.f; }
  const int kr0 = r * 128 + ((h ^ ((((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1))) << 4);
  const int vr0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2, c0 = 2 * (grp & 1) + ((lane & 3) >> 1);
  const int tbA = (4 * hh + q4) * 128 + ((c0 ^ hh) << 4) + 8 * (lane & 1);
  const int tbB = (4 * hh + q4 + 8) * 128 + ((c0 ^ hh ^ 2) << 4) + 8 * (lane & 1);
  const int tsw = (q4 >> 1) & 1;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto ktr = [&](const unsigned char* kt_, int tokbase, int dt) __attribute__((always_inline)) {
    const int o = tokbase * 128 + ((dt ^ tsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kt_ + tbA + o));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kt_ + tbB + o));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  dma_wait();
  __syncthreads();
  const int t0 = ((wave * 64) / P) * d.tiles;
#pragma unroll 1
  for (int t = t0; t < t0 + d.tiles; ++t) {
    const unsigned char* Kt = smem + t * 2 * TB;
    const unsigned char* Vt = Kt + TB;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      bf16x8 kf[4], vf[4], ktf[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8*)(Kt + ((kr0 ^ (ks * 32)) + kt * 4096));
        vf[ks] = *(const bf16x8*)(Vt + ((vr0 ^ (ks * 32)) + kt * 4096));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ktf[s2][dt] = ktr(Kt, kt * 32 + 16 * s2, dt);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(kf[ks], qf[j][ks], s);
          dp = mfma32(vf[ks], dof[j][ks], dp);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s[rr] = __builtin_amdgcn_exp2f(s[rr] - lse[j]) * (dp[rr] - delta[j]);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 db = pack8(s, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[j][dt] = mfma32(ktf[s2][dt], db, dq[j][dt]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long qrow = tok0 + wave * 64 + j * 32 + r;
    if (qrow >= d.ntok) continue;
    bf16* og = (bf16*)a.dq + (size_t)qrow * C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 ov;
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = f2bf(dq[j][dt][4 * g + k] * 0.125f);
        *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h) = ov;
      }
  }
#endif
}

// ---- the whole backward of a super-block in ONE launch: delta, dQ, dK, dV ----------------------------------------------------
// Dense attention inside 256-token frames is HBM-bound (4 * 64 * P FLOP per 4 * 128 B token row of q | k | v | out: 128 FLOP/B at
// P = 256, a third of the ridge), so the lever is bytes: attn_delta (reads dO, O) + dQ (reads q, k, v, dO) + dK/dV (reads q, k, v,
// dO) moved 13 tensor passes; here q, k, v, O, dO are read ONCE and dq, dk, dv written: 8 passes.
// 8 waves = 8 x 32 rows.  Pass A (lane = query): this wave's Q / dO / O rows from global (delta = dO . O in registers, lse | delta
// parked in LDS for pass B), K (dual-use image) | V (row image) of the super-block from LDS region A -> dQ.  Pass B (lane = key):
// K / V row fragments of this wave's 32 keys from region A, Q | dO (transposing image) from region B -> dK, dV.  Both regions are
// requested up front; pass A starts as soon as region A has landed (counted vmcnt: region B's 8 copies stay in flight under it).
template <int SW0, int SW1>
__device__ __forceinline__ void frame_burst512(const bf16* g0, const bf16* g1, long long tok0, long long ntok, int C /* pitch of g0 */,
                                               int C1 /* pitch of g1 */, int head, unsigned lds0, int tid) {
  constexpr int TB = 64 * 128, OOB = (int)0x80000000;
  const int wave = tid >> 6;
  const long long left = ntok - tok0;
  const int rows = left >= 256 ? 256 : (int)left;
  const i32x4 r0 = make_rsrc(g0 + (size_t)tok0 * C, rows * C * 2);
  const i32x4 r1 = make_rsrc(g1 + (size_t)tok0 * C1, rows * C1 * 2);
  const int row = tid >> 3, pp = tid & 7;
  const int dual = (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
  const int sw0 = SW0 == 0 ? ((row >> 1) & 7) : SW0 == 1 ? 4 * ((row >> 1) & 1) : dual;
  const int sw1 = SW1 == 0 ? ((row >> 1) & 7) : SW1 == 1 ? 4 * ((row >> 1) & 1) : dual;
  const int v0 = (row * C + head * 64 + (pp ^ sw0) * 8) * 2, v1 = (row * C1 + head * 64 + (pp ^ sw1) * 8) * 2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool ok = t * 64 + row < rows;
    const unsigned dst = lds0 + t * 2 * TB + wave * 1024;
    dma16(r0, ok ? v0 : OOB, t * 64 * C * 2, dst);
    dma16(r1, ok ? v1 : OOB, t * 64 * C1 * 2, dst + TB);
  }
}

__global__ __launch_bounds__(512, 2) void frame_attn_bwd_kernel(const FrameAttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128, REG = 4 * 2 * TB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * REG + 2048];     // A: [tile][K | V], B: [tile][Q | dO], lse[256] | delta[256]
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, C = a.C, P = a.Lq;
  const long long tok0 = (long long)blockIdx.x * 256;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  float* lse_lds = (float*)(smem + 2 * REG);
  float* del_lds = lse_lds + 256;
  const long long row = tok0 + wave * 32 + r;                // this lane's query row (pass A) = key row (pass B)
  const bool in = row < d.ntok;
  // ---- this wave's Q / dO / O rows, lse; delta
  bf16x8 qf[4], dof[4];
  float lse = 0.f, delta = 0.f;
  {
    const size_t base = (size_t)row * C + head * 64 + h * 8;
    const bf16* qg = (const bf16*)a.q + base;
    const bf16* dog = (const bf16*)a.dout + base;
    const bf16* og = (const bf16*)a.out + base;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u}, w = u32x4{0u, 0u, 0u, 0u}, o = u32x4{0u, 0u, 0u, 0u};
      if (in) {
        v = *(const u32x4*)(qg + ks * 16);
        w = *(const u32x4*)(dog + ks * 16);
        o = *(const u32x4*)(og + ks * 16);
      }
      qf[ks] = __builtin_bit_cast(bf16x8, v);
      dof[ks] = __builtin_bit_cast(bf16x8, w);
      const bf16x8 ofr = __builtin_bit_cast(bf16x8, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += bf2f(dof[ks][e]) * bf2f(ofr[e]);
    }
    delta += __shfl_xor(delta, 32);
    if (in) {
      const long long b = row / P;
      lse = a.lse[(size_t)(b * a.heads + head) * P + (row - b * P)];
    }
    if (h == 0) { lse_lds[wave * 32 + r] = lse; del_lds[wave * 32 + r] = delta; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));      // every ordinary load consumed before an LDS-DMA is in flight
  }
  frame_burst512<2, 0>((const bf16*)a.k, (const bf16*)a.v, tok0, d.ntok, C, C, head, lds0, tid);
  frame_burst512<2, 2>((const bf16*)a.q, (const bf16*)a.dout, tok0, d.ntok, C, C, head, lds0 + REG, tid);
  const int t0 = ((wave * 32) / P) * d.tiles;                // first tile of this wave's frame inside the super-block
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2;
  const int kr0 = r * 128 + ((h ^ ((((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1))) << 4);     // K, dual-use image: row reads
  const int vr0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);                                                         // V, row image
  // transposing read of a dual-use image (K in pass A; Q and dO in pass B)
  const int c0 = 2 * (grp & 1) + ((lane & 3) >> 1);
  const int tbA = (4 * hh + q4) * 128 + ((c0 ^ hh) << 4) + 8 * (lane & 1);
  const int tbB = (4 * hh + q4 + 8) * 128 + ((c0 ^ hh ^ 2) << 4) + 8 * (lane & 1);
  const int tsw = (q4 >> 1) & 1;
  auto ktr = [&](const unsigned char* kt_, int tokbase, int dt) __attribute__((always_inline)) {
    const int o = tokbase * 128 + ((dt ^ tsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kt_ + tbA + o));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kt_ + tbB + o));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // region A (the first 8 copies of this wave) has landed
  __syncthreads();
  // ---- pass A: dQ
  {
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
#pragma unroll 1
    for (int t = t0; t < t0 + d.tiles; ++t) {
      const unsigned char* Kt = smem + t * 2 * TB;
      const unsigned char* Vt = Kt + TB;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        bf16x8 kf[4], vf[4], ktf[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          kf[ks] = *(const bf16x8*)(Kt + ((kr0 ^ (ks * 32)) + kt * 4096));
          vf[ks] = *(const bf16x8*)(Vt + ((vr0 ^ (ks * 32)) + kt * 4096));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) ktf[s2][dt] = ktr(Kt, kt * 32 + 16 * s2, dt);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(kf[ks], qf[ks], s);
          dp = mfma32(vf[ks], dof[ks], dp);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s[rr] = __builtin_amdgcn_exp2f(s[rr] - lse) * (dp[rr] - delta);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 db = pack8(s, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(ktf[s2][dt], db, dq[dt]);
        }
      }
    }
    if (in) {
      bf16* og = (bf16*)a.dq + (size_t)row * C + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 ov;
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = f2bf(dq[dt][4 * g + k] * 0.125f);
          *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h) = ov;
        }
    }
  }
  dma_wait();                                                // region B has landed (and this wave's dQ stores are out)
  __syncthreads();
  // ---- pass B: dK, dV
  {
    // this wave's key rows live in tile (wave / 2), rows (wave & 1) * 32 + r of region A
    const unsigned char* Kown = smem + (wave >> 1) * 2 * TB + (wave & 1) * 4096;
    const unsigned char* Vown = Kown + TB;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const bf16x8*)(Kown + (kr0 ^ (ks * 32)));
      vf[ks] = *(const bf16x8*)(Vown + (vr0 ^ (ks * 32)));
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
#pragma unroll 1
    for (int t = t0; t < t0 + d.tiles; ++t) {
      const unsigned char* Qt = smem + REG + t * 2 * TB;
      const unsigned char* dOt = Qt + TB;
#pragma unroll 1
      for (int qt = 0; qt < 2; ++qt) {
        bf16x8 qa[4], da[4], dotf[2][2], qtf[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          qa[ks] = *(const bf16x8*)(Qt + ((kr0 ^ (ks * 32)) + qt * 4096));
          da[ks] = *(const bf16x8*)(dOt + ((kr0 ^ (ks * 32)) + qt * 4096));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dotf[s2][dt] = ktr(dOt, qt * 32 + 16 * s2, dt);
            qtf[s2][dt] = ktr(Qt, qt * 32 + 16 * s2, dt);
          }
        f32x16 s, dp, pv;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(qa[ks], kf[ks], s);                   // S[q][key]
          dp = mfma32(da[ks], vf[ks], dp);                 // dP[q][key]
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int qb4 = t * 64 + qt * 32 + 8 * g4 + 4 * h;
          const float4 ls = *(const float4*)(lse_lds + qb4), de = *(const float4*)(del_lds + qb4);
          const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int rr = 4 * g4 + k;
            const float p = __builtin_amdgcn_exp2f(s[rr] - lsv[k]);
            pv[rr] = p;
            s[rr] = p * (dp[rr] - dev[k]) * 0.125f;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack8(pv, s2);
          const bf16x8 db = pack8(s, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = mfma32(dotf[s2][dt], pb, dv[dt]);
            dk[dt] = mfma32(qtf[s2][dt], db, dk[dt]);
          }
        }
      }
    }
    if (in) {
      bf16* dkg = (bf16*)a.dk + (size_t)row * C + head * 64;
      bf16* dvg = (bf16*)a.dv + (size_t)row * C + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o1, o2;
#pragma unroll
          for (int k = 0; k < 4; ++k) { o1[k] = f2bf(dk[dt][4 * g + k] * (1.f / SCALE_LOG2)); o2[k] = f2bf(dv[dt][4 * g + k]); }
          *(bf16x4*)(dkg + dt * 32 + 8 * g + 4 * h) = o1;
          *(bf16x4*)(dvg + dt * 32 + 8 * g + 4 * h) = o2;
        }
    }
  }
#endif
}

// ---- the same two kernels on the RAW attn_qkv output -----------------------------------------------------------------------------
// FrameAttention normalises every 64-channel head vector of q, k and v (attention_modules.py:112: normalize(y, dim=-1)).  As a pass
// of its own (qkv_norm_kernel / qkv_norm_bwd_kernel) that moved 6 + 9 tensor rows per token around an attention kernel that moves
// 4 + 8: the normalisation was HALF of the layer's time (profiles/r06_ab_frame_attn.txt).  Here the kernels read qkv [tok][3C]
// (channel = s C + head 64 + c) directly: this wave's own rows are normalised in registers, the staged K / V (and Q) images are
// normalised IN LDS -- 8 lanes per 128-byte row, exactly the arithmetic of qkv_norm_kernel, values rounded to bf16 where the
// two-pass form rounded them -- and the backward's epilogues apply the adjoint of the normalisation (qkv_norm_bwd_kernel's formula)
// to dq, dk, dv while they are fp32 accumulators and write dqkv.
// one [4 tiles][64 rows][128 B] image inside a region (tile stride 2 TB): x <- x * scale / (eps + |x| / 8) per row
// sum over the 8 consecutive lanes that share a row (lanes 8k .. 8k+7), every lane gets the total: two quad permutes and a
// half-row mirror on the DPP path (full-rate VALU) -- __shfl_xor is a ds_bpermute round trip through LDS, three of them per pass
// made this phase the longest of the forward (profiles/r06_frame_attn_stamps.txt: 11.6 K of 40 K cycles)
__device__ __forceinline__ float row8_sum(float v) {
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));     // row_half_mirror: quad 0 <-> quad 1
  return v;
}

template <int NT>
__device__ __forceinline__ void frame_lds_norm(unsigned char* img, float scale, int tid) {
  constexpr int TB = 64 * 128;
#pragma unroll
  for (int pass = 0; pass < 2048 / NT; ++pass) {
    const int e = pass * NT + tid, R = e >> 3, p = e & 7;
    bf16x8* at = (bf16x8*)(img + (R >> 6) * 2 * TB + (R & 63) * 128 + p * 16);
    const bf16x8 x = *at;
    float f[8], ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = bf2f(x[i]); ss += f[i] * f[i]; }
    ss = row8_sum(ss);
    const float inv = scale * __builtin_amdgcn_rcpf(1e-4f + __builtin_amdgcn_sqrtf(ss) * 0.125f);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(f[i] * inv);
    *at = o;
  }
}

// this lane's half (channels ks * 16 + h * 8 + 0..7, ks = 0..3) of a raw 64-channel row -> normalised fragments
__device__ __forceinline__ void frame_row_norm(bf16x8 (&f)[4], float scale) {
  float ss = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += bf2f(f[ks][e]) * bf2f(f[ks][e]);
  ss += __shfl_xor(ss, 32);
  const float inv = scale * __builtin_amdgcn_rcpf(1e-4f + __builtin_amdgcn_sqrtf(ss) * 0.125f);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) f[ks][e] = f2bf(bf2f(f[ks][e]) * inv);
}

struct FrameQkvDev {
  const bf16* qkv; bf16* out; float* lse; const bf16* dout; bf16* dqkv;
  long long ntok;
  int P, heads, C, tiles;
};
#ifdef FRAME_STAMP          // diagnostic build (make variant VSRC=attention VNAME=fstamp): 8 x s_memrealtime per workgroup into `lse + off`
#define FSTAMP(i) do { if (tid == 0) fst[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256, 2) void frame_attn_qkv_fwd_kernel(const FrameQkvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 2 * TB];     // [tile][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, C = d.C, C3 = 3 * d.C, P = d.P;
  const long long tok0 = (long long)blockIdx.x * 256;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
#ifdef FRAME_STAMP
  long long* fst = (long long*)d.dqkv + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
#endif
  FSTAMP(0);
  // K | V of the super-block are requested first; this wave's Q rows are loaded and normalised under the copy (the compiler's
  // wait for these ordinary loads also covers the older LDS-DMA copies: both are needed before the barrier anyway)
  frame_burst<0, 1>(d.qkv + C, d.qkv + 2 * C, tok0, d.ntok, C3, head, lds0, tid);
  FSTAMP(1);
  const bf16* qg = d.qkv + head * 64;
  bf16x8 qf[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long qrow = tok0 + wave * 64 + j * 32 + r;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (qrow < d.ntok) v = *(const u32x4*)(qg + (size_t)qrow * C3 + ks * 16 + h * 8);
      qf[j][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) frame_row_norm(qf[j], SCALE_LOG2);      // q' = log2(e) / 8 * normalised q (one rounding, as qkv_norm_kernel)
  FSTAMP(2);
  f32x16 o[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[j][0][i] = 0.f; o[j][1][i] = 0.f; }
  float l2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const int kb0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);
  const int vb0 = (4 * hh + q4) * 128 + pcol * 2, vsw = (q4 >> 1) & 1;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto vtr = [&](const unsigned char* vt, int tokbase, int dt) __attribute__((always_inline)) {
    const unsigned char* p0 = vt + vb0 + tokbase * 128 + ((dt ^ vsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 128));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  dma_wait();
  __syncthreads();
  FSTAMP(3);
  frame_lds_norm<256>(smem, 1.f, tid);                 // K
  frame_lds_norm<256>(smem + TB, 1.f, tid);            // V
  __syncthreads();
  FSTAMP(4);
  const int t0 = ((wave * 64) / P) * d.tiles;
#pragma unroll 1
  for (int t = t0; t < t0 + d.tiles; ++t) {
    const unsigned char* Kt = smem + t * 2 * TB;
    const unsigned char* Vt = Kt + TB;
    bf16x8 kf[2][4], vf[2][2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *(const bf16x8*)(Kt + ((kb0 ^ (ks * 32)) + kt * 4096));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[kt][s2][dt] = vtr(Vt, kt * 32 + 16 * s2, dt);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x16 s[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[kt] = mfma32(kf[kt][ks], qf[j][ks], s[kt]);
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const float p = __builtin_amdgcn_exp2f(s[kt][rr] - SOFTMAX_OFF);
          s[kt][rr] = p;
          l2[j][rr & 1] += p;
        }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack8(s[kt], s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) o[j][dt] = mfma32(vf[kt][s2][dt], pb, o[j][dt]);
        }
    }
  }
  FSTAMP(5);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long qrow = tok0 + wave * 64 + j * 32 + r;
    float l = l2[j][0] + l2[j][1];
    l += __shfl_xor(l, 32);
    if (qrow >= d.ntok) continue;
    const float inv = (l > 0.f) ? 1.f / l : 0.f;
    bf16* og = d.out + (size_t)qrow * C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 ov;
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = f2bf(o[j][dt][4 * g + k] * inv);
        *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h) = ov;
      }
    if (h == 0) {
      const long long b = qrow / P;
      d.lse[(size_t)(b * d.heads + head) * P + (qrow - b * P)] = SOFTMAX_OFF + log2f(fmaxf(l, 1e-30f));
    }
  }
  FSTAMP(6);
#endif
}

// adjoint of x -> x / (eps + |x| / 8) on this lane's half of a row held as fp32 accumulators g[dt][4 g4 + k] <-> channel
// dt * 32 + 8 g4 + 4 h + k (qkv_norm_bwd_kernel: o = g k1 - x k2), written to `dst` (the row's 64-channel slot of dqkv); x from `src`
__device__ __forceinline__ void frame_norm_adjoint_store(const f32x16 (&g)[2], float gscale, const bf16* src, bf16* dst, int h) {
  float x[2][16], ss = 0.f, dot = 0.f;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const bf16x4 v = *(const bf16x4*)(src + dt * 32 + 8 * g4 + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float f = bf2f(v[k]);
        x[dt][4 * g4 + k] = f;
        ss += f * f;
        dot += f * (g[dt][4 * g4 + k] * gscale);
      }
    }
  ss += __shfl_xor(ss, 32);
  dot += __shfl_xor(dot, 32);
  const float n = sqrtf(ss), sden = 1e-4f + n * 0.125f;
  const float k1 = 1.f / sden, k2 = (n > 0.f) ? dot * 0.125f / (sden * sden * n) : 0.f;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 ov;
#pragma unroll
      for (int k = 0; k < 4; ++k) ov[k] = f2bf(g[dt][4 * g4 + k] * gscale * k1 - x[dt][4 * g4 + k] * k2);
      *(bf16x4*)(dst + dt * 32 + 8 * g4 + 4 * h) = ov;
    }
}

__global__ __launch_bounds__(512, 2) void frame_attn_qkv_bwd_kernel(const FrameQkvDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128, REG = 4 * 2 * TB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * REG + 2048];     // A: [tile][K | V], B: [tile][Q | dO], lse[256] | delta[256]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, C = d.C, C3 = 3 * d.C, P = d.P;
  const long long tok0 = (long long)blockIdx.x * 256;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  float* lse_lds = (float*)(smem + 2 * REG);
  float* del_lds = lse_lds + 256;
  const long long row = tok0 + wave * 32 + r;                // this lane's query row (pass A) = key row (pass B)
  const bool in = row < d.ntok;
  frame_burst512<2, 0>(d.qkv + C, d.qkv + 2 * C, tok0, d.ntok, C3, C3, head, lds0, tid);      // region A; the own rows load under it
  bf16x8 qf[4], dof[4];
  float lse = 0.f, delta = 0.f;
  {
    const bf16* qg = d.qkv + (size_t)row * C3 + head * 64 + h * 8;
    const bf16* dog = d.dout + (size_t)row * C + head * 64 + h * 8;
    const bf16* og = d.out + (size_t)row * C + head * 64 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u}, w = u32x4{0u, 0u, 0u, 0u}, o = u32x4{0u, 0u, 0u, 0u};
      if (in) {
        v = *(const u32x4*)(qg + ks * 16);
        w = *(const u32x4*)(dog + ks * 16);
        o = *(const u32x4*)(og + ks * 16);
      }
      qf[ks] = __builtin_bit_cast(bf16x8, v);
      dof[ks] = __builtin_bit_cast(bf16x8, w);
      const bf16x8 ofr = __builtin_bit_cast(bf16x8, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += bf2f(dof[ks][e]) * bf2f(ofr[e]);
    }
    delta += __shfl_xor(delta, 32);
    frame_row_norm(qf, SCALE_LOG2);
    if (in) {
      const long long b = row / P;
      lse = d.lse[(size_t)(b * d.heads + head) * P + (row - b * P)];
    }
    if (h == 0) { lse_lds[wave * 32 + r] = lse; del_lds[wave * 32 + r] = delta; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));      // every ordinary load consumed before region B is requested
  }
  frame_burst512<2, 2>(d.qkv, d.dout, tok0, d.ntok, C3, C, head, lds0 + REG, tid);             // region B: lands under pass A
  const int t0 = ((wave * 32) / P) * d.tiles;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2;
  const int kr0 = r * 128 + ((h ^ ((((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1))) << 4);
  const int vr0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  const int c0 = 2 * (grp & 1) + ((lane & 3) >> 1);
  const int tbA = (4 * hh + q4) * 128 + ((c0 ^ hh) << 4) + 8 * (lane & 1);
  const int tbB = (4 * hh + q4 + 8) * 128 + ((c0 ^ hh ^ 2) << 4) + 8 * (lane & 1);
  const int tsw = (q4 >> 1) & 1;
  auto ktr = [&](const unsigned char* kt_, int tokbase, int dt) __attribute__((always_inline)) {
    const int o = tokbase * 128 + ((dt ^ tsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kt_ + tbA + o));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kt_ + tbB + o));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // region A (the first 8 copies of this wave) has landed
  __syncthreads();
  frame_lds_norm<512>(smem, 1.f, tid);                        // K
  frame_lds_norm<512>(smem + TB, 1.f, tid);                   // V
  __syncthreads();
  // ---- pass A: dQ
  {
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
#pragma unroll 1
    for (int t = t0; t < t0 + d.tiles; ++t) {
      const unsigned char* Kt = smem + t * 2 * TB;
      const unsigned char* Vt = Kt + TB;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        bf16x8 kf[4], vf[4], ktf[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          kf[ks] = *(const bf16x8*)(Kt + ((kr0 ^ (ks * 32)) + kt * 4096));
          vf[ks] = *(const bf16x8*)(Vt + ((vr0 ^ (ks * 32)) + kt * 4096));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) ktf[s2][dt] = ktr(Kt, kt * 32 + 16 * s2, dt);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(kf[ks], qf[ks], s);
          dp = mfma32(vf[ks], dof[ks], dp);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s[rr] = __builtin_amdgcn_exp2f(s[rr] - lse) * (dp[rr] - delta);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 db = pack8(s, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(ktf[s2][dt], db, dq[dt]);
        }
      }
    }
    if (in) frame_norm_adjoint_store(dq, 0.125f, d.qkv + (size_t)row * C3 + head * 64, d.dqkv + (size_t)row * C3 + head * 64, h);
  }
  dma_wait();                                                // region B has landed (and this wave's dq stores are out)
  __syncthreads();
  frame_lds_norm<512>(smem + REG, SCALE_LOG2, tid);           // Q (the image pass B multiplies: q' as in the forward)
  __syncthreads();
  // ---- pass B: dK, dV
  {
    const unsigned char* Kown = smem + (wave >> 1) * 2 * TB + (wave & 1) * 4096;
    const unsigned char* Vown = Kown + TB;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const bf16x8*)(Kown + (kr0 ^ (ks * 32)));
      vf[ks] = *(const bf16x8*)(Vown + (vr0 ^ (ks * 32)));
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
#pragma unroll 1
    for (int t = t0; t < t0 + d.tiles; ++t) {
      const unsigned char* Qt = smem + REG + t * 2 * TB;
      const unsigned char* dOt = Qt + TB;
#pragma unroll 1
      for (int qt = 0; qt < 2; ++qt) {
        bf16x8 qa[4], da[4], dotf[2][2], qtf[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          qa[ks] = *(const bf16x8*)(Qt + ((kr0 ^ (ks * 32)) + qt * 4096));
          da[ks] = *(const bf16x8*)(dOt + ((kr0 ^ (ks * 32)) + qt * 4096));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dotf[s2][dt] = ktr(dOt, qt * 32 + 16 * s2, dt);
            qtf[s2][dt] = ktr(Qt, qt * 32 + 16 * s2, dt);
          }
        f32x16 s, dp, pv;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(qa[ks], kf[ks], s);
          dp = mfma32(da[ks], vf[ks], dp);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int qb4 = t * 64 + qt * 32 + 8 * g4 + 4 * h;
          const float4 ls = *(const float4*)(lse_lds + qb4), de = *(const float4*)(del_lds + qb4);
          const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int rr = 4 * g4 + k;
            const float p = __builtin_amdgcn_exp2f(s[rr] - lsv[k]);
            pv[rr] = p;
            s[rr] = p * (dp[rr] - dev[k]) * 0.125f;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack8(pv, s2);
          const bf16x8 db = pack8(s, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = mfma32(dotf[s2][dt], pb, dv[dt]);
            dk[dt] = mfma32(qtf[s2][dt], db, dk[dt]);
          }
        }
      }
    }
    if (in) {
      frame_norm_adjoint_store(dk, 1.f / SCALE_LOG2, d.qkv + (size_t)row * C3 + C + head * 64, d.dqkv + (size_t)row * C3 + C + head * 64, h);
      frame_norm_adjoint_store(dv, 1.f, d.qkv + (size_t)row * C3 + 2 * C + head * 64, d.dqkv + (size_t)row * C3 + 2 * C + head * 64, h);
    }
  }
#endif
}

// the frame kernels serve: dense (mask_mode 0), no table, no KV ring strides, no split-KV, Lq == Lk == P in {64, 128, 256}
static inline bool frame_attn_ok(const OnirisAttnArgs& a) {
  return !(a.frame_kernel & 1) && a.mask_mode == 0 && !a.kv_num && !a.sched && a.kv_splits <= 1 && a.k_bstride == 0 && a.v_bstride == 0 &&
         a.Lq == a.Lk && (a.Lq == 64 || a.Lq == 128 || a.Lq == 256);
}
