// VideoAttention / FrameAttention kernels for gfx950 (head dim 64, MFMA 32x32x16 bf16 -> fp32).
//
// Flash-style block-sparse attention that consumes the reference's BlockMask table (128-token kv blocks per
// 128-token q block, bit-exact copy of make_train_mask / make_infer_mask) and applies mask_mod per element on
// partially masked tiles -- the semantics of the compiled FlexAttention kernel.
//
// Forward / dQ: one workgroup = 128 query rows (4 waves x 32 rows), S^T = K.Q^T is computed with the KEY index on
// the MFMA rows and the QUERY on the lanes, so a lane owns one query row: row max / row sum are 32 in-register ops
// plus one cross-half shuffle, and the S^T accumulator is directly the B operand of O^T += V^T.P^T (no LDS round
// trip).  K / V tiles (64 keys) are prefetched into registers one tile ahead and staged row-major in LDS (144-byte
// rows); row fragments are 16-byte reads, the transposed operands (V^T, K^T, Q^T, dO^T: MFMA k index = token) come
// from the SAME tiles through the gfx950 transposing read ds_read_b64_tr_b16.  dK/dV: one workgroup = 128 keys,
// lane = key, S = Q.K^T with queries on the rows, P and dS feed dV^T += dO^T.P and dK^T += Q^T.dS from registers.
#include <type_traits>
#include "common.h"
#include "lds_dma.h"
#include "../../include/oniris.h"

#define NEG_BIG (-1.0e30f)
// Softmax without a running maximum: VideoAttention / FrameAttention L2-normalise every 64-vector of q and k to
// norm 8 (attention_modules.py:38) and the xPos factor of an ALLOWED (causal) pair is <= 1 (RoPe.py:60-67), so
// |q.k|/8 <= 8 and the log2-domain score is within +-11.6: exp2(score - SOFTMAX_OFF) cannot overflow, and dropping
// the max / rescale work removes about three quarters of the per-element VALU instructions (the kernel is
// VALU-bound at head dim 64: 256 MFMA FLOP per score element against ~10 VALU instructions with a running max).
#define SOFTMAX_OFF 12.0f
#define SCALE_LOG2 (0.125f * 1.4426950408889634f)
#define KROW 144   // bytes per row of a [64 tok][64 ch] bf16 tile (128 + 16 pad)

struct AttnDev {
  OnirisAttnArgs a;
  int pshift;     // log2(P)
  int qf_off;     // (Lk - Lq) / P  (frame offset of the queries inside the key sequence, mask_mode 1)
  int tshift;     // log2(table block / 128): the BlockMask BLOCK_SIZE is P when P >= 128
};

// ---- mask_mod on token indices ---------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ bool tok_allowed(int qtok, int ktok, int pshift, int T, int qf_off) {
  if (MODE == 0) return true;
  const int qf = qtok >> pshift, kf = ktok >> pshift;
  if (MODE == 1) return kf <= qf + qf_off;
  // TrainingMask (attention_masking.py:15-24): clean q sees clean kf <= qf; noisy q (f = qf-T) sees clean kf < f
  // and its own noisy frame.
  if (qf < T) return kf <= qf;
  return (kf < T && kf < qf - T) || (kf == qf);
}

// 0 = nothing allowed, 1 = partially masked, 2 = everything allowed, for token ranges [q0,q1] x [k0,k1]
template <int MODE>
__device__ __forceinline__ int classify(int q0, int q1, int k0, int k1, int pshift, int T, int qf_off) {
  if (MODE == 0) return 2;
  const int qa = q0 >> pshift, qb = q1 >> pshift, ka = k0 >> pshift, kb = k1 >> pshift;
  if (MODE == 1) {
    if (kb <= qa + qf_off) return 2;
    if (ka > qb + qf_off) return 0;
    return 1;
  }
  if (qb < T) {                       // clean queries (a 128-token block never straddles the clean|noisy border)
    if (kb <= qa) return 2;
    if (ka > qb) return 0;
    return 1;
  }
  if (kb < T) {                       // noisy queries, clean keys
    if (kb < qa - T) return 2;
    if (ka >= qb - T) return 0;
    return 1;
  }
  if (ka == kb && qa == qb && ka == qa) return 2;
  if (kb < qa || ka > qb) return 0;
  return 1;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s2) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(v[8 * s2 + e]);
  return o;
}

// ---- 64-token x 64-channel bf16 tiles: global -> registers (prefetch) -> LDS rows of KROW bytes -----------------
__device__ __forceinline__ void tile_load(u32x4 (&v)[2], const bf16* g, int tok0, int L, int C, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 256;
    const int row = e >> 3, part = e & 7;
    v[i] = u32x4{0u, 0u, 0u, 0u};
    if (tok0 + row < L) v[i] = *(const u32x4*)(g + (size_t)(tok0 + row) * C + part * 8);
  }
}
__device__ __forceinline__ void tile_store(unsigned char* lds, const u32x4 (&v)[2], int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 256;
    *(u32x4*)(lds + (e >> 3) * KROW + (e & 7) * 16) = v[i];
  }
}
// Transposed fragment of a row-major tile via the gfx950 transposing LDS read: this lane gets channel
// chbase + (lane & 31) of the 8 tokens {tokbase + 4h + 0..3, tokbase + 8 + 4h + 0..3} -- exactly the k order in which
// an S^T / P accumulator (rows = tokens) is consumed as the other MFMA operand.
__device__ __forceinline__ bf16x8 trfrag(const unsigned char* lds, int tokbase, int chbase, int lane) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int grp = lane >> 4, hh = grp >> 1, q = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);
  const unsigned char* p0 = lds + (tokbase + 4 * hh + q) * KROW + (chbase + pcol) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * KROW));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// XCD-aware workgroup order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
// 4 MB L2).  All workgroups of one (batch, head) pair read the same K / V (1 MB each at L = 8192), so a pair is
// pinned to ONE XCD whenever the number of pairs is a multiple of 8: x = position inside the pair's row of the grid,
// bh = pair index.  (Measured: without it K/V stream from the Infinity Cache, ~1 us per tile.)
__device__ __forceinline__ void attn_block_decode(int& x, int& head, int& b) {
  const int nx = gridDim.x, nbh = gridDim.y * gridDim.z;
  const int L = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);
  int bh;
  if ((nbh & 7) == 0) {
    const int xcd = L & 7, k = L >> 3;
    bh = xcd + 8 * (k / nx);
    x = k % nx;
  } else {                                           // (the grid's own coordinates: no division)
    x = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
    return;
  }
  head = bh % gridDim.y;
  b = bh / gridDim.y;
}

// ================================================================================================================
// forward.  K / V tiles (64 keys x 64 channels, 128-byte rows) go global -> LDS by LDS-DMA into two alternating
// buffers: the copy of tile i+1 runs under the MFMA / softmax work of tile i, one barrier per tile, no staging
// registers.  The rows are unpadded (the DMA image is lane-linear), so the 16-byte pieces are XOR-swizzled on the
// source side: K (read row-wise, ds_read_b128, 16 rows per access group) with (row>>1)&7, V (read through the
// transposing ds_read_b64_tr_b16, 4 rows x 64 bytes) with 4*bit1(row).
// KS key streams per workgroup: the 4 waves are 4/KS query waves (32 rows each) x KS streams; stream j walks the key
// tiles j, j+KS, ... of the block's list (own LDS buffers) and the streams' (O, l) partial results meet in LDS at the
// end -- without a running max they simply add.  With a causal table the last query blocks have the longest lists
// and set the launch time; KS = 2 halves that critical path.
template <int MODE, int KS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128;                     // bytes of one tile
  constexpr int QW = 4 / KS, NTS = 64 * QW, NPC = 512 / NTS;      // query waves, threads per stream, pieces per thread and tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[KS * 2 * 2 * TB];     // [stream][buffer][K | V]
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int qwv = wave % QW, st = wave / QW, tis = tid % NTS;
  // split-KV (decode of one frame against a long KV ring: Lq / 128 query blocks x heads x B workgroups is a handful):
  // kv_splits workgroups share a query block, each walks a contiguous range of its key tiles and leaves an un-normalised
  // partial (O, l) in split_ws; attn_split_reduce_kernel adds them (no running maximum here: partials simply add)
  const int nsp = (MODE == 0 && KS == 1 && a.kv_splits > 1) ? a.kv_splits : 1;
  int bx_, head, b;
  attn_block_decode(bx_, head, b);
  const int nqb = (nsp == 1) ? (int)gridDim.x : (int)gridDim.x / nsp;
  const int sp = (nsp == 1) ? 0 : bx_ % nsp;
  const int qbw = nqb - 1 - ((nsp == 1) ? bx_ : bx_ / nsp);      // heaviest (latest) query blocks first
  const int C = a.C, Lq = a.Lq, Lk = a.Lk;
  const int qw0 = qbw * (32 * QW) + qwv * 32;
  const int qrow = qw0 + r;
  const int qb = (qbw * (32 * QW)) >> 7;           // 128-token block of the mask table

  const bf16* qg = (const bf16*)a.q + (size_t)b * Lq * C + head * 64;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (qrow < Lq) v = *(const u32x4*)(qg + (size_t)qrow * C + ks * 16 + h * 8);
    qf[ks] = __builtin_bit_cast(bf16x8, v);
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float l2[2] = {0.f, 0.f};                        // this lane's half of the row sum (two chains: packed adds)

  const int trow = qb >> d.tshift, tmask = (1 << d.tshift) - 1;
  const int nkv = a.kv_num ? (a.kv_num[trow] << d.tshift) : (Lk + 127) / 128;
  const int nsub = nkv * 2;
  // the row's block list sits in one VGPR (lane i = entry i; rows with more than 64 entries fall back to memory):
  // a scalar load per tile put ~1 us of load latency into every iteration
  const int nent = a.kv_num ? a.kv_num[trow] : 0;
  const int kvl = (a.kv_idx && lane < nent) ? a.kv_idx[(size_t)trow * a.tab_cols + lane] : 0;
  asm volatile("" ::"v"(kvl));                      // consume the ordinary load before any LDS-DMA is in flight
  auto key_start = [&](int idx) __attribute__((always_inline)) {
    const int j = idx >> 1, jj = j >> d.tshift;
    int e = j;
    if (a.kv_idx) e = ((nent <= 64 ? __builtin_amdgcn_readlane(kvl, jj) : a.kv_idx[(size_t)trow * a.tab_cols + jj]) << d.tshift) + (j & tmask);
    return e * 128 + (idx & 1) * 64;
  };

  // DMA descriptors: NPC 16-byte pieces per thread and tile
  constexpr int OOB = (int)0x80000000;
  int kvo[NPC], vvo[NPC], prow[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int e = i * NTS + tis, row = e >> 3, pp = e & 7;
    prow[i] = row;
    kvo[i] = (row * C + head * 64 + (pp ^ ((row >> 1) & 7)) * 8) * 2;
    vvo[i] = (row * C + head * 64 + (pp ^ (4 * ((row >> 1) & 1))) * 8) * 2;
  }
  const i32x4 rs_k = make_rsrc((const bf16*)a.k + (size_t)b * (a.k_bstride ? (size_t)a.k_bstride : (size_t)Lk * C), Lk * C * 2);
  const i32x4 rs_v = make_rsrc((const bf16*)a.v + (size_t)b * (a.v_bstride ? (size_t)a.v_bstride : (size_t)Lk * C), Lk * C * 2);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  auto issue = [&](int key0, int bsel) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (st * 2 + bsel) * 2 * TB + qwv * 1024);
    const int left = Lk - key0, so = key0 * C * 2;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const bool ok = prow[i] < left;
      dma16(rs_k, ok ? kvo[i] : OOB, so, dst + i * (NTS * 16));
      dma16(rs_v, ok ? vvo[i] : OOB, so, dst + TB + i * (NTS * 16));
    }
  };
  // fragment addresses
  const int kb0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);                 // K rows kt*32 + r, k-step ks: ^ (ks*32), + kt*4096
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);
  const int vb0 = (4 * hh + q4) * 128 + pcol * 2, vsw = (q4 >> 1) & 1;    // V^T: + tokbase*128 + ((dt ^ vsw)*64)
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

  int bsel = 0;
  const int sub0 = (nsp == 1) ? 0 : (int)((long long)nsub * sp / nsp);                    // this split's tiles
  const int sub1 = (nsp == 1) ? nsub : (int)((long long)nsub * (sp + 1) / nsp);
  if (sub0 + st < sub1) issue(key_start(sub0 + st), 0);
  const int niter = (sub1 - sub0 + KS - 1) / KS;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int idx = sub0 + it * KS + st;
    const bool act = idx < sub1;
    const int key0 = act ? key_start(idx) : 0;
    dma_wait();
    __syncthreads();                               // tile idx has landed for everybody; buffer bsel^1 is free again
    if (idx + KS < sub1) issue(key_start(idx + KS), bsel ^ 1);
    const unsigned char* Kt = smem + (st * 2 + bsel) * 2 * TB;
    const unsigned char* Vt = Kt + TB;
    bsel ^= 1;
    if (!act) continue;
    int cls = (key0 >= Lk) ? 0 : classify<MODE>(qw0, qw0 + 31, key0, key0 + 63, d.pshift, a.T, d.qf_off);
    if (key0 + 63 >= Lk && cls == 2) cls = 1;
    if (cls == 0 || qw0 >= Lq) continue;

    // all fragment reads of the tile go out first (K row fragments, then the transposed V fragments): the V reads
    // complete under the S MFMAs and the softmax, nothing in the chain below waits for LDS
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
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[kt] = mfma32(kf[kt][ks], qf[ks], s[kt]);
    }
    // two straight-line versions (one uniform branch per tile): fully allowed tiles carry no mask code at all
    auto softmax = [&](auto masked_) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_)::value;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          float p = __builtin_amdgcn_exp2f(s[kt][rr] - SOFTMAX_OFF);            // (q carries log2(e)/8: qkv_norm_kernel)
          if constexpr (MASKED) {
            const int key = key0 + kt * 32 + mfma_row(rr, lane);
            if (key >= Lk || !tok_allowed<MODE>(qrow, key, d.pshift, a.T, d.qf_off)) p = 0.f;
          }
          s[kt][rr] = p;
          l2[rr & 1] += p;
        }
    };
    if (cls == 2) softmax(std::false_type{});
    else softmax(std::true_type{});
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(s[kt], s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(vf[kt][s2][dt], pb, o[dt]);     // O^T[dv][q] += V^T[dv][key] P^T[key][q]
      }
  }
  float l = l2[0] + l2[1];
  if constexpr (KS > 1) {                          // streams 1 .. KS-1 hand their partial (O, l) to stream 0 through LDS
    float* red = (float*)smem;                     // [KS - 1][QW][33][64] floats
    __syncthreads();
    if (st >= 1) {
      float* rw = red + (size_t)((st - 1) * QW + qwv) * 33 * 64;
#pragma unroll
      for (int i = 0; i < 16; ++i) { rw[i * 64 + lane] = o[0][i]; rw[(16 + i) * 64 + lane] = o[1][i]; }
      rw[32 * 64 + lane] = l;
    }
    __syncthreads();
    if (st != 0) return;
#pragma unroll
    for (int s_ = 0; s_ < KS - 1; ++s_) {
      const float* rr_ = red + (size_t)(s_ * QW + qwv) * 33 * 64;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] += rr_[i * 64 + lane]; o[1][i] += rr_[(16 + i) * 64 + lane]; }
      l += rr_[32 * 64 + lane];
    }
  }
  if (qrow >= Lq) return;
  l += __shfl_xor(l, 32);                          // the other half of the keys of every tile lives in lane ^ 32
  if (nsp > 1) {                                   // partial of this split: [split][b][head][q][64 channels | l] fp32
    float* pw = a.split_ws + ((((size_t)sp * a.B + b) * a.heads + head) * Lq + qrow) * 65;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[dt * 32 + 8 * g + 4 * h + k] = o[dt][4 * g + k];
    if (h == 0) pw[64] = l;
    return;
  }
  const float inv = (l > 0.f) ? 1.f / l : 0.f;
  bf16* og = (bf16*)a.out + ((size_t)b * Lq + qrow) * C + head * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 ov;
#pragma unroll
      for (int k = 0; k < 4; ++k) ov[k] = f2bf(o[dt][4 * g + k] * inv);
      *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h) = ov;
    }
  if (a.lse && h == 0) a.lse[(size_t)(b * a.heads + head) * Lq + qrow] = SOFTMAX_OFF + log2f(fmaxf(l, 1e-30f));
#endif
}

// out[b][q][head*64 + c] = sum_s O_s / sum_s l_s over the kv_splits partials of attn_fwd_kernel (one thread per (b, head, q, 8 channels))
__global__ void attn_split_reduce_kernel(const float* __restrict__ ws, bf16* __restrict__ out, float* __restrict__ lse, int nsp,
                                         int B, int heads, int Lq, int C) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int part = (int)(gid & 7);
  const long long row = gid >> 3;                  // (b * heads + head) * Lq + q
  if (row >= (long long)B * heads * Lq) return;
  const int q = (int)(row % Lq), head = (int)((row / Lq) % heads), b = (int)(row / ((long long)Lq * heads));
  float acc[8], l = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const size_t plane = (size_t)B * heads * Lq * 65;
  for (int s_ = 0; s_ < nsp; ++s_) {
    const float* p = ws + (size_t)s_ * plane + (size_t)row * 65;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += p[part * 8 + i];
    l += p[64];
  }
  const float inv = (l > 0.f) ? 1.f / l : 0.f;
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f2bf(acc[i] * inv);
  *(bf16x8*)(out + ((size_t)b * Lq + q) * C + head * 64 + part * 8) = o;
  if (lse && part == 0) lse[row] = SOFTMAX_OFF + log2f(fmaxf(l, 1e-30f));
}

#include <cstdlib>
#include "attention_ws.h"
#include "attention_frame.h"

// ================================================================================================================
// backward: dQ   (the forward's structure: lane = query row, LDS-DMA double-buffered K / V tiles, KS key streams
// whose partial dQ add up in LDS).  K is read both row-wise (S^T = K.Q^T) and transposed (dQ^T += K^T.dS^T): it is
// staged once, in the dual-use image of attention_bwd_ws.h (conflict-free for both kinds of read); V is only read
// row-wise (dP^T = V.dO^T).
template <int MODE, int KS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128;
  constexpr int QW = 4 / KS, NTS = 64 * QW, NPC = 512 / NTS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[KS * 2 * 2 * TB];     // [stream][buffer][K | V]
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int qwv = wave % QW, st = wave / QW, tis = tid % NTS;
  const int nqb = gridDim.x;
  int bx_, head, b;
  attn_block_decode(bx_, head, b);
  const int qbw = nqb - 1 - bx_;
  const int C = a.C, Lq = a.Lq, Lk = a.Lk;
  const int qw0 = qbw * (32 * QW) + qwv * 32;
  const int qrow = qw0 + r;
  const int qb = (qbw * (32 * QW)) >> 7;

  const bf16* qg = (const bf16*)a.q + (size_t)b * Lq * C + head * 64;
  const bf16* dog = (const bf16*)a.dout + (size_t)b * Lq * C + head * 64;
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 v = u32x4{0u, 0u, 0u, 0u}, w = u32x4{0u, 0u, 0u, 0u};
    if (qrow < Lq) {
      v = *(const u32x4*)(qg + (size_t)qrow * C + ks * 16 + h * 8);
      w = *(const u32x4*)(dog + (size_t)qrow * C + ks * 16 + h * 8);
    }
    qf[ks] = __builtin_bit_cast(bf16x8, v);
    dof[ks] = __builtin_bit_cast(bf16x8, w);
  }
  float lse = 0.f, delta = 0.f;
  if (qrow < Lq) {
    lse = a.lse[(size_t)(b * a.heads + head) * Lq + qrow];
    delta = a.delta[(size_t)(b * a.heads + head) * Lq + qrow];
  }
  asm volatile("" ::"v"(lse), "v"(delta));         // consume the ordinary loads before any LDS-DMA is in flight
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

  const int trow = qb >> d.tshift, tmask = (1 << d.tshift) - 1;
  const int nkv = a.kv_num ? (a.kv_num[trow] << d.tshift) : (Lk + 127) / 128;
  const int nsub = nkv * 2;
  // the row's block list sits in one VGPR (lane i = entry i; rows with more than 64 entries fall back to memory):
  // a scalar load per tile put ~1 us of load latency into every iteration
  const int nent = a.kv_num ? a.kv_num[trow] : 0;
  const int kvl = (a.kv_idx && lane < nent) ? a.kv_idx[(size_t)trow * a.tab_cols + lane] : 0;
  asm volatile("" ::"v"(kvl));                      // consume the ordinary load before any LDS-DMA is in flight
  auto key_start = [&](int idx) __attribute__((always_inline)) {
    const int j = idx >> 1, jj = j >> d.tshift;
    int e = j;
    if (a.kv_idx) e = ((nent <= 64 ? __builtin_amdgcn_readlane(kvl, jj) : a.kv_idx[(size_t)trow * a.tab_cols + jj]) << d.tshift) + (j & tmask);
    return e * 128 + (idx & 1) * 64;
  };

  constexpr int OOB = (int)0x80000000;
  int kvo[NPC], vvo[NPC], prow[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int e = i * NTS + tis, row = e >> 3, pp = e & 7;
    prow[i] = row;
    // K is read row-wise (S^T) AND transposed (dQ^T): one image, chunk c of row R at c ^ f(R), f = bit1 << 2 | bit3 << 1 | bit2,
    // is conflict-free for both (attention_bwd_ws.h)
    kvo[i] = (row * C + head * 64 + (pp ^ ((((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1))) * 8) * 2;
    vvo[i] = (row * C + head * 64 + (pp ^ ((row >> 1) & 7)) * 8) * 2;            // V: row-read swizzle
  }
  const i32x4 rs_k = make_rsrc((const bf16*)a.k + (size_t)b * Lk * C, Lk * C * 2);
  const i32x4 rs_v = make_rsrc((const bf16*)a.v + (size_t)b * Lk * C, Lk * C * 2);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  auto issue = [&](int key0, int bsel) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (st * 2 + bsel) * 2 * TB + qwv * 1024);
    const int left = Lk - key0, so = key0 * C * 2;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const bool ok = prow[i] < left;
      dma16(rs_k, ok ? kvo[i] : OOB, so, dst + i * (NTS * 16));
      dma16(rs_v, ok ? vvo[i] : OOB, so, dst + TB + i * (NTS * 16));
    }
  };
  // fragment addresses: row reads of K (tr swizzle: piece ^ 4*bit1(row)) and V (piece ^ ((row>>1)&7)); rows kt*32 + r
  const int kr0 = r * 128 + ((h ^ ((((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1))) << 4);   // k-step ks: chunk (2ks+h) ^ f(r)
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

  int bsel = 0;
  if (st < nsub) issue(key_start(st), 0);
  const int niter = (nsub + KS - 1) / KS;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int idx = it * KS + st;
    const bool act = idx < nsub;
    const int key0 = act ? key_start(idx) : 0;
    dma_wait();
    __syncthreads();
    if (idx + KS < nsub) issue(key_start(idx + KS), bsel ^ 1);
    const unsigned char* Kt = smem + (st * 2 + bsel) * 2 * TB;
    const unsigned char* Vt = Kt + TB;
    bsel ^= 1;
    if (!act) continue;
    int cls = (key0 >= Lk) ? 0 : classify<MODE>(qw0, qw0 + 31, key0, key0 + 63, d.pshift, a.T, d.qf_off);
    if (key0 + 63 >= Lk && cls == 2) cls = 1;
    if (cls == 0 || qw0 >= Lq) continue;
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
      auto dsoft = [&](auto masked_) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_)::value;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          float p = __builtin_amdgcn_exp2f(s[rr] - lse);                     // (q carries log2(e)/8: qkv_norm_kernel)
          if constexpr (MASKED) {
            const int key = key0 + kt * 32 + mfma_row(rr, lane);
            if (key >= Lk || !tok_allowed<MODE>(qrow, key, d.pshift, a.T, d.qf_off)) p = 0.f;
          }
          s[rr] = p * (dp[rr] - delta);                   // dS (the 1/8 of the score scale is applied to dQ in the epilogue)
        }
      };
      if (cls == 2) dsoft(std::false_type{});
      else dsoft(std::true_type{});
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 db = pack8(s, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(ktf[s2][dt], db, dq[dt]);     // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
      }
    }
  }
  if constexpr (KS > 1) {
    float* red = (float*)smem;                     // [QW][32][64] floats
    __syncthreads();
    if (st == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { red[(qwv * 32 + i) * 64 + lane] = dq[0][i]; red[(qwv * 32 + 16 + i) * 64 + lane] = dq[1][i]; }
    }
    __syncthreads();
    if (st != 0) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] += red[(qwv * 32 + i) * 64 + lane]; dq[1][i] += red[(qwv * 32 + 16 + i) * 64 + lane]; }
  }
  if (qrow >= Lq) return;
  bf16* og = (bf16*)a.dq + ((size_t)b * Lq + qrow) * C + head * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 ov;
#pragma unroll
      for (int k = 0; k < 4; ++k) ov[k] = f2bf(dq[dt][4 * g + k] * 0.125f);
      *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h) = ov;
    }
#endif
}

// ================================================================================================================
// backward: dK, dV   (lane = key).  Q / dO tiles (64 queries) and the tile's lse / delta go global -> LDS by LDS-DMA
// into two alternating buffers (one barrier per tile); both tiles are read row-wise (S = Q.K^T, dP = dO.V^T) and
// transposed (dK^T += Q^T.dS, dV^T += dO^T.P) and carry the transposing-read swizzle.
template <int MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TB = 64 * 128, BUFB = 2 * TB + 512;        // Q | dO | lse[64] delta[64]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int nch = a.dkv_chunks > 1 ? a.dkv_chunks : 1;      // query-list chunks per key block (see OnirisAttnArgs)
  int bx_, head, b;
  attn_block_decode(bx_, head, b);
  const int kb = bx_ / nch, chunk = bx_ % nch;
  const int C = a.C, Lq = a.Lq, Lk = a.Lk;
  const int kw0 = kb * 128 + wave * 32;
  const int krow = kw0 + r;

  const bf16* kg = (const bf16*)a.k + (size_t)b * Lk * C + head * 64;
  const bf16* vg = (const bf16*)a.v + (size_t)b * Lk * C + head * 64;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 v = u32x4{0u, 0u, 0u, 0u}, w = u32x4{0u, 0u, 0u, 0u};
    if (krow < Lk) {
      v = *(const u32x4*)(kg + (size_t)krow * C + ks * 16 + h * 8);
      w = *(const u32x4*)(vg + (size_t)krow * C + ks * 16 + h * 8);
    }
    kf[ks] = __builtin_bit_cast(bf16x8, v);
    vf[ks] = __builtin_bit_cast(bf16x8, w);
    asm volatile("" ::"v"(kf[ks]), "v"(vf[ks]));    // consume the ordinary loads before any LDS-DMA is in flight
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

  const int trow = kb >> d.tshift, tmask = (1 << d.tshift) - 1;
  const int nq = a.q_num ? (a.q_num[trow] << d.tshift) : (Lq + 127) / 128;
  const int nsub = nq * 2;
  const int nent = a.q_num ? a.q_num[trow] : 0;      // (the row's block list in one VGPR: see attn_fwd_kernel)
  const int qvl = (a.q_idx && lane < nent) ? a.q_idx[(size_t)trow * a.qtab_cols + lane] : 0;
  asm volatile("" ::"v"(qvl));
  auto q_start = [&](int idx) __attribute__((always_inline)) {
    const int j = idx >> 1, jj = j >> d.tshift;
    int e = j;
    if (a.q_idx) e = ((nent <= 64 ? __builtin_amdgcn_readlane(qvl, jj) : a.q_idx[(size_t)trow * a.qtab_cols + jj]) << d.tshift) + (j & tmask);
    return e * 128 + (idx & 1) * 64;
  };

  constexpr int OOB = (int)0x80000000;
  int tvo[2], prow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = i * 256 + tid, row = e >> 3, pp = e & 7;
    prow[i] = row;
    tvo[i] = (row * C + head * 64 + (pp ^ (4 * ((row >> 1) & 1))) * 8) * 2;
  }
  const i32x4 rs_q = make_rsrc((const bf16*)a.q + (size_t)b * Lq * C, Lq * C * 2);
  const i32x4 rs_do = make_rsrc((const bf16*)a.dout + (size_t)b * Lq * C, Lq * C * 2);
  const i32x4 rs_l = make_rsrc(a.lse + (size_t)(b * a.heads + head) * Lq, Lq * 4);
  const i32x4 rs_d = make_rsrc(a.delta + (size_t)(b * a.heads + head) * Lq, Lq * 4);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  auto issue = [&](int q0, int bsel) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + bsel * BUFB + wave * 1024);
    const int left = Lq - q0, so = q0 * C * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = prow[i] < left;
      dma16(rs_q, ok ? tvo[i] : OOB, so, dst + i * 4096);
      dma16(rs_do, ok ? tvo[i] : OOB, so, dst + TB + i * 4096);
    }
    if (wave == 0) {                               // lanes 0..15: lse[q0..q0+63], lanes 16..31: delta (16 B per lane)
      const unsigned sdst = __builtin_amdgcn_readfirstlane(lds0 + bsel * BUFB + 2 * TB);
      const int l16 = lane & 15;
      const int vo = (l16 * 4 < left) ? l16 * 16 : OOB;
      if (lane < 16) dma16(rs_l, vo, q0 * 4, sdst);
      else if (lane < 32) dma16(rs_d, vo, q0 * 4, sdst);
    }
  };
  // row fragments (rows qt*32 + r, 4-way conflict with this swizzle) and transposed fragments
  const int rr0 = r * 128 + ((h ^ (4 * ((r >> 1) & 1))) << 4);
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);
  const int tb0 = (4 * hh + q4) * 128 + pcol * 2, tsw = (q4 >> 1) & 1;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto ttr = [&](const unsigned char* t_, int tokbase, int dt) __attribute__((always_inline)) {
    const unsigned char* p0 = t_ + tb0 + tokbase * 128 + ((dt ^ tsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 128));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };

  const int i0 = (int)((long long)nsub * chunk / nch), i1 = (int)((long long)nsub * (chunk + 1) / nch);
  int bsel = 0;
  if (i0 < i1) issue(q_start(i0), 0);
#pragma unroll 1
  for (int idx = i0; idx < i1; ++idx) {
    const int q0 = q_start(idx);
    dma_wait();
    __syncthreads();
    if (idx + 1 < i1) issue(q_start(idx + 1), bsel ^ 1);
    const unsigned char* Qt = smem + bsel * BUFB;
    const unsigned char* dOt = Qt + TB;
    const float* lse_lds = (const float*)(Qt + 2 * TB);
    const float* del_lds = lse_lds + 64;
    bsel ^= 1;
    if (kw0 >= Lk || q0 >= Lq) continue;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qq0 = q0 + qt * 32;
      int cls = classify<MODE>(qq0, qq0 + 31, kw0, kw0 + 31, d.pshift, a.T, d.qf_off);
      if ((qq0 + 31 >= Lq || kw0 + 31 >= Lk) && cls == 2) cls = 1;
      if (cls == 0 || qq0 >= Lq) continue;
      bf16x8 qa[4], da[4], dotf[2][2], qtf[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qa[ks] = *(const bf16x8*)(Qt + ((rr0 ^ (ks * 32)) + qt * 4096));
        da[ks] = *(const bf16x8*)(dOt + ((rr0 ^ (ks * 32)) + qt * 4096));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dotf[s2][dt] = ttr(dOt, qt * 32 + 16 * s2, dt);
          qtf[s2][dt] = ttr(Qt, qt * 32 + 16 * s2, dt);
        }
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(qa[ks], kf[ks], s);                   // S[q][key]
        dp = mfma32(da[ks], vf[ks], dp);                 // dP[q][key]
      }
      f32x16 pv;
      auto dsoft = [&](auto masked_) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_)::value;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int qb4 = qt * 32 + 8 * g4 + 4 * h;          // 4 consecutive query rows per accumulator group
          const float4 ls = *(const float4*)(lse_lds + qb4), de = *(const float4*)(del_lds + qb4);
          const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int rr = 4 * g4 + k;
            float p = __builtin_amdgcn_exp2f(s[rr] - lsv[k]);                // (q carries log2(e)/8: qkv_norm_kernel)
            if constexpr (MASKED) {
              const int qtok = q0 + qb4 + k;
              if (qtok >= Lq || krow >= Lk || !tok_allowed<MODE>(qtok, krow, d.pshift, a.T, d.qf_off)) p = 0.f;
            }
            pv[rr] = p;
            s[rr] = p * (dp[rr] - dev[k]) * 0.125f;          // dS (scaled)
          }
        }
      };
      if (cls == 2) dsoft(std::false_type{});
      else dsoft(std::true_type{});
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(pv, s2);
        const bf16x8 db = pack8(s, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = mfma32(dotf[s2][dt], pb, dv[dt]);     // dV^T[dv][key] += dO^T[dv][q] P[q][key]
          dk[dt] = mfma32(qtf[s2][dt], db, dk[dt]);      // dK^T[d][key]  += Q^T[d][q] dS[q][key]
        }
      }
    }
  }
  if (krow >= Lk) return;
  if (nch > 1) {                                     // fp32 partial sums of this chunk (added by attn_dkv_reduce_kernel)
    const size_t plane = (size_t)nch * a.B * Lk * C;
    float* pk = a.dkv_part + (((size_t)chunk * a.B + b) * Lk + krow) * C + head * 64;
    float* pv = pk + plane;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *(float4*)(pk + dt * 32 + 8 * g + 4 * h) = make_float4(dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
        *(float4*)(pv + dt * 32 + 8 * g + 4 * h) = make_float4(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
      }
    return;
  }
  bf16* dkg = (bf16*)a.dk + ((size_t)b * Lk + krow) * C + head * 64;
  bf16* dvg = (bf16*)a.dv + ((size_t)b * Lk + krow) * C + head * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o1, o2;
#pragma unroll
      for (int k = 0; k < 4; ++k) { o1[k] = f2bf(dk[dt][4 * g + k] * (1.f / SCALE_LOG2)); o2[k] = f2bf(dv[dt][4 * g + k]); }   // dK^T was summed against q' = c q
      *(bf16x4*)(dkg + dt * 32 + 8 * g + 4 * h) = o1;
      *(bf16x4*)(dvg + dt * 32 + 8 * g + 4 * h) = o2;
    }
#endif
}

#include "attention_bwd_ws.h"
#include "attention_bwd_dq_ws.h"

// dk|dv = sum over the chunks (in chunk order) of the fp32 partials; 8 elements per thread
__global__ void attn_dkv_reduce_kernel(const float* __restrict__ part, bf16* __restrict__ dk, bf16* __restrict__ dv,
                                       size_t n8, int nch) {      // (dk partials were summed against q' = c q: * 1/c here)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n8) return;
  const int which = i >= n8;
  const size_t e = (which ? i - n8 : i) * 8;
  const size_t n = n8 * 8;
  const float* p = part + (size_t)which * nch * n + e;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int c = 0; c < nch; ++c) {
    const float4 u = *(const float4*)(p + (size_t)c * n), w = *(const float4*)(p + (size_t)c * n + 4);
    acc[0] += u.x; acc[1] += u.y; acc[2] += u.z; acc[3] += u.w;
    acc[4] += w.x; acc[5] += w.y; acc[6] += w.z; acc[7] += w.w;
  }
  const float sc = which ? 1.f : (1.f / SCALE_LOG2);
  bf16x8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = f2bf(acc[k] * sc);
  *(bf16x8*)((which ? dv : dk) + e) = o;
}

// ================================================================================================================
// small HBM-bound helpers

// qkv [tok][3C] (channel = s*C + head*64 + c) -> q,k,v [tok][C], each 64-vector normalised: x / (eps + |x|/8)
// kv_tpb > 0: k and v go to a KV ring -- token r of batch b lands at element b * kv_bstride + (kv_off + r) * C
__global__ void qkv_norm_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ q, bf16* __restrict__ k,
                                bf16* __restrict__ v, long long nvec, int C, long long kv_tpb, long long kv_bstride,
                                long long kv_off) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long vec = gid >> 3;
  const int part = (int)(gid & 7);
  if (vec >= nvec) return;
  const int hpt = 3 * C / 64;                       // vectors per token
  const long long tok = vec / hpt;
  const int vi = (int)(vec % hpt);                  // = s*heads + head
  const int s = vi / (C / 64), hd = vi % (C / 64);
  const bf16x8 x = *(const bf16x8*)(qkv + tok * 3 * C + (size_t)vi * 64 + part * 8);
  float f[8], ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = bf2f(x[i]); ss += f[i] * f[i]; }
  ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
  // q leaves with the softmax scale folded in (one rounding): q' = log2(e)/8 * normalised q, so that every attention kernel
  // gets its log2-domain scores straight out of the MFMA (exp2(q'.k - lse), no multiply per score element)
  const float inv = ((s == 0) ? SCALE_LOG2 : 1.f) / (1e-4f + sqrtf(ss) * 0.125f);
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f2bf(f[i] * inv);
  bf16* dst = (s == 0) ? q : (s == 1) ? k : v;
  long long row = tok * C;
  if (s != 0 && kv_tpb > 0) row = (tok / kv_tpb) * kv_bstride + (kv_off + tok % kv_tpb) * C;
  *(bf16x8*)(dst + row + hd * 64 + part * 8) = o;
}

__global__ void qkv_norm_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dq,
                                    const bf16* __restrict__ dk, const bf16* __restrict__ dv,
                                    bf16* __restrict__ dqkv, long long nvec, int C) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long vec = gid >> 3;
  const int part = (int)(gid & 7);
  if (vec >= nvec) return;
  const int hpt = 3 * C / 64;
  const long long tok = vec / hpt;
  const int vi = (int)(vec % hpt);
  const int s = vi / (C / 64), hd = vi % (C / 64);
  const bf16x8 x = *(const bf16x8*)(qkv + tok * 3 * C + (size_t)vi * 64 + part * 8);
  const bf16* src = (s == 0) ? dq : (s == 1) ? dk : dv;
  const bf16x8 g = *(const bf16x8*)(src + tok * C + hd * 64 + part * 8);
  float f[8], gg[8], ss = 0.f, dot = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = bf2f(x[i]); gg[i] = bf2f(g[i]); ss += f[i] * f[i]; dot += f[i] * gg[i]; }
  ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
  dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
  const float n = sqrtf(ss), sden = 1e-4f + n * 0.125f;
  const float k1 = 1.f / sden, k2 = (n > 0.f) ? dot * 0.125f / (sden * sden * n) : 0.f;
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f2bf(gg[i] * k1 - f[i] * k2);
  *(bf16x8*)(dqkv + tok * 3 * C + (size_t)vi * 64 + part * 8) = o;
}

// qkv_norm_kernel + the rotary embedding of q and k in one pass (training: position = (token / P) mod pos_mod; tables
// [pos][64] fp32: cos, sin, scale, scale duplicated over the two halves like RoPe.py:21-32).  The rotation partner of
// channel c is c ^ 32 = the same register of lane ^ 4 (8 lanes x 8 channels cover a head).  The normalised value is
// rotated in fp32: q, k are rounded to bf16 once (the two-kernel path rounds the normalised vector first).
//   q' = log2(e)/8 * (u cos + rot(u) sin) * scale,   k' = (u cos + rot(u) sin) / scale,   v' = u      (u = x/(eps+|x|/8))
__global__ void qkv_norm_rope_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ q, bf16* __restrict__ k,
                                     bf16* __restrict__ v, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                     const float* __restrict__ scale_t, long long nvec, int C, int P, int pos_mod) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long vec = gid >> 3;
  const int part = (int)(gid & 7);
  if (vec >= nvec) return;                            // (nvec * 8 is a multiple of 64: whole waves leave together)
  const int hpt = 3 * C / 64;
  const long long tok = vec / hpt;
  const int vi = (int)(vec % hpt);
  const int s = vi / (C / 64), hd = vi % (C / 64);
  const bf16x8 x = *(const bf16x8*)(qkv + tok * 3 * C + (size_t)vi * 64 + part * 8);
  float f[8], ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = bf2f(x[i]); ss += f[i] * f[i]; }
  ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
  const float inv = ((s == 0) ? SCALE_LOG2 : 1.f) / (1e-4f + sqrtf(ss) * 0.125f);
  const size_t tb = (size_t)((tok / P) % pos_mod) * 64 + part * 8;
  const float sg = (part < 4) ? -1.f : 1.f;          // rotate_half: [-x2, x1]
  // the three table rows of this lane's 8 channels as 16-byte loads (element-wise 4-byte loads made the pass VMEM-issue
  // bound: 24 table loads per 16 bytes of output, 1.7 TB/s)
  float cs[8], sn[8], sc[8];
  if (s != 2) {
    *(float4*)&cs[0] = *(const float4*)(cos_t + tb); *(float4*)&cs[4] = *(const float4*)(cos_t + tb + 4);
    *(float4*)&sn[0] = *(const float4*)(sin_t + tb); *(float4*)&sn[4] = *(const float4*)(sin_t + tb + 4);
    *(float4*)&sc[0] = *(const float4*)(scale_t + tb); *(float4*)&sc[4] = *(const float4*)(scale_t + tb + 4);
  }
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float u = f[i] * inv;
    const float up = __shfl_xor(u, 4);                // (every lane of the 8-lane group takes part: no divergence above)
    float val = u;
    if (s != 2) {
      val = u * cs[i] + sg * up * sn[i];
      val = (s == 0) ? val * sc[i] : val / sc[i];
    }
    o[i] = f2bf(val);
  }
  bf16* dst = (s == 0) ? q : (s == 1) ? k : v;
  *(bf16x8*)(dst + tok * C + hd * 64 + part * 8) = o;
}

// One NEW frame per sequence of the KV-cached sampler (attention_modules.py:51-70): normalisation of q, k, v and the rotary
// embedding of q and k at the frame's position `pos` (= n_keys - 1: tables row pos) in one pass.  Outputs: q rotated (with
// the softmax scale), k UN-rotated and v into the KV ring behind the committed frames (the cache keeps un-rotated keys:
// every later frame count re-rotates them, RoPe.py:55-57), k rotated into the ring's ROTATED image `kr`, whose committed
// frames were rotated once for this frame count (KVRing.rotate_committed) instead of once per UNet evaluation.
__global__ void qkv_norm_rope_eval_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ q, bf16* __restrict__ k,
                                          bf16* __restrict__ v, bf16* __restrict__ kr, const float* __restrict__ cos_t,
                                          const float* __restrict__ sin_t, const float* __restrict__ scale_t, long long nvec,
                                          int C, long long kv_tpb, long long kv_bstride, long long kv_off, int pos) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long vec = gid >> 3;
  const int part = (int)(gid & 7);
  if (vec >= nvec) return;                            // (nvec * 8 is a multiple of 64: whole waves leave together)
  const int hpt = 3 * C / 64;
  const long long tok = vec / hpt;
  const int vi = (int)(vec % hpt);
  const int s = vi / (C / 64), hd = vi % (C / 64);
  const bf16x8 x = *(const bf16x8*)(qkv + tok * 3 * C + (size_t)vi * 64 + part * 8);
  float f[8], ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = bf2f(x[i]); ss += f[i] * f[i]; }
  ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
  const float inv = ((s == 0) ? SCALE_LOG2 : 1.f) / (1e-4f + sqrtf(ss) * 0.125f);
  const size_t tb = (size_t)pos * 64 + part * 8;
  const float sg = (part < 4) ? -1.f : 1.f;          // rotate_half: [-x2, x1]
  bf16x8 plain, rot;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // like the two-launch path (qkv_norm, then rope on its bf16 output) the rotation sees the bf16-rounded vector, so that
    // the rotated key of a frame is the same number whether it is rotated now or re-rotated from the ring later
    const float u = bf2f(f2bf(f[i] * inv));
    const float up = __shfl_xor(u, 4);
    plain[i] = f2bf(u);
    float val = u;
    if (s != 2) {
      val = u * cos_t[tb + i] + sg * up * sin_t[tb + i];
      const float scl = scale_t[tb + i];
      val = (s == 0) ? val * scl : val / scl;
    }
    rot[i] = f2bf(val);
  }
  const long long ring = (tok / kv_tpb) * kv_bstride + (kv_off + tok % kv_tpb) * C + hd * 64 + part * 8;
  if (s == 0) *(bf16x8*)(q + tok * C + hd * 64 + part * 8) = rot;
  else if (s == 1) { *(bf16x8*)(k + ring) = plain; *(bf16x8*)(kr + ring) = rot; }
  else *(bf16x8*)(v + ring) = plain;
}

// The sampler's attention input in ONE launch (31 evaluations per generated frame, every graph node costs ~5 us): the
// attn_qkv 1x1 convolution of the new frame(s) (MPConv, attention_modules.py:47), the per-head normalisation and -- when
// tables are given -- the rotary embedding at table row `pos`, written where qkv_norm_rope_eval_kernel / qkv_norm_kernel
// write: q [tok][C] (with the softmax scale), k / v dense or behind the committed frames of the KV ring, kr = the rotated
// key into the ring's rotated image.  One workgroup = 128 tokens x one (q|k|v, head) slice of 64 output channels; wave w
// owns tokens 32w..32w+31 (MFMA D[channel][token]: a lane holds 32 of the head's 64 channels of ONE token, lane ^ 32 the
// others; the rotation partner c + 32 of channel c is the other accumulator of the same lane).  The convolution result is
// rounded to bf16 before the normalisation, like the tensor the two-launch path stores in between.
template <int KC>
__global__ __launch_bounds__(256) void qkv_eval_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                       bf16* __restrict__ q, bf16* __restrict__ k, bf16* __restrict__ v,
                                                       bf16* __restrict__ kr, const float* __restrict__ cos_t,
                                                       const float* __restrict__ sin_t, const float* __restrict__ scale_t,
                                                       long long M, int C, int CinP, long long kv_tpb, long long kv_bstride,
                                                       long long kv_off, int pos) {
  // K is staged KC input channels at a time (the whole K for C <= 256: every load of the tile is in flight at once -- the
  // launch is pure latency, ~10 us with four 64-channel rounds of load / barrier / multiply, half of that in one round)
  constexpr int ROWB = KC * 2 + 16;                          // + 16 bytes: conflict-free 16-byte fragment reads
  constexpr int PCS = KC / 8;                                // 16-byte pieces per row
  __shared__ __attribute__((aligned(16))) unsigned char xs[128 * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char ws[64 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const long long m0 = (long long)blockIdx.x * 128;
  // (slot and head from the grid: a division by a run-time value is ~40 instructions in front of the first address of a launch that
  // lasts 17 K cycles; round 6)
  const int hd = blockIdx.y, s = blockIdx.z, vi = s * (int)gridDim.y + hd;
  const bf16* wrow = w + (size_t)vi * 64 * CinP;
  const int mrows = (M - m0 < 128) ? (int)(M - m0) : 128;    // rows beyond M are never read back by a valid token
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  // the rotation's table values depend on the lane only: requested FIRST, so that their round trip runs under the tile loads instead
  // of behind the MFMAs (round 6: the launch is a latency chain, 129 of them make an evaluation)
  const bool rope = cos_t != nullptr && s != 2;
  const size_t tb = (size_t)pos * 64;
  float tc0[16], ts0[16], tc1[16], ts1[16], tq0[16], tq1[16];
  if (rope) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      const float4 a0 = *(const float4*)(cos_t + tb + c), a1 = *(const float4*)(cos_t + tb + c + 32);
      const float4 b0 = *(const float4*)(sin_t + tb + c), b1 = *(const float4*)(sin_t + tb + c + 32);
      const float4 q0 = *(const float4*)(scale_t + tb + c), q1 = *(const float4*)(scale_t + tb + c + 32);
      tc0[4 * g] = a0.x; tc0[4 * g + 1] = a0.y; tc0[4 * g + 2] = a0.z; tc0[4 * g + 3] = a0.w;
      tc1[4 * g] = a1.x; tc1[4 * g + 1] = a1.y; tc1[4 * g + 2] = a1.z; tc1[4 * g + 3] = a1.w;
      ts0[4 * g] = b0.x; ts0[4 * g + 1] = b0.y; ts0[4 * g + 2] = b0.z; ts0[4 * g + 3] = b0.w;
      ts1[4 * g] = b1.x; ts1[4 * g + 1] = b1.y; ts1[4 * g + 2] = b1.z; ts1[4 * g + 3] = b1.w;
      tq0[4 * g] = q0.x; tq0[4 * g + 1] = q0.y; tq0[4 * g + 2] = q0.z; tq0[4 * g + 3] = q0.w;
      tq1[4 * g] = q1.x; tq1[4 * g + 1] = q1.y; tq1[4 * g + 2] = q1.z; tq1[4 * g + 3] = q1.w;
    }
  }
  for (int c0 = 0; c0 < C; c0 += KC) {
    if (c0) __syncthreads();
#pragma unroll
    for (int i = 0; i < 64 * PCS / 256; ++i) {               // weight tile: 64 rows
      const int e = i * 256 + tid, row = e / PCS, pc = e % PCS;
      *(u32x4*)(ws + row * ROWB + pc * 16) = *(const u32x4*)(wrow + (size_t)row * CinP + c0 + pc * 8);
    }
#pragma unroll
    for (int i = 0; i < 128 * PCS / 256; ++i) {              // x tile: 128 rows
      const int e = i * 256 + tid, row = e / PCS, pc = e % PCS;
      if (row < mrows) *(u32x4*)(xs + row * ROWB + pc * 16) = *(const u32x4*)(x + (size_t)(m0 + row) * C + c0 + pc * 8);
    }
    __syncthreads();
    if (wave * 32 < mrows) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        const bf16x8 xf = *(const bf16x8*)(xs + (wave * 32 + r) * ROWB + (ks * 2 + h) * 16);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const bf16x8 wf = *(const bf16x8*)(ws + (n * 32 + r) * ROWB + (ks * 2 + h) * 16);
          acc[n] = mfma32(wf, xf, acc[n]);
        }
      }
    }
  }
  const long long tok = m0 + wave * 32 + r;
  float f[2][16], ss = 0.f;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[n][i] = bf2f(f2bf(acc[n][i])); ss += f[n][i] * f[n][i]; }
  ss += __shfl_xor(ss, 32);
  const float inv = ((s == 0) ? SCALE_LOG2 : 1.f) / (1e-4f + sqrtf(ss) * 0.125f);
  if (tok >= M) return;
  const long long dense = tok * C + hd * 64;
  long long ring = dense;
  if (kv_tpb > 0) {                                          // (one sequence: tok < kv_tpb, no 64-bit division on the way to the stores)
    const long long sq = (tok < kv_tpb) ? 0 : tok / kv_tpb;
    ring = sq * kv_bstride + (kv_off + tok - sq * kv_tpb) * C + hd * 64;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 plain[2], rot[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int i = 4 * g + kk, c = 8 * g + 4 * h + kk;       // channel c (first half) and c + 32 (second half) of the head
      // (as in qkv_norm_rope_eval_kernel the rotation sees the bf16-rounded normalised vector)
      const float u0 = bf2f(f2bf(f[0][i] * inv)), u1 = bf2f(f2bf(f[1][i] * inv));
      plain[0][kk] = f2bf(u0); plain[1][kk] = f2bf(u1);
      float v0 = u0, v1 = u1;
      if (rope) {
        v0 = u0 * tc0[i] - u1 * ts0[i];                       // rotate_half: [-x2, x1]
        v1 = u1 * tc1[i] + u0 * ts1[i];
        const float s0 = tq0[i], s1 = tq1[i];
        v0 = (s == 0) ? v0 * s0 : v0 / s0;
        v1 = (s == 0) ? v1 * s1 : v1 / s1;
      }
      rot[0][kk] = f2bf(v0); rot[1][kk] = f2bf(v1);
    }
    const int co = 8 * g + 4 * h;
    if (s == 0) {
      *(bf16x4*)(q + dense + co) = rot[0]; *(bf16x4*)(q + dense + co + 32) = rot[1];
    } else if (s == 1) {
      *(bf16x4*)(k + ring + co) = plain[0]; *(bf16x4*)(k + ring + co + 32) = plain[1];
      if (kr) { *(bf16x4*)(kr + ring + co) = rot[0]; *(bf16x4*)(kr + ring + co + 32) = rot[1]; }
    } else {
      *(bf16x4*)(v + ring + co) = plain[0]; *(bf16x4*)(v + ring + co + 32) = plain[1];
    }
  }
}

// qkv_eval_kernel for a FEW tiles (one frame per sequence in the cached sampler; round 6, after conv1x1_few.h): a workgroup owns 32 tokens x
// the 64 channels of one (slot, head) and its four waves split the K -- wave w takes the 64-channel rounds w, w + 4, ... with every operand
// loaded global -> registers in MFMA fragment layout (no LDS staging, no barrier in the K loop); the partial tiles meet in LDS once and
// wave 0 runs the epilogue of qkv_eval_kernel (same operations, same order; the conv result sums K in another order).
__global__ __launch_bounds__(256) void qkv_eval_few_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                           bf16* __restrict__ q, bf16* __restrict__ k, bf16* __restrict__ v,
                                                           bf16* __restrict__ kr, const float* __restrict__ cos_t,
                                                           const float* __restrict__ sin_t, const float* __restrict__ scale_t,
                                                           long long M, int C, int CinP, long long kv_tpb, long long kv_bstride,
                                                           long long kv_off, int pos) {
  __shared__ float red[3 * 2 * 16 * 64];           // waves 1..3 -> wave 0: [wave - 1][n-tile][accumulator register][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const long long tok = (long long)blockIdx.x * 32 + r;
  const int hd = blockIdx.y, s = blockIdx.z, vi = s * (int)gridDim.y + hd;
  const bool tvalid = tok < M;
  const bf16* xrow = x + (size_t)(tvalid ? tok : 0) * C;
  const bf16* wrow0 = w + ((size_t)vi * 64 + r) * CinP;
  const bf16* wrow1 = wrow0 + (size_t)32 * CinP;
  const bool rope = cos_t != nullptr && s != 2;
  const size_t tb = (size_t)pos * 64;
  // (wave 0 runs the epilogue: its rotary-table values are requested first, like qkv_eval_kernel's)
  float tc0[16], ts0[16], tc1[16], ts1[16], tq0[16], tq1[16];
  if (rope && wave == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      const float4 a0 = *(const float4*)(cos_t + tb + c), a1 = *(const float4*)(cos_t + tb + c + 32);
      const float4 b0 = *(const float4*)(sin_t + tb + c), b1 = *(const float4*)(sin_t + tb + c + 32);
      const float4 q0 = *(const float4*)(scale_t + tb + c), q1 = *(const float4*)(scale_t + tb + c + 32);
      tc0[4 * g] = a0.x; tc0[4 * g + 1] = a0.y; tc0[4 * g + 2] = a0.z; tc0[4 * g + 3] = a0.w;
      tc1[4 * g] = a1.x; tc1[4 * g + 1] = a1.y; tc1[4 * g + 2] = a1.z; tc1[4 * g + 3] = a1.w;
      ts0[4 * g] = b0.x; ts0[4 * g + 1] = b0.y; ts0[4 * g + 2] = b0.z; ts0[4 * g + 3] = b0.w;
      ts1[4 * g] = b1.x; ts1[4 * g + 1] = b1.y; ts1[4 * g + 2] = b1.z; ts1[4 * g + 3] = b1.w;
      tq0[4 * g] = q0.x; tq0[4 * g + 1] = q0.y; tq0[4 * g + 2] = q0.z; tq0[4 * g + 3] = q0.w;
      tq1[4 * g] = q1.x; tq1[4 * g + 1] = q1.y; tq1[4 * g + 2] = q1.z; tq1[4 * g + 3] = q1.w;
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  const int nround = C >> 6;
#pragma unroll 1
  for (int ch = wave; ch < nround; ch += 4) {
    const int c0 = ch * 64 + h * 8;
    u32x4 w0[4], w1[4], xv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = c0 + ks * 16;
      w0[ks] = *(const u32x4*)(wrow0 + c);
      w1[ks] = *(const u32x4*)(wrow1 + c);
      xv[ks] = tvalid ? *(const u32x4*)(xrow + c) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 xf = __builtin_bit_cast(bf16x8, xv[ks]);
      acc[0] = mfma32(__builtin_bit_cast(bf16x8, w0[ks]), xf, acc[0]);
      acc[1] = mfma32(__builtin_bit_cast(bf16x8, w1[ks]), xf, acc[1]);
    }
  }
  if (wave != 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) red[(((wave - 1) * 2 + n) * 16 + i) * 64 + lane] = acc[n][i];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[n][i] = ((acc[n][i] + red[((0 * 2 + n) * 16 + i) * 64 + lane]) + red[((1 * 2 + n) * 16 + i) * 64 + lane]) +
                  red[((2 * 2 + n) * 16 + i) * 64 + lane];
  float f[2][16], ss = 0.f;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[n][i] = bf2f(f2bf(acc[n][i])); ss += f[n][i] * f[n][i]; }
  ss += __shfl_xor(ss, 32);
  const float inv = ((s == 0) ? SCALE_LOG2 : 1.f) / (1e-4f + sqrtf(ss) * 0.125f);
  if (!tvalid) return;
  const long long dense = tok * C + hd * 64;
  long long ring = dense;
  if (kv_tpb > 0) {
    const long long sq = (tok < kv_tpb) ? 0 : tok / kv_tpb;
    ring = sq * kv_bstride + (kv_off + tok - sq * kv_tpb) * C + hd * 64;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 plain[2], rot[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int i = 4 * g + kk;
      const float u0 = bf2f(f2bf(f[0][i] * inv)), u1 = bf2f(f2bf(f[1][i] * inv));
      plain[0][kk] = f2bf(u0); plain[1][kk] = f2bf(u1);
      float v0 = u0, v1 = u1;
      if (rope) {
        v0 = u0 * tc0[i] - u1 * ts0[i];
        v1 = u1 * tc1[i] + u0 * ts1[i];
        const float s0 = tq0[i], s1 = tq1[i];
        v0 = (s == 0) ? v0 * s0 : v0 / s0;
        v1 = (s == 0) ? v1 * s1 : v1 / s1;
      }
      rot[0][kk] = f2bf(v0); rot[1][kk] = f2bf(v1);
    }
    const int co = 8 * g + 4 * h;
    if (s == 0) {
      *(bf16x4*)(q + dense + co) = rot[0]; *(bf16x4*)(q + dense + co + 32) = rot[1];
    } else if (s == 1) {
      *(bf16x4*)(k + ring + co) = plain[0]; *(bf16x4*)(k + ring + co + 32) = plain[1];
      if (kr) { *(bf16x4*)(kr + ring + co) = rot[0]; *(bf16x4*)(kr + ring + co + 32) = rot[1]; }
    } else {
      *(bf16x4*)(v + ring + co) = plain[0]; *(bf16x4*)(v + ring + co + 32) = plain[1];
    }
  }
}

// adjoint of qkv_norm_rope_kernel: dq (w.r.t. the UNSCALED rotated q, as the attention backward returns it), dk, dv ->
// dqkv.  R^T g = g cos - rot(g sin) and the scale vector is equal in both halves, so it commutes with the rotation.
__global__ void qkv_norm_rope_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dq,
                                         const bf16* __restrict__ dk, const bf16* __restrict__ dv, bf16* __restrict__ dqkv,
                                         const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                         const float* __restrict__ scale_t, long long nvec, int C, int P, int pos_mod) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long vec = gid >> 3;
  const int part = (int)(gid & 7);
  if (vec >= nvec) return;
  const int hpt = 3 * C / 64;
  const long long tok = vec / hpt;
  const int vi = (int)(vec % hpt);
  const int s = vi / (C / 64), hd = vi % (C / 64);
  const bf16x8 x = *(const bf16x8*)(qkv + tok * 3 * C + (size_t)vi * 64 + part * 8);
  const bf16* gsrc = (s == 0) ? dq : (s == 1) ? dk : dv;
  const bf16x8 g = *(const bf16x8*)(gsrc + tok * C + hd * 64 + part * 8);
  const size_t tb = (size_t)((tok / P) % pos_mod) * 64 + part * 8;
  const float sg = (part < 4) ? 1.f : -1.f;          // adjoint of rotate_half
  float cs[8], sn[8], sc[8];                          // (16-byte table loads, see qkv_norm_rope_kernel)
  if (s != 2) {
    *(float4*)&cs[0] = *(const float4*)(cos_t + tb); *(float4*)&cs[4] = *(const float4*)(cos_t + tb + 4);
    *(float4*)&sn[0] = *(const float4*)(sin_t + tb); *(float4*)&sn[4] = *(const float4*)(sin_t + tb + 4);
    *(float4*)&sc[0] = *(const float4*)(scale_t + tb); *(float4*)&sc[4] = *(const float4*)(scale_t + tb + 4);
  }
  float f[8], gg[8], ss = 0.f, dot = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f[i] = bf2f(x[i]);
    float gv = bf2f(g[i]);
    if (s != 2) gv = (s == 0) ? gv * sc[i] : gv / sc[i];
    const float gs = (s != 2) ? gv * sn[i] : 0.f;
    const float gp = __shfl_xor(gs, 4);               // (g sin) of the partner channel
    gg[i] = (s != 2) ? gv * cs[i] + sg * gp : gv;
    ss += f[i] * f[i]; dot += f[i] * gg[i];
  }
  ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
  dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
  const float n = sqrtf(ss), sden = 1e-4f + n * 0.125f;
  const float k1 = 1.f / sden, k2 = (n > 0.f) ? dot * 0.125f / (sden * sden * n) : 0.f;
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f2bf(gg[i] * k1 - f[i] * k2);
  *(bf16x8*)(dqkv + tok * 3 * C + (size_t)vi * 64 + part * 8) = o;
}

// rotary embedding over the frame index (+ optional transposed copy); one workgroup = 64 tokens x 1 head
__global__ __launch_bounds__(256) void rope_kernel(const bf16* __restrict__ x, bf16* __restrict__ xr,
                                                   bf16* __restrict__ xt, const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t, const float* __restrict__ scale_t,
                                                   int mode, int L, int P, int C, int heads, int pos_offset,
                                                   int pos_mod, long long x_bstride, long long xr_bstride) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const int tok0 = blockIdx.x * 64, head = blockIdx.y, b = blockIdx.z;
  const bf16* xg = x + (size_t)b * x_bstride + head * 64;
  {
    const int row = tid >> 2, part = tid & 3;       // 16 channels per thread
    const int tok = tok0 + row;
    bf16x8 v0, v1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v0[i] = f2bf(0.f); v1[i] = f2bf(0.f); }
    if (tok < L) {
      v0 = *(const bf16x8*)(xg + (size_t)tok * C + part * 16);
      v1 = *(const bf16x8*)(xg + (size_t)tok * C + part * 16 + 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { tile[row][part * 16 + i] = bf2f(v0[i]); tile[row][part * 16 + 8 + i] = bf2f(v1[i]); }
  }
  __syncthreads();
  if (mode != 0) {
    const int row = tid >> 2, part = tid & 3;
    const int tok = tok0 + row;
    float out[16];
    if (tok < L) {
      const int pos = pos_offset + ((tok / P) % pos_mod);
      const float* cs = cos_t + (size_t)pos * 64;
      const float* sn = sin_t + (size_t)pos * 64;
      const float* sc = scale_t + (size_t)pos * 64;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = part * 16 + i;
        const float a0 = tile[row][c];
        const float a1 = tile[row][c ^ 32];
        float sg = (c < 32) ? -1.f : 1.f;            // rotate_half: [-x2, x1]
        if (mode >= 3) sg = -sg;                     // adjoint
        float val = a0 * cs[c] + sg * a1 * sn[c];
        const float scl = sc[c];
        val = (mode == 1 || mode == 3) ? val * scl : val / scl;
        out[i] = val;
      }
    }
    __syncthreads();
    if (tok < L) {
#pragma unroll
      for (int i = 0; i < 16; ++i) tile[row][part * 16 + i] = out[i];
    }
    __syncthreads();
  }
  if (xr) {
    const int row = tid >> 2, part = tid & 3;
    const int tok = tok0 + row;
    if (tok < L) {
      bf16x8 v0, v1;
#pragma unroll
      for (int i = 0; i < 8; ++i) { v0[i] = f2bf(tile[row][part * 16 + i]); v1[i] = f2bf(tile[row][part * 16 + 8 + i]); }
      bf16* dst = xr + (size_t)b * xr_bstride + head * 64 + (size_t)tok * C + part * 16;
      *(bf16x8*)dst = v0;
      *(bf16x8*)(dst + 8) = v1;
    }
  }
  if (xt) {
    const int ch = tid >> 2, part = tid & 3;        // 16 tokens per thread
    bf16* dst = xt + ((size_t)(b * heads + head) * 64 + ch) * L + tok0 + part * 16;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (tok0 + part * 16 + hh * 8 < L) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f2bf(tile[part * 16 + hh * 8 + i][ch]);
        *(bf16x8*)(dst + hh * 8) = v;
      }
    }
  }
}

// delta[b][h][tok] = sum_c dO * O
__global__ void attn_delta_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out, float* __restrict__ delta,
                                  const float* __restrict__ lse, float* __restrict__ neg, long long nvec, int L, int C,
                                  int heads) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long vec = gid >> 3;
  const int part = (int)(gid & 7);
  if (vec >= nvec) return;
  const long long tokg = vec / heads;                // b*L + tok
  const int hd = (int)(vec % heads);
  const bf16x8 a = *(const bf16x8*)(dout + tokg * C + hd * 64 + part * 8);
  const bf16x8 o = *(const bf16x8*)(out + tokg * C + hd * 64 + part * 8);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += bf2f(a[i]) * bf2f(o[i]);
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  if (part == 0) {
    const long long bb = tokg / L, tok = tokg % L;
    const long long at = (bb * heads + hd) * L + tok;
    delta[at] = s;
    if (neg) {                                     // row constants of the persistent dK/dV kernel: [0] = -lse, [1] = -delta
      neg[at] = -lse[at];
      neg[nvec + at] = -s;
    }
  }
}

// ================================================================================================================
// host entry points
static int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

static int attn_prepare(const OnirisAttnArgs* args, AttnDev& d, const char* who) {
  if (!args) { oniris_set_error("%s: null args", who); return ONIRIS_EINVAL; }
  d.a = *args;
  const OnirisAttnArgs& a = d.a;
  if (!(a.B > 0 && a.heads > 0 && a.Lq > 0 && a.Lk > 0 && a.C == a.heads * 64)) {
    oniris_set_error("%s: bad sizes (C must be heads*64)", who); return ONIRIS_EINVAL;
  }
  if (a.Lk % 8 != 0 || a.Lq % 8 != 0) { oniris_set_error("%s: Lq, Lk must be multiples of 8", who); return ONIRIS_EINVAL; }
  d.pshift = 0; d.qf_off = 0; d.tshift = 0;
  if (a.kv_num || a.q_num) {
    const int tb = a.tab_block > 0 ? a.tab_block : 128;
    d.tshift = (tb % 128 == 0) ? ilog2_exact(tb / 128) : -1;
    if (d.tshift < 0) { oniris_set_error("%s: table block %d must be 128 * 2^k", who, tb); return ONIRIS_EUNSUPPORTED; }
  }
  if (a.mask_mode != 0) {
    d.pshift = ilog2_exact(a.P);
    if (d.pshift < 0) { oniris_set_error("%s: P=%d must be a power of two", who, a.P); return ONIRIS_EUNSUPPORTED; }
    if (a.mask_mode == 1) d.qf_off = (a.Lk - a.Lq) / a.P;
    if (a.mask_mode == 2 && (a.Lq != a.Lk || a.Lq != 2 * a.T * a.P || (a.T * a.P) % 128 != 0)) {
      oniris_set_error("%s: training mask needs Lq == Lk == 2*T*P and T*P %% 128 == 0", who); return ONIRIS_EINVAL;
    }
  }
  if (a.mask_mode < 0 || a.mask_mode > 2) { oniris_set_error("%s: bad mask_mode", who); return ONIRIS_EINVAL; }
  return ONIRIS_OK;
}

#define ATTN_DISPATCH(KERN, GRID)                                                                     \
  switch (d.a.mask_mode) {                                                                            \
    case 0: ONIRIS_KLAUNCH(KERN<0>, GRID, dim3(256), 0, stream, d); break;                        \
    case 1: ONIRIS_KLAUNCH(KERN<1>, GRID, dim3(256), 0, stream, d); break;                        \
    default: ONIRIS_KLAUNCH(KERN<2>, GRID, dim3(256), 0, stream, d); break;                       \
  }

extern "C" int oniris_attn_fwd(const OnirisAttnArgs* args, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnDev d;
  int rc = attn_prepare(args, d, "attn_fwd");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(d.a.q && d.a.k && d.a.v && d.a.out, "attn_fwd: null pointer");
  ONIRIS_CHECK_ARG(d.a.v_bstride == 0 || (!d.a.sched && d.a.v_bstride >= (int64_t)d.a.Lk * d.a.C),
                   "attn_fwd: v_bstride is served by the grid kernel only and must cover a sequence");
  ONIRIS_CHECK_ARG(d.a.k_bstride == 0 || (!d.a.sched && d.a.k_bstride >= (int64_t)d.a.Lk * d.a.C),
                   "attn_fwd: k_bstride is served by the grid kernel only and must cover a sequence");
  if (d.a.sched) {                                  // persistent, statically balanced, wave-specialised kernel (attention_ws.h)
    ONIRIS_CHECK_ARG(d.a.sched_wgs > 0 && d.a.sched_slots > 0, "attn_fwd: empty schedule");
    ONIRIS_CHECK_ARG(d.a.mask_mode != 0, "attn_fwd: the scheduled kernel serves the table-driven masks");
    ONIRIS_CHECK_ARG(d.a.kv_num && d.a.kv_idx && d.a.tab_cols <= 64, "attn_fwd: the scheduled kernel needs a table with <= 64 blocks per row");
    // the kernel's assumptions (only the LAST listed block of a table row is partially masked; 128-row query blocks):
    // true for the DART training table (Lq == Lk == 2*T*P) and for the causal prefill table (Lq == Lk), nothing else
    ONIRIS_CHECK_ARG(d.a.Lq % 128 == 0 && d.a.Lq == d.a.Lk,
                     "attn_fwd: the scheduled kernel needs Lq == Lk, a multiple of 128 (got %d, %d): launch without a schedule",
                     d.a.Lq, d.a.Lk);
    if (d.a.mask_mode == 1) oniris_launch(attn_fwd_ws_kernel<1>, dim3(d.a.sched_wgs), dim3(512), stream, d);
    else oniris_launch(attn_fwd_ws_kernel<2>, dim3(d.a.sched_wgs), dim3(512), stream, d);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  if (frame_attn_ok(d.a)) {                          // dense attention inside frames of 64 / 128 / 256 tokens (attention_frame.h)
    FrameAttnDev f{d.a, (long long)d.a.B * d.a.Lq, d.a.Lq / 64};
    ONIRIS_CHECK_ARG((f.ntok + 255) / 256 < (1LL << 31), "attn_fwd: too many tokens");
    const long long nsb = (f.ntok + 255) / 256;
    if (d.a.Lq == 256 && nsb * d.a.heads <= 64 && !(d.a.frame_kernel & 4))         // few 256-token frames: two query halves per frame
      oniris_launch(frame_attn_fwd_kernel<1>, dim3((unsigned)(2 * nsb), d.a.heads), dim3(256), stream, f);
    else
      oniris_launch(frame_attn_fwd_kernel<2>, dim3((unsigned)nsb, d.a.heads), dim3(256), stream, f);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  // two key streams per workgroup (64 query rows) when the key lists are long and causal, else one (128 rows)
  const bool split = d.a.mask_mode != 0 && d.a.Lk >= 2048;
  if (d.a.kv_splits > 1) {                           // split-KV decode: partials + reduction (two launches)
    ONIRIS_CHECK_ARG(d.a.mask_mode == 0 && d.a.split_ws && d.a.kv_splits <= 256,
                     "attn_fwd: kv_splits serves the dense (mask_mode 0) kernel and needs split_ws");
    const dim3 gs(cdiv(d.a.Lq, 128) * d.a.kv_splits, d.a.heads, d.a.B);
    ONIRIS_KLAUNCH((attn_fwd_kernel<0, 1>), gs, dim3(256), 0, stream, d);
    ONIRIS_LAUNCH_CHECK();
    const long long nthr = (long long)d.a.B * d.a.heads * d.a.Lq * 8;
    ONIRIS_KLAUNCH(attn_split_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                       (const float*)d.a.split_ws, (bf16*)d.a.out, d.a.lse, d.a.kv_splits, d.a.B, d.a.heads, d.a.Lq, d.a.C);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  // decode of a frame against a short ring (below the split-KV threshold): a handful of workgroups, each a serial walk over the key
  // tiles (~0.5 us per tile: 13 us per layer at 18 cached frames, three times the launch floor) -- four key streams per workgroup
  // of 32 query rows cut that walk by four inside ONE launch (round 6)
  if (d.a.mask_mode == 0 && !(d.a.frame_kernel & 2) && d.a.Lk >= 256 && d.a.Lk > d.a.Lq &&
      (long long)cdiv(d.a.Lq, 32) * d.a.heads * d.a.B <= 512) {
    ONIRIS_KLAUNCH((attn_fwd_kernel<0, 4>), dim3(cdiv(d.a.Lq, 32), d.a.heads, d.a.B), dim3(256), 0, stream, d);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  const dim3 grid(cdiv(d.a.Lq, split ? 64 : 128), d.a.heads, d.a.B);
  if (split) {
    if (d.a.mask_mode == 1) ONIRIS_KLAUNCH((attn_fwd_kernel<1, 2>), grid, dim3(256), 0, stream, d);
    else ONIRIS_KLAUNCH((attn_fwd_kernel<2, 2>), grid, dim3(256), 0, stream, d);
  } else {
    switch (d.a.mask_mode) {
      case 0: ONIRIS_KLAUNCH((attn_fwd_kernel<0, 1>), grid, dim3(256), 0, stream, d); break;
      case 1: ONIRIS_KLAUNCH((attn_fwd_kernel<1, 1>), grid, dim3(256), 0, stream, d); break;
      default: ONIRIS_KLAUNCH((attn_fwd_kernel<2, 1>), grid, dim3(256), 0, stream, d); break;
    }
  }
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_attn_bwd_dq(const OnirisAttnArgs* args, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnDev d;
  int rc = attn_prepare(args, d, "attn_bwd_dq");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(d.a.q && d.a.k && d.a.v && d.a.dout && d.a.lse && d.a.delta && d.a.dq,
                   "attn_bwd_dq: null pointer");
  if (d.a.sched) {                                  // persistent, statically balanced, wave-specialised kernel (attention_bwd_dq_ws.h):
    // the forward's work list (query blocks of 128 rows); lse / delta point at the NEGATED rows (oniris_attn_bwd_prep)
    ONIRIS_CHECK_ARG(d.a.mask_mode != 0 && d.a.kv_num && d.a.kv_idx && d.a.tab_cols <= 64 && d.a.sched_wgs > 0 && d.a.sched_slots > 0,
                     "attn_bwd_dq: the scheduled kernel needs a table-driven mask with <= 64 blocks per row");
    ONIRIS_CHECK_ARG(d.a.Lq % 128 == 0 && d.a.Lq == d.a.Lk && d.a.Lq / 128 < 65536,
                     "attn_bwd_dq: the scheduled kernel needs Lq == Lk, a multiple of 128 (got %d, %d)", d.a.Lq, d.a.Lk);
    // mask_mode 1 + a block-diagonal table = dense attention inside every frame (FrameAttention with P = 128 * 2^k tokens)
    if (d.a.mask_mode == 1) oniris_launch(attn_bwd_dq_ws_kernel<1>, dim3(d.a.sched_wgs), dim3(512), stream, d);
    else oniris_launch(attn_bwd_dq_ws_kernel<2>, dim3(d.a.sched_wgs), dim3(512), stream, d);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  if (frame_attn_ok(d.a)) {                          // dense attention inside frames of 64 / 128 / 256 tokens (attention_frame.h)
    FrameAttnDev f{d.a, (long long)d.a.B * d.a.Lq, d.a.Lq / 64};
    oniris_launch(frame_attn_dq_kernel, dim3((unsigned)((f.ntok + 255) / 256), d.a.heads), dim3(256), stream, f);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  const bool split = d.a.mask_mode != 0 && d.a.Lk >= 2048;
  const dim3 grid(cdiv(d.a.Lq, split ? 64 : 128), d.a.heads, d.a.B);
  if (split) {
    if (d.a.mask_mode == 1) ONIRIS_KLAUNCH((attn_bwd_dq_kernel<1, 2>), grid, dim3(256), 0, stream, d);
    else ONIRIS_KLAUNCH((attn_bwd_dq_kernel<2, 2>), grid, dim3(256), 0, stream, d);
  } else {
    switch (d.a.mask_mode) {
      case 0: ONIRIS_KLAUNCH((attn_bwd_dq_kernel<0, 1>), grid, dim3(256), 0, stream, d); break;
      case 1: ONIRIS_KLAUNCH((attn_bwd_dq_kernel<1, 1>), grid, dim3(256), 0, stream, d); break;
      default: ONIRIS_KLAUNCH((attn_bwd_dq_kernel<2, 1>), grid, dim3(256), 0, stream, d); break;
    }
  }
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_frame_attn_bwd(const OnirisAttnArgs* args, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnDev d;
  int rc = attn_prepare(args, d, "frame_attn_bwd");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(d.a.q && d.a.k && d.a.v && d.a.out && d.a.dout && d.a.lse && d.a.dq && d.a.dk && d.a.dv, "frame_attn_bwd: null pointer");
  OnirisAttnArgs chk = d.a;
  chk.frame_kernel = 0;
  ONIRIS_CHECK_ARG(frame_attn_ok(chk), "frame_attn_bwd: dense attention inside frames of 64 / 128 / 256 tokens only (mask_mode 0, Lq == Lk, "
                                       "no table / schedule / ring strides / split)");
  FrameAttnDev f{d.a, (long long)d.a.B * d.a.Lq, d.a.Lq / 64};
  ONIRIS_CHECK_ARG((f.ntok + 255) / 256 < (1LL << 31), "frame_attn_bwd: too many tokens");
  oniris_launch(frame_attn_bwd_kernel, dim3((unsigned)((f.ntok + 255) / 256), d.a.heads), dim3(512), stream, f);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

static int frame_qkv_dev(FrameQkvDev& f, const void* qkv, void* out, float* lse, const void* dout, void* dqkv, int64_t n_frames, int P,
                         int heads, const char* who) {
  ONIRIS_CHECK_ARG(qkv && out && lse && n_frames > 0 && heads > 0, "%s: bad arguments", who);
  ONIRIS_CHECK_ARG(P == 64 || P == 128 || P == 256, "%s: frames of 64 / 128 / 256 tokens (got %d)", who, P);
  f.qkv = (const bf16*)qkv; f.out = (bf16*)out; f.lse = lse; f.dout = (const bf16*)dout; f.dqkv = (bf16*)dqkv;
  f.ntok = (long long)n_frames * P; f.P = P; f.heads = heads; f.C = heads * 64; f.tiles = P / 64;
  ONIRIS_CHECK_ARG((f.ntok + 255) / 256 < (1LL << 31) && heads < 65536, "%s: too many tokens", who);
  return ONIRIS_OK;
}

extern "C" int oniris_frame_attn_qkv_fwd(const void* qkv, void* out, float* lse, int64_t n_frames, int P, int heads,
                                         oniris_stream_t stream_) {
  FrameQkvDev f{};
  int rc = frame_qkv_dev(f, qkv, out, lse, nullptr, nullptr, n_frames, P, heads, "frame_attn_qkv_fwd");
  if (rc) return rc;
  oniris_launch(frame_attn_qkv_fwd_kernel, dim3((unsigned)((f.ntok + 255) / 256), heads), dim3(256), (hipStream_t)stream_, f);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

#ifdef FRAME_STAMP
extern "C" int oniris_frame_attn_qkv_fwd_stamp(const void* qkv, void* out, float* lse, void* stamps, int64_t n_frames, int P, int heads,
                                               oniris_stream_t stream_) {
  FrameQkvDev f{};
  int rc = frame_qkv_dev(f, qkv, out, lse, nullptr, stamps, n_frames, P, heads, "frame_attn_qkv_fwd_stamp");
  if (rc) return rc;
  oniris_launch(frame_attn_qkv_fwd_kernel, dim3((unsigned)((f.ntok + 255) / 256), heads), dim3(256), (hipStream_t)stream_, f);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}
#endif

extern "C" int oniris_frame_attn_qkv_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, int64_t n_frames,
                                         int P, int heads, oniris_stream_t stream_) {
  FrameQkvDev f{};
  int rc = frame_qkv_dev(f, qkv, (void*)out, (float*)lse, dout, dqkv, n_frames, P, heads, "frame_attn_qkv_bwd");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(dout && dqkv, "frame_attn_qkv_bwd: dout / dqkv");
  oniris_launch(frame_attn_qkv_bwd_kernel, dim3((unsigned)((f.ntok + 255) / 256), heads), dim3(512), (hipStream_t)stream_, f);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_attn_bwd_dkv(const OnirisAttnArgs* args, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnDev d;
  int rc = attn_prepare(args, d, "attn_bwd_dkv");
  if (rc) return rc;
  ONIRIS_CHECK_ARG(d.a.q && d.a.k && d.a.v && d.a.dout && d.a.lse && d.a.delta && d.a.dk && d.a.dv,
                   "attn_bwd_dkv: null pointer");
  if (d.a.sched) {                                  // persistent, statically balanced, wave-specialised kernel (attention_bwd_ws.h):
    // items = 64-key blocks with their whole query list (schedule built over Lk / 64 items per pair); no partial sums
    ONIRIS_CHECK_ARG(d.a.sched_wgs > 0 && d.a.sched_slots > 0, "attn_bwd_dkv: empty schedule");
    ONIRIS_CHECK_ARG(d.a.mask_mode != 0, "attn_bwd_dkv: the scheduled kernel serves the table-driven masks");
    ONIRIS_CHECK_ARG(d.a.q_num && d.a.q_idx && d.a.qtab_cols <= 64,
                     "attn_bwd_dkv: the scheduled kernel needs the transposed table with <= 64 blocks per row");
    ONIRIS_CHECK_ARG(d.a.Lq % 128 == 0 && d.a.Lq == d.a.Lk && d.a.Lk / 64 < 65536,
                     "attn_bwd_dkv: the scheduled kernel needs Lq == Lk, a multiple of 128 (got %d, %d)", d.a.Lq, d.a.Lk);
    ONIRIS_CHECK_ARG(d.a.dkv_item_keys == 0 || d.a.dkv_item_keys == 64 || d.a.dkv_item_keys == 128,
                     "attn_bwd_dkv: dkv_item_keys is 0 / 64 or 128");
    if (d.a.dkv_item_keys == 128) {
      if (d.a.mask_mode == 1) oniris_launch(attn_bwd_dkv_ws_kernel<1, 128>, dim3(d.a.sched_wgs), dim3(512), stream, d);
      else oniris_launch(attn_bwd_dkv_ws_kernel<2, 128>, dim3(d.a.sched_wgs), dim3(512), stream, d);
    } else {
      if (d.a.mask_mode == 1) oniris_launch(attn_bwd_dkv_ws_kernel<1, 64>, dim3(d.a.sched_wgs), dim3(512), stream, d);
      else oniris_launch(attn_bwd_dkv_ws_kernel<2, 64>, dim3(d.a.sched_wgs), dim3(512), stream, d);
    }
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  const int nch = d.a.dkv_chunks > 1 ? d.a.dkv_chunks : 1;
  ONIRIS_CHECK_ARG(nch == 1 || d.a.dkv_part, "attn_bwd_dkv: dkv_chunks > 1 needs the dkv_part scratch");
  ONIRIS_CHECK_ARG(nch <= 64, "attn_bwd_dkv: at most 64 chunks");
  const dim3 grid(cdiv(d.a.Lk, 128) * nch, d.a.heads, d.a.B);
  ATTN_DISPATCH(attn_bwd_dkv_kernel, grid);
  ONIRIS_LAUNCH_CHECK();
  if (nch > 1) {
    const size_t n8 = (size_t)d.a.B * d.a.Lk * d.a.C / 8;
    ONIRIS_KLAUNCH(attn_dkv_reduce_kernel, dim3((unsigned)((2 * n8 + 255) / 256)), dim3(256), 0, stream,
                       (const float*)d.a.dkv_part, (bf16*)d.a.dk, (bf16*)d.a.dv, n8, nch);
    ONIRIS_LAUNCH_CHECK();
  }
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_norm_rope(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                                    const float* scale_t, int64_t n_tokens, int C, int P, int pos_mod, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && q && k && v && cos_t && sin_t && scale_t && n_tokens > 0 && C > 0 && C % 64 == 0 && P > 0 && pos_mod > 0,
                   "qkv_norm_rope: bad arguments");
  const long long nvec = (long long)n_tokens * 3 * C / 64;
  ONIRIS_KLAUNCH(qkv_norm_rope_kernel, dim3((unsigned)((nvec * 8 + 255) / 256)), dim3(256), 0, stream, (const bf16*)qkv,
                     (bf16*)q, (bf16*)k, (bf16*)v, cos_t, sin_t, scale_t, nvec, C, P, pos_mod);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_norm_rope_eval(const void* qkv, void* q, void* k, void* v, void* kr, const float* cos_t,
                                         const float* sin_t, const float* scale_t, int64_t n_tokens, int C,
                                         int64_t kv_tokens_per_batch, int64_t kv_batch_stride, int64_t kv_token_offset, int pos,
                                         oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && q && k && v && kr && cos_t && sin_t && scale_t && n_tokens > 0 && C % 64 == 0 && pos >= 0 &&
                   kv_tokens_per_batch > 0 && n_tokens % kv_tokens_per_batch == 0 && n_tokens % 8 == 0 &&
                   kv_batch_stride >= (kv_token_offset + kv_tokens_per_batch) * C, "qkv_norm_rope_eval: bad arguments");
  const long long nvec = (long long)n_tokens * (3 * C / 64);
  ONIRIS_KLAUNCH(qkv_norm_rope_eval_kernel, dim3((unsigned)((nvec * 8 + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)qkv, (bf16*)q, (bf16*)k, (bf16*)v, (bf16*)kr, cos_t, sin_t, scale_t, nvec, C,
                     (long long)kv_tokens_per_batch, (long long)kv_batch_stride, (long long)kv_token_offset, pos);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_eval(const void* x, const void* w, void* q, void* k, void* v, void* kr, const float* cos_t,
                               const float* sin_t, const float* scale_t, int64_t n_tokens, int C, int CinP,
                               int64_t kv_tokens_per_batch, int64_t kv_batch_stride, int64_t kv_token_offset, int pos,
                               oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(x && w && q && k && v && n_tokens > 0 && C > 0 && C % 64 == 0 && CinP >= C && CinP % 8 == 0 && pos >= 0,
                   "qkv_eval: bad arguments");
  ONIRIS_CHECK_ARG((cos_t && sin_t && scale_t) || (!cos_t && !sin_t && !scale_t && !kr), "qkv_eval: all rotary tables or none");
  ONIRIS_CHECK_ARG(kv_tokens_per_batch == 0 || (kv_tokens_per_batch > 0 && n_tokens % kv_tokens_per_batch == 0 &&
                                                kv_batch_stride >= (kv_token_offset + kv_tokens_per_batch) * C),
                   "qkv_eval: bad KV ring geometry");
  // a few tiles (one frame per sequence; at most one workgroup per CU): 32-token tiles, the four waves of a workgroup split the K
  // (qkv_eval_few_kernel; ONIRIS_QKV_EVAL_FEW=0 in the environment: off, A/B)
  static const int qkv_few = getenv("ONIRIS_QKV_EVAL_FEW") ? atoi(getenv("ONIRIS_QKV_EVAL_FEW")) : 1;
  if (qkv_few && ((n_tokens + 31) / 32) * (3LL * C / 64) <= 256) {
    const dim3 gf((unsigned)((n_tokens + 31) / 32), (unsigned)(C / 64), 3u);
    ONIRIS_KLAUNCH(qkv_eval_few_kernel, gf, dim3(256), 0, stream, (const bf16*)x, (const bf16*)w, (bf16*)q, (bf16*)k, (bf16*)v, (bf16*)kr,
                   cos_t, sin_t, scale_t, (long long)n_tokens, C, CinP, (long long)kv_tokens_per_batch, (long long)kv_batch_stride,
                   (long long)kv_token_offset, pos);
    ONIRIS_LAUNCH_CHECK();
    return ONIRIS_OK;
  }
  const dim3 grid((unsigned)((n_tokens + 127) / 128), (unsigned)(C / 64), 3u);        // (x: token tile, y: head, z: q | k | v)
#define QKV_EVAL_LAUNCH(KC_)                                                                                              \
  ONIRIS_KLAUNCH(qkv_eval_kernel<KC_>, grid, dim3(256), 0, stream, (const bf16*)x, (const bf16*)w, (bf16*)q, (bf16*)k,  \
                     (bf16*)v, (bf16*)kr, cos_t, sin_t, scale_t, (long long)n_tokens, C, CinP,                             \
                     (long long)kv_tokens_per_batch, (long long)kv_batch_stride, (long long)kv_token_offset, pos)
  if (C % 256 == 0) QKV_EVAL_LAUNCH(256);
  else if (C % 128 == 0) QKV_EVAL_LAUNCH(128);
  else QKV_EVAL_LAUNCH(64);
#undef QKV_EVAL_LAUNCH
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_norm_rope_bwd(const void* qkv, const void* dq, const void* dk, const void* dv, void* dqkv,
                                        const float* cos_t, const float* sin_t, const float* scale_t, int64_t n_tokens, int C,
                                        int P, int pos_mod, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && dq && dk && dv && dqkv && cos_t && sin_t && scale_t && n_tokens > 0 && C > 0 && C % 64 == 0 && P > 0 &&
                   pos_mod > 0, "qkv_norm_rope_bwd: bad arguments");
  const long long nvec = (long long)n_tokens * 3 * C / 64;
  ONIRIS_KLAUNCH(qkv_norm_rope_bwd_kernel, dim3((unsigned)((nvec * 8 + 255) / 256)), dim3(256), 0, stream, (const bf16*)qkv,
                     (const bf16*)dq, (const bf16*)dk, (const bf16*)dv, (bf16*)dqkv, cos_t, sin_t, scale_t, nvec, C, P, pos_mod);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_norm(const void* qkv, void* q, void* k, void* v, int64_t n_tokens, int C,
                               int64_t kv_tokens_per_batch, int64_t kv_batch_stride, int64_t kv_token_offset,
                               oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && q && k && v && n_tokens > 0 && C > 0 && C % 64 == 0, "qkv_norm: bad arguments");
  ONIRIS_CHECK_ARG(kv_tokens_per_batch == 0 || (kv_tokens_per_batch > 0 && n_tokens % kv_tokens_per_batch == 0 &&
                   kv_token_offset >= 0 && kv_batch_stride >= (kv_token_offset + kv_tokens_per_batch) * C),
                   "qkv_norm: bad KV ring description");
  const long long nvec = (long long)n_tokens * 3 * C / 64;
  const long long nthr = nvec * 8;
  ONIRIS_KLAUNCH(qkv_norm_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, (const bf16*)qkv,
                     (bf16*)q, (bf16*)k, (bf16*)v, nvec, C, (long long)kv_tokens_per_batch, (long long)kv_batch_stride,
                     (long long)kv_token_offset);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_qkv_norm_bwd(const void* qkv, const void* dq, const void* dk, const void* dv, void* dqkv,
                                   int64_t n_tokens, int C, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(qkv && dq && dk && dv && dqkv && n_tokens > 0 && C > 0 && C % 64 == 0, "qkv_norm_bwd: bad arguments");
  const long long nvec = (long long)n_tokens * 3 * C / 64;
  const long long nthr = nvec * 8;
  ONIRIS_KLAUNCH(qkv_norm_bwd_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)qkv, (const bf16*)dq, (const bf16*)dk, (const bf16*)dv, (bf16*)dqkv, nvec, C);
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_rope(const void* x, void* xr, void* xt, const float* cos_t, const float* sin_t,
                           const float* scale_t, int mode, int B, int frames, int P, int C, int pos_offset, int pos_mod,
                           int64_t x_batch_stride, int64_t xr_batch_stride, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(x && (xr || xt) && B > 0 && frames > 0 && P > 0 && C % 64 == 0 && mode >= 0 && mode <= 4,
                   "rope: bad arguments");
  ONIRIS_CHECK_ARG(mode == 0 || (cos_t && sin_t && scale_t && pos_mod > 0), "rope: tables missing");
  const int L = frames * P;
  ONIRIS_CHECK_ARG(L % 8 == 0, "rope: frames*P must be a multiple of 8");
  ONIRIS_CHECK_ARG(x_batch_stride == 0 || x_batch_stride >= (int64_t)L * C, "rope: batch stride smaller than a sequence");
  ONIRIS_CHECK_ARG(xr_batch_stride == 0 || xr_batch_stride >= (int64_t)L * C, "rope: output batch stride smaller than a sequence");
  const dim3 grid(cdiv(L, 64), C / 64, B);
  ONIRIS_KLAUNCH(rope_kernel, grid, dim3(256), 0, stream, (const bf16*)x, (bf16*)xr, (bf16*)xt, cos_t, sin_t,
                     scale_t, mode, L, P, C, C / 64, pos_offset, pos_mod > 0 ? pos_mod : 1,
                     (long long)(x_batch_stride > 0 ? x_batch_stride : (int64_t)L * C),
                     (long long)(xr_batch_stride > 0 ? xr_batch_stride : (int64_t)L * C));
  ONIRIS_LAUNCH_CHECK();
  return ONIRIS_OK;
}

extern "C" int oniris_attn_bwd_prep(const void* dout, const void* out, float* delta, void* doutt, const float* lse, float* neg,
                                    int B, int heads, int L, int C, oniris_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ONIRIS_CHECK_ARG(dout && out && delta && B > 0 && heads > 0 && L > 0 && C == heads * 64, "attn_bwd_prep: bad arguments");
  ONIRIS_CHECK_ARG(!neg || lse, "attn_bwd_prep: the negated row constants need lse");
  const long long nvec = (long long)B * L * heads;
  ONIRIS_KLAUNCH(attn_delta_kernel, dim3((unsigned)((nvec * 8 + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)dout, (const bf16*)out, delta, lse, neg, nvec, L, C, heads);
  ONIRIS_LAUNCH_CHECK();
  if (doutt) return oniris_rope(dout, nullptr, doutt, nullptr, nullptr, nullptr, 0, B, L, 1, C, 0, 1, 0, 0, stream_);
  return ONIRIS_OK;
}
