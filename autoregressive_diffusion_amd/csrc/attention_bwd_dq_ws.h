// VideoAttention backward, dQ for gfx950: persistent, statically balanced, wave-specialised, three-stage software
// pipeline (included by attention.hip; the scheduled path of oniris_attn_bwd_dq).  Same launch shape, work list and
// loader protocol as the forward kernel (attention_ws.h): one 512-thread workgroup per CU walks 128-row query blocks
// (oniris_attn_schedule: weight = key blocks of the table row + 1).
//
// Waves 4..7 = LOADERS: K | V blocks of 128 keys (32 KB) through a four-slot ring, three blocks ahead.  K carries the
// dual-use image (chunk c of row R at c ^ f(R), f = bit1 << 2 | bit3 << 1 | bit2: conflict-free for the row reads of
// S^T = K.Q^T AND the transposing reads of dQ^T += K^T.dS^T, see attention_bwd_ws.h), V the row-read swizzle.
//
// Waves 0..3 = COMPUTE, wave w = query rows [32 w, 32 w + 32) of the block against EVERY key (no key split: dQ of a row
// is finished by one lane group, nothing is merged).  Q / dO fragments and the row constants come straight from global
// memory at the top of an item (two items per workgroup at the gym shape).  The unit of work is a micro-step of 32 keys:
//   A: S'^T = K.Q^T - lse, dP^T = V.dO^T            (8 MFMAs; -lse, from oniris_attn_bwd_prep's `neg` rows, is the S chain's
//      initial accumulator; the mask of the list's LAST block enters the same way as -1e30)
//   V: dS^T = exp2(S') * (dP - delta)                (16 exponentials, 16 adds, 16 multiplies, 8 converts per lane)
//   C: dQ^T += K^T.dS^T                              (4 MFMAs, A operand = transposing reads of the K image)
// Pipeline slot u issues  A(u + 2) | V(u + 1) | C(u): its twelve MFMAs depend only on earlier slots, so the matrix pipe runs
// while the VALU does the exponentials in between (one compute wave per SIMD: nothing else would overlap them).  The K / V
// row fragments of micro-step u + 3 and the transposed fragments of u + 1 are read from LDS behind the MFMAs that consumed
// the previous ones.  q arrives with log2(e)/8 folded in (qkv_norm_rope_kernel); the 1/8 of the score scale is applied to dQ
// in the epilogue (dq is the gradient w.r.t. the UNSCALED normalised q, what oniris_qkv_norm[_rope]_bwd takes).
// OnirisAttnArgs.lse / .delta of THIS kernel point at the NEGATED rows.
#pragma once

template <int MODE>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_ws_kernel(const AttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int SLOT = 2 * 128 * 128;              // K [128 keys][128 B] | V [128 keys][128 B]
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * SLOT];
  const OnirisAttnArgs& a = d.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C, Lq = a.Lq, Lk = a.Lk;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  const int nslots = a.sched_slots;
  const int32_t* sched = a.sched + (size_t)blockIdx.x * nslots;
  auto item_at = [&](int i) __attribute__((always_inline)) { return i < nslots ? __builtin_amdgcn_readfirstlane(sched[i]) : -1; };
  const int tmask = (1 << d.tshift) - 1;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    const int drow = 8 * lw + (lane >> 3), dpp = lane & 7;        // piece (lw + 4 i): row 32 i + drow, 16-byte part dpp
    const int ksw = (dpp ^ ((((drow >> 1) & 1) << 2) | (((drow >> 3) & 1) << 1) | ((drow >> 2) & 1))) * 16;
    const int vsw_ = (dpp ^ ((drow >> 1) & 7)) * 16;
    constexpr int OOB = (int)0x80000000;
    struct Src { i32x4 rs_k, rs_v; int kvo, vvo, nblk, kvl; };
    auto open_item = [&](int itm) __attribute__((always_inline)) {
      Src s;
      const int pair = itm >> 16, qb128 = itm & 0xffff;
      const int b = pair / a.heads, head = pair - b * a.heads;
      const int trow = qb128 >> d.tshift;
      const int nent = __builtin_amdgcn_readfirstlane(a.kv_num[trow]);
      s.nblk = nent << d.tshift;
      s.kvl = (lane < nent) ? a.kv_idx[(size_t)trow * a.tab_cols + lane] : 0;
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(s.kvl)::"memory");     // (hipcc does not count the asm DMAs: wait by hand)
      s.rs_k = make_rsrc((const bf16*)a.k + (size_t)b * Lk * C, Lk * C * 2);
      s.rs_v = make_rsrc((const bf16*)a.v + (size_t)b * Lk * C, Lk * C * 2);
      s.kvo = (drow * C + head * 64) * 2 + ksw;
      s.vvo = (drow * C + head * 64) * 2 + vsw_;
      return s;
    };
    auto issue_block = [&](const Src& s, int j) __attribute__((always_inline)) {       // block j of the list -> slot j % 4
      const int key0 = (((__builtin_amdgcn_readlane(s.kvl, j >> d.tshift)) << d.tshift) + (j & tmask)) * 128;
      const unsigned dst = lds0 + (j & 3) * SLOT + lw * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = drow + 32 * i < Lk - key0;
        const int so = (key0 + 32 * i) * C * 2;
        dma16(s.rs_k, ok ? s.kvo : OOB, so, dst + i * 4096);
        dma16(s.rs_v, ok ? s.vvo : OOB, so, dst + i * 4096 + 16384);
      }
    };
    int item = item_at(0);
    if (item < 0) return;
    Src cur = open_item(item);
#pragma unroll 1
    for (int j = 0; j < 3 && j < cur.nblk; ++j) issue_block(cur, j);
#pragma unroll 1
    for (int it = 0;; ++it) {
      const int nblk = cur.nblk;
      // requests so far, in order: blocks 0 .. min(nblk, 3) - 1.  barrier_j needs blocks <= j + 1 landed.
#pragma unroll 1
      for (int j = 0; j < nblk; ++j) {
        if (j + 2 < nblk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // block j + 2 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // barrier_j: block j + 1 landed; block j - 1 released
        if (j + 3 < nblk) issue_block(cur, j + 3);
      }
      const int next = item_at(it + 1);
      Src nxt = cur;
      if (next >= 0) nxt = open_item(next);        // (nothing of this item is in flight any more)
      __syncthreads();                             // E1: the compute waves are done with the ring
      if (next >= 0) {
#pragma unroll 1
        for (int j = 0; j < 3 && j < nxt.nblk; ++j) issue_block(nxt, j);
      }
      if (next < 0) break;
      cur = nxt;
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- compute waves
  const int r = lane & 31, h = lane >> 5;
  // fragment addresses inside a slot: row reads of K (dual-use image) and V, micro-step kt = rows 32 kt + r
  const int kr0 = r * 128 + ((h ^ ((((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1))) << 4);   // k-step ks: ^ (ks * 32)
  const int vr0 = 16384 + r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
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

  int item = item_at(0);
#pragma unroll 1
  for (int it = 0; item >= 0; ++it) {
    const int pair = item >> 16, qb128 = item & 0xffff;
    const int b = pair / a.heads, head = pair - b * a.heads;
    // per-item copy of the lane id for everything outside the block loop (see attention_ws.h: keeps hipcc from hoisting
    // lane-constant addresses out of the item loop and spilling them around the block loop)
    int le = lane;
    asm volatile("" : "+v"(le));
    const int r_ = le & 31, h_ = le >> 5;
    const int qrow = qb128 * 128 + wave * 32 + r_;                  // (Lq is a multiple of 128: every row exists)
    const int trow = qb128 >> d.tshift;
    const int nblk = __builtin_amdgcn_readfirstlane(a.kv_num[trow]) << d.tshift;
    const int last_e = __builtin_amdgcn_readfirstlane(a.kv_idx[(size_t)trow * a.tab_cols + ((nblk - 1) >> d.tshift)]);
    const int last_key0 = ((last_e << d.tshift) + ((nblk - 1) & tmask)) * 128;

    bf16x8 qf[4], dof[4];
    {
      const bf16* qg = (const bf16*)a.q + ((size_t)b * Lq + qrow) * C + head * 64;
      const bf16* dog = (const bf16*)a.dout + ((size_t)b * Lq + qrow) * C + head * 64;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(qg + ks * 16 + h_ * 8));
        dof[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(dog + ks * 16 + h_ * 8));
      }
    }
    const float nl = a.lse[(size_t)(b * a.heads + head) * Lq + qrow];        // -lse, -delta (see the header)
    const float nd = a.delta[(size_t)(b * a.heads + head) * Lq + qrow];
    f32x16 dq[2], nl16;                            // (-lse as the ready-made initial accumulator of the S chain; -delta is added
#pragma unroll                                     //  in stage V: a second constant vector does not fit the register budget)
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; nl16[i] = nl; }

    bf16x8 kf[4], vf[4], ktf[2][2], pA[2], pB[2];
    f32x16 sA, dA, sB, dB;
    auto load_kv = [&](const unsigned char* S0, int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8*)(S0 + ((kr0 ^ (ks * 32)) + kt * 4096));
        vf[ks] = *(const bf16x8*)(S0 + ((vr0 ^ (ks * 32)) + kt * 4096));
      }
    };
    auto load_kt = [&](const unsigned char* S0, int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ktf[s2][dt] = ktr(S0, kt * 32 + 16 * s2, dt);
    };
    // stage A of micro-step kt: both chains start from the row constants; in the list's last block the mask enters as
    // -1e30 (an accumulator register group rr = 4 g .. 4 g + 3 holds 4 consecutive, 4-aligned keys: one frame and one side
    // of Lk, so the mask is evaluated once per group)
    auto stageA = [&](f32x16& s, f32x16& dp, auto masked_, int kt) __attribute__((always_inline)) {
      if constexpr (decltype(masked_)::value) {
        f32x16 m;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key = last_key0 + 32 * kt + 8 * g + 4 * h_;
          const float v = (key < Lk && tok_allowed<MODE>(qrow, key, d.pshift, a.T, d.qf_off)) ? nl : NEG_BIG;
#pragma unroll
          for (int k = 0; k < 4; ++k) m[4 * g + k] = v;
        }
        s = mfma32(kf[0], qf[0], m);
      } else {
        s = mfma32(kf[0], qf[0], nl16);
      }
      dp = mfma32(vf[0], dof[0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        s = mfma32(kf[ks], qf[ks], s);
        dp = mfma32(vf[ks], dof[ks], dp);
      }
    };
    auto stageV = [&](const f32x16& s, const f32x16& dp, bf16x8 (&pb)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[s2][e] = f2bf(__builtin_amdgcn_exp2f(s[8 * s2 + e]) * (dp[8 * s2 + e] + nd));
    };
    auto stageC = [&](bf16x8 (&pb)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(ktf[s2][dt], pb[s2], dq[dt]);   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
    };

    __syncthreads();                               // barrier_0: blocks 0 and 1 landed
    // pipeline fill: A(0), A(1), V(0); fragments of micro-step 2 and the transposed ones of micro-step 0 requested
    load_kv(smem, 0);
    if (nblk == 1) stageA(sA, dA, std::true_type{}, 0); else stageA(sA, dA, std::false_type{}, 0);
    load_kv(smem, 1);
    if (nblk == 1) stageA(sB, dB, std::true_type{}, 1); else stageA(sB, dB, std::false_type{}, 1);
    stageV(sA, dA, pA);
    load_kv(smem, 2);
    load_kt(smem, 0);

    // One block = four slots.  Slot u:  A(u + 2) | V(u + 1) | C(u); micro-steps 4, 5 are 0, 1 of block j + 1.
    // NEXT: 0 = block j + 1 exists and is not the last, 1 = block j + 1 is the last (masked), 2 = block j is the last.
    auto block = [&](auto next_, int j) __attribute__((always_inline)) {
      constexpr int NEXT = decltype(next_)::value;
      const unsigned char* S0 = smem + (j & 3) * SLOT;
      const unsigned char* S1 = smem + ((j + 1) & 3) * SLOT;
      if (j > 0) __syncthreads();                  // barrier_j: block j + 1 landed, block j - 1 released
      // Issue order of a slot (steady state), pinned: 8 A-MFMAs with 2 exponentials + 2 multiplies + 1 convert in every
      // gap, the 8 row-fragment reads of the next A behind them, then the 4 C-MFMAs with the transposed reads between.
#define DQ_PIN_A                                                                                                   \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                               \
  __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                                                               \
  __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
#define DQ_PIN_C                                                                                                   \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                               \
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#ifdef DQ_NOPIN
#define DQ_PIN
#else
#define DQ_PIN                                                                                                     \
  DQ_PIN_A DQ_PIN_A DQ_PIN_A DQ_PIN_A DQ_PIN_A DQ_PIN_A DQ_PIN_A DQ_PIN_A                                          \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                               \
  DQ_PIN_C DQ_PIN_C DQ_PIN_C DQ_PIN_C
#endif
      // slot 0:  A(2) | V(1) | C(0)        then reads: row fragments of 3, transposed of 1
      stageA(sA, dA, std::integral_constant<bool, NEXT == 2>{}, 2);
      stageV(sB, dB, pB);
      load_kv(S0, 3);
      stageC(pA);
      load_kt(S0, 1);
      if constexpr (NEXT == 0) { DQ_PIN }
      // slot 1:  A(3) | V(2) | C(1)        then reads: row fragments of (j + 1, 0), transposed of 2
      stageA(sB, dB, std::integral_constant<bool, NEXT == 2>{}, 3);
      stageV(sA, dA, pA);
      if constexpr (NEXT != 2) load_kv(S1, 0);
      stageC(pB);
      load_kt(S0, 2);
      if constexpr (NEXT == 0) { DQ_PIN }
      // slot 2:  A(j + 1, 0) | V(3) | C(2)  then reads: row fragments of (j + 1, 1), transposed of 3
      if constexpr (NEXT != 2) stageA(sA, dA, std::integral_constant<bool, NEXT == 1>{}, 0);
      stageV(sB, dB, pB);
      if constexpr (NEXT != 2) load_kv(S1, 1);
      stageC(pA);
      load_kt(S0, 3);
      if constexpr (NEXT == 0) { DQ_PIN }
      // slot 3:  A(j + 1, 1) | V(j + 1, 0) | C(3)  then reads: row fragments of (j + 1, 2), transposed of (j + 1, 0)
      if constexpr (NEXT != 2) {
        stageA(sB, dB, std::integral_constant<bool, NEXT == 1>{}, 1);
        stageV(sA, dA, pA);
        load_kv(S1, 2);
      }
      stageC(pB);
      if constexpr (NEXT != 2) load_kt(S1, 0);
      if constexpr (NEXT == 0) { DQ_PIN }
#undef DQ_PIN
#undef DQ_PIN_A
#undef DQ_PIN_C
    };
#pragma unroll 1
    for (int j = 0; j + 2 < nblk; ++j) block(std::integral_constant<int, 0>{}, j);
    if (nblk >= 2) block(std::integral_constant<int, 1>{}, nblk - 2);
    block(std::integral_constant<int, 2>{}, nblk - 1);

    __syncthreads();                               // E1: the ring is free for the next item's first blocks
    {
      bf16* og = (bf16*)a.dq + ((size_t)b * Lq + qrow) * C + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 ov;
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = f2bf(dq[dt][4 * g + k] * 0.125f);
          *(bf16x4*)(og + dt * 32 + 8 * g + 4 * h_) = ov;
        }
    }
    item = item_at(it + 1);
  }
#endif
}
