// VideoAttention backward, dK / dV for gfx950: persistent, statically balanced, wave-specialised
// (included by attention.hip; the scheduled path of oniris_attn_bwd_dkv).
//
// The grid kernel (attn_bwd_dkv_kernel) gives a workgroup 128 keys and -- because under the causal DART table the first key
// blocks are attended by every later query block and the last by one -- splits every key block's query list over
// `dkv_chunks` workgroups that write fp32 partial sums (134 MB per layer at the gym shape) for a second kernel to add.
// Here a work ITEM is 64 keys of one (batch, head) pair with its WHOLE query list: the heaviest item (64 clean keys of
// frame 0: 63 query blocks of 128 rows at T = 64, P = 64) is below the per-CU average of the launch (66 block units),
// so a longest-first assignment of whole items to one workgroup per CU (oniris_attn_schedule, the forward's scheduler)
// balances the launch and dK / dV leave the kernel finished, in bf16: no partial sums, no reduction kernel.
//
// Launch: one 512-thread workgroup per CU.  Waves 4..7 = LOADERS: they only issue LDS-DMA -- 128-row blocks of Q | dO
// (16 KB each) + the rows' lse | delta (1 KB) through a four-slot ring, three blocks ahead of the consumers, and the first
// three blocks of the next item while the compute waves finish the current one.  Waves 0..3 = COMPUTE, wave (kh, qh) =
// keys [32 kh, 32 kh + 32) of the item (lane = key; K and V fragments stay in registers for the whole item) x query half
// qh of every 128-row block (two 32-row sub-blocks).  Per sub-block: S' = Q.K^T - lse and dP' = dO.V^T - delta (8 MFMAs,
// query on the rows, key on the lanes; -lse and -delta, written by oniris_attn_bwd_prep, are the chains' initial
// accumulators), P = exp2(S'), dS = P dP' in registers (2 VALU instructions + 2 converts per element), then
// dV^T += dO^T.P and dK^T += Q^T.dS (8 MFMAs) with the transposed operands read from the SAME LDS tiles
// (ds_read_b64_tr_b16).  The two query halves of a key half add their dK / dV through LDS at the end of the item.
// q arrives with log2(e)/8 folded in (qkv_norm_rope_kernel): dK is rescaled by 1/log2(e) in the epilogue.
// OnirisAttnArgs.lse / .delta of THIS kernel point at the NEGATED rows (oniris_attn_bwd_prep's `neg` output).
#pragma once

// KEYS = 128 (round 5): a work item is a whole 128-token table block.  Wave w of the compute waves owns keys [32 w, 32 w + 32) of
// it against ALL four 32-row sub-blocks of every 128-row query block: 64 MFMAs per wave and block instead of 32, i.e. half the
// LDS-DMA instructions (the loaders' issue time, which bounds the 64-key form: profiles/r05_pmc_attn_bwd_sq.txt) and half the
// ring bytes per MFMA, and nothing to merge at the end of an item.  The heaviest item doubles, so the host picks this form only
// when a workgroup's average load is well above it (ops._attn_core_bwd: B = 8 at the bench shape, not B = 2).
template <int MODE, int KEYS = 64>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_ws_kernel(const AttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TQ = 128 * 128;                    // one tile: 128 rows x 128 B (64 channels of a head)
  constexpr int SLOT = 2 * TQ + 1024;              // Q | dO | lse[128] | delta[128]
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
    // ONE image for the row reads (ds_read_b128) and the transposing reads (ds_read_b64_tr_b16) of a tile: the 16-byte
    // chunk c of row R sits at chunk c ^ f(R), f(R) = bit1(R) << 2 | bit3(R) << 1 | bit2(R) -- both kinds of read are
    // conflict-free (bank model of MI355X_MICROARCH.md, LDS section; the transposing-read swizzle alone left the row reads
    // 4-way conflicted, and with four compute waves on one LDS that was the kernel's bound)
    const int tsw_ = (dpp ^ ((((drow >> 1) & 1) << 2) | (((drow >> 3) & 1) << 1) | ((drow >> 2) & 1))) * 16;
    constexpr int OOB = (int)0x80000000;
    struct Src { i32x4 rs_q, rs_do, rs_l, rs_d; int qvo, nblk, qvl; };
    auto open_item = [&](int itm) __attribute__((always_inline)) {
      Src s;
      const int pair = itm >> 16, kb64 = itm & 0xffff;                // (KEYS = 128: the index of the 128-key block)
      const int b = pair / a.heads, head = pair - b * a.heads;
      const int trow = (KEYS == 128 ? kb64 : (kb64 >> 1)) >> d.tshift;
      const int nent = __builtin_amdgcn_readfirstlane(a.q_num[trow]);
      s.nblk = nent << d.tshift;
      s.qvl = (lane < nent) ? a.q_idx[(size_t)trow * a.qtab_cols + lane] : 0;
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(s.qvl)::"memory");     // (hipcc does not count the asm DMAs: wait by hand)
      s.rs_q = make_rsrc((const bf16*)a.q + (size_t)b * Lq * C, Lq * C * 2);
      s.rs_do = make_rsrc((const bf16*)a.dout + (size_t)b * Lq * C, Lq * C * 2);
      s.rs_l = make_rsrc(a.lse + (size_t)(b * a.heads + head) * Lq, Lq * 4);
      s.rs_d = make_rsrc(a.delta + (size_t)(b * a.heads + head) * Lq, Lq * 4);
      s.qvo = (drow * C + head * 64) * 2 + tsw_;
      return s;
    };
    auto issue_block = [&](const Src& s, int j) __attribute__((always_inline)) {       // block j of the list -> slot j % 4
      const int q0 = (((__builtin_amdgcn_readlane(s.qvl, j >> d.tshift)) << d.tshift) + (j & tmask)) * 128;
      const unsigned dst = lds0 + (j & 3) * SLOT + lw * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = drow + 32 * i < Lq - q0;
        const int so = (q0 + 32 * i) * C * 2;
        dma16(s.rs_q, ok ? s.qvo : OOB, so, dst + i * 4096);
        dma16(s.rs_do, ok ? s.qvo : OOB, so, dst + i * 4096 + TQ);
      }
      if (lw == 0) {                               // lanes 0..31: lse[q0 .. q0+127] (16 B each), lanes 32..63: delta
        const int l32 = lane & 31;
        const int vo = (l32 * 4 < Lq - q0) ? l32 * 16 : OOB;
        const unsigned sdst = lds0 + (j & 3) * SLOT + 2 * TQ;
        if (lane < 32) dma16(s.rs_l, vo, q0 * 4, sdst);
        else dma16(s.rs_d, vo, q0 * 4, sdst);
      }
    };
    int item = item_at(0);
    if (item < 0) return;
    Src cur = open_item(item);
#pragma unroll 1
    for (int j = 0; j < 3 && j < cur.nblk; ++j) issue_block(cur, j);
#pragma unroll 1
    for (int it = 0;; ++it) {
      const int nblk = cur.nblk, nb = nblk > 0 ? nblk : 1;
      // requests so far, in order: blocks 0 .. min(nblk, 3) - 1.  barrier_j needs blocks <= j + 1 landed; a block is 8 DMA
      // instructions of a loader wave, 10 of wave 4 (lse / delta)
#pragma unroll 1
      for (int j = 0; j < nb; ++j) {
        if (j + 2 < nblk) {
          if (lw == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");             // block j + 2 may still be in flight
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                           // barrier_j: block j + 1 landed; block j - 1 released
        if (j + 3 < nblk) issue_block(cur, j + 3);
      }
      const int next = item_at(it + 1);
      Src nxt = cur;
      if (next >= 0) nxt = open_item(next);        // (nothing of this item is in flight any more)
      __syncthreads();                             // E1: the compute waves are done with the ring
      if (next >= 0) {
#pragma unroll 1
        for (int j = 0; j < 3 && j < nxt.nblk; ++j) issue_block(nxt, j);      // slots 0..2; the merge uses slot 3
      }
      __syncthreads();                             // E2
      if (next < 0) break;
      cur = nxt;
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- compute waves
  const int kh = (KEYS == 128) ? wave : (wave & 1), qh = (KEYS == 128) ? 0 : (wave >> 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;

  int item = item_at(0);
#pragma unroll 1
  for (int it = 0; item >= 0; ++it) {
    const int pair = item >> 16, kb64 = item & 0xffff;
    const int b = pair / a.heads, head = pair - b * a.heads;
    // per-item copy of the lane id: keeps hipcc from hoisting (and spilling) lane-constant addresses out of the item loop
    int le = lane;
    asm volatile("" : "+v"(le));
    const int r = le & 31, h = le >> 5;
    const int kw0 = kb64 * KEYS + kh * 32;
    const int krow = kw0 + r;
    const int rr0 = r * 128 + ((h ^ ((((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1))) << 4);
    // transposing reads: lane (grp, q4, p) takes row tokbase + 4 hh + q4 (+ 8 for the second read), logical bytes
    // 8 p + 32 (grp & 1) + 64 dt of it: chunk c = 2 (grp & 1) + (p >> 1) + 4 dt, stored at c ^ f(row)
    const int grp = le >> 4, hh = grp >> 1, q4 = (le & 15) >> 2, c0 = 2 * (grp & 1) + ((le & 3) >> 1);
    const int tbA = (4 * hh + q4) * 128 + ((c0 ^ hh) << 4) + 8 * (le & 1);
    const int tbB = (4 * hh + q4 + 8) * 128 + ((c0 ^ hh ^ 2) << 4) + 8 * (le & 1);
    const int tsw = (q4 >> 1) & 1;
    auto ttr = [&](const unsigned char* t_, int tokbase, int dt) __attribute__((always_inline)) {
      const int o = tokbase * 128 + ((dt ^ tsw) * 64);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t_ + tbA + o));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t_ + tbB + o));
      s16x8 v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      return __builtin_bit_cast(bf16x8, v);
    };

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
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

    const int trow = (KEYS == 128 ? kb64 : (kb64 >> 1)) >> d.tshift;
    const int nent = __builtin_amdgcn_readfirstlane(a.q_num[trow]);
    const int nblk = nent << d.tshift, nb = nblk > 0 ? nblk : 1;
    const int qvl = (lane < nent) ? a.q_idx[(size_t)trow * a.qtab_cols + lane] : 0;
    auto q_start = [&](int j) __attribute__((always_inline)) {
      return (((__builtin_amdgcn_readlane(qvl, j >> d.tshift)) << d.tshift) + (j & tmask)) * 128;
    };

    // The stream of work is the list's 32-row sub-blocks u = (block j, qs): rows [64 qh + 32 qs, +32) of block j.
    //   A(u): S' = Q.K^T - lse and dP' = dO.V^T - delta: the row constants (negated by oniris_attn_bwd_prep) are the INITIAL
    //         accumulators of the two MFMA chains; a partially masked sub-block starts S' at -1e30 where the pair is masked
    //         (P = exp2(S') = 0 there, and so is dS): the softmax code below is the same for every sub-block;
    //   B(u): P = exp2(S'), dS = P dP' (the 1/8 of the score scale is applied once, to dK, in the epilogue), packed to bf16;
    //   C(u): dV^T += dO^T.P, dK^T += Q^T.dS with the transposed fragments of the same tiles.
    // MASKED: the sub-block may be partially masked.  Under the DART training table and the causal prefill table that is
    // the FIRST block of a key block's query list only (the diagonal block: clean queries of the keys' own frames, or -- for
    // noisy keys -- the only entry); every later block is fully allowed.  The hot loop therefore carries no mask code and
    // no branch (a branch splits hipcc's scheduling region: the MFMAs of A and the VALU work of B must share one).
    auto stageA = [&](auto masked_, f32x16& s, f32x16& dp, const unsigned char* Qt, int trw, int q0) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_)::value;
      const unsigned char* dOt = Qt + TQ;
      const float* nl = (const float*)(Qt + 2 * TQ) + trw + 4 * h;
      const float* nd = nl + 128;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 ls = *(const float4*)(nl + 8 * g4), de = *(const float4*)(nd + 8 * g4);
        s[4 * g4] = ls.x; s[4 * g4 + 1] = ls.y; s[4 * g4 + 2] = ls.z; s[4 * g4 + 3] = ls.w;
        dp[4 * g4] = de.x; dp[4 * g4 + 1] = de.y; dp[4 * g4 + 2] = de.z; dp[4 * g4 + 3] = de.w;
      }
      if constexpr (MASKED) {
        const int qq0 = q0 + trw;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int qtok = qq0 + 8 * (rr >> 2) + 4 * h + (rr & 3);
          s[rr] = tok_allowed<MODE>(qtok, krow, d.pshift, a.T, d.qf_off) ? s[rr] : NEG_BIG;
        }
      }
      bf16x8 qa[4], da[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qa[ks] = *(const bf16x8*)(Qt + ((rr0 ^ (ks * 32)) + trw * 128));
        da[ks] = *(const bf16x8*)(dOt + ((rr0 ^ (ks * 32)) + trw * 128));
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(qa[ks], kf[ks], s);                   // S'[q][key]
        dp = mfma32(da[ks], vf[ks], dp);                 // dP'[q][key]
      }
    };
    // B and C of a sub-block, interleaved by halves: the exponentials / products / converts of key-step half s2 = 1 ride in
    // the gaps of the four MFMAs of half s2 = 0 (one set of S' / dP' accumulators: a second set, for a deeper pipeline
    // across sub-blocks, does not fit the 256 registers next to dK, dV, K and V: it spilled)
    auto stageBC = [&](const f32x16& s, const f32x16& dp, const unsigned char* Qt, int trw) __attribute__((always_inline)) {
      const unsigned char* dOt = Qt + TQ;
      bf16x8 pb[2], db[2], dotf[2][2], qtf[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { dotf[s2][dt] = ttr(dOt, trw + 16 * s2, dt); qtf[s2][dt] = ttr(Qt, trw + 16 * s2, dt); }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_amdgcn_exp2f(s[8 * s2 + e]);             // (q carries log2(e)/8: qkv_norm_rope_kernel)
          pb[s2][e] = f2bf(p);
          db[s2][e] = f2bf(p * dp[8 * s2 + e]);
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = mfma32(dotf[s2][dt], pb[s2], dv[dt]);     // dV^T[dv][key] += dO^T[dv][q] P[q][key]
          dk[dt] = mfma32(qtf[s2][dt], db[s2], dk[dt]);      // dK^T[d][key]  += Q^T[d][q] dS[q][key]
        }
      // pinned order: the 16 transposing reads, the 8 + 8 + 8 VALU instructions of half 0, then each MFMA of half 0 with
      // two exponentials, two products and two converts of half 1 behind it, then the MFMAs of half 1
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
#pragma unroll
      for (int g_ = 0; g_ < 4; ++g_) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    };

    const int trw0 = qh * 64, trw1 = qh * 64 + 32;
    const bool work = kw0 < Lk && nblk > 0;
    f32x16 sA, dpA;
    auto block = [&](auto first_, int j) __attribute__((always_inline)) {
      const unsigned char* S0 = smem + (j & 3) * SLOT;
      const int q0 = q_start(j);
      if constexpr (KEYS == 128) {
#pragma unroll
        for (int trw = 0; trw < 128; trw += 32) {
          stageA(first_, sA, dpA, S0, trw, q0);
          stageBC(sA, dpA, S0, trw);
        }
      } else {
        stageA(first_, sA, dpA, S0, trw0, q0);
        stageBC(sA, dpA, S0, trw0);
        stageA(first_, sA, dpA, S0, trw1, q0);
        stageBC(sA, dpA, S0, trw1);
      }
    };
    __syncthreads();                               // barrier_0: blocks 0 and 1 landed
    if (work) block(std::true_type{}, 0);
#pragma unroll 1
    for (int j = 1; j < nb; ++j) {
      __syncthreads();                             // barrier_j: block j + 1 landed, block j - 1 released
      if (work) block(std::false_type{}, j);
    }

    // ---- epilogue: the two query halves of a key half meet in ring slot 3 (2 x 16 KB), wave (kh, 0) writes dK, dV
    float* red = (float*)(smem + 3 * SLOT) + (kh & 1) * 4096 + le;       // [kh][64 registers][64 lanes]
    __syncthreads();                               // E1: every compute wave is done with the ring
    if (KEYS == 64 && qh == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        red[i * 64] = dk[0][i]; red[(16 + i) * 64] = dk[1][i];
        red[(32 + i) * 64] = dv[0][i]; red[(48 + i) * 64] = dv[1][i];
      }
    }
    __syncthreads();                               // E2
    if (qh == 0 && krow < Lk) {
      if constexpr (KEYS == 64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dk[0][i] += red[i * 64]; dk[1][i] += red[(16 + i) * 64];
          dv[0][i] += red[(32 + i) * 64]; dv[1][i] += red[(48 + i) * 64];
        }
      }
      bf16* dkg = (bf16*)a.dk + ((size_t)b * Lk + krow) * C + head * 64;
      bf16* dvg = (bf16*)a.dv + ((size_t)b * Lk + krow) * C + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o1, o2;
#pragma unroll
          for (int k = 0; k < 4; ++k) { o1[k] = f2bf(dk[dt][4 * g + k] * (0.125f / SCALE_LOG2)); o2[k] = f2bf(dv[dt][4 * g + k]); }   // dK^T was summed against q' = c q and the unscaled P dP'; 1/8 = the score scale
          *(bf16x4*)(dkg + dt * 32 + 8 * g + 4 * h) = o1;
          *(bf16x4*)(dvg + dt * 32 + 8 * g + 4 * h) = o2;
        }
    }
    item = item_at(it + 1);
  }
#endif
}
