// VideoAttention forward for gfx950: persistent, statically balanced, wave-specialised, three-stage software pipeline
// (included by attention.hip; the training-table / causal-prefill path of oniris_attn_fwd).
//
// Launch: one 512-thread workgroup per CU; the host hands every workgroup a list of 128-row query blocks whose
// key-block counts add up to the same total (oniris_attn_schedule; a (batch, head) pair stays on one XCD's L2).
//
// Waves 4..7 = LOADERS (one per SIMD).  They only issue LDS-DMA (buffer_load ... lds, ~100 cycles of issue each:
// that time runs beside the compute wave of the same SIMD) and wait for it: K | V blocks of 128 keys (32 KB) go
// through a four-slot ring three blocks ahead of the consumers; the next item's Q tile and first three blocks are
// requested while the compute waves finish the current item, so an item starts without a pipeline fill.
//
// Waves 0..3 = COMPUTE, wave (qw, st) = query rows [64 qw, 64 qw + 64) (two 32-row MFMA blocks x = 0, 1) x key half
// st of every 128-key block.  S^T = K.Q^T has the key on the MFMA rows and the query on the lanes, so a lane owns a
// query row: its row sum is plain adds and the S^T accumulator, exponentiated and packed, is the B operand of
// O^T += V^T.P^T.  The unit of work is a MICRO-STEP t = (block, 32-key half kt, x): 4 S^T MFMAs, 16 exponentials +
// 16 adds + 8 converts per lane, 4 P.V MFMAs.  Pipeline slot t issues  QK(t+2) | softmax(t+1) | PV(t):  the eight
// MFMAs of a slot depend only on results of EARLIER slots, so the matrix pipe runs back to back while the VALU does
// the exponentials of the micro-step in between (at head dim 64 a micro-step is 256 MFMA cycles against ~290 cycles
// of VALU issue: the loop is built to keep both busy, not to hide one behind the other).  K / V fragments are read
// from LDS two slots before their first use (two register sets each).
//
// q arrives with log2(e)/8 folded in (qkv_norm_kernel); no running maximum and no offset: q, k are unit vectors times
// 8 (qkv normalisation), |score| <= 11.6 in the log2 domain, so exp2 stays far inside fp32 / bf16 range.
// Only the LAST block of a list is partially masked (diagonal / own noisy block): its mask enters as the initial
// accumulator of the S^T MFMAs (0 / -1e30), the softmax code is the same everywhere.
// LDS: ring 4 x 32 KB + two Q tiles 2 x 16 KB = 160 KB.  Barriers per item: one per block + two in the epilogue,
// executed by all eight waves.
#pragma once

#ifdef ATTN_STAMP
#define WS_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WS_STAMP(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; }
#define WS_SETTLE asm volatile("" ::"v"(o[0][0]), "v"(o[0][1]), "v"(o[1][0]), "v"(o[1][1]), "v"(sA), "v"(sB), "v"(osum));
#else
#define WS_STAMP_DECL
#define WS_STAMP(i)
#define WS_SETTLE
#endif

template <int MODE>
__global__ __launch_bounds__(512, 2) void attn_fwd_ws_kernel(const AttnDev d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int SLOT = 2 * 128 * 128;              // K [128 keys][128 B] | V [128 keys][128 B]
  constexpr int QOFF = 4 * SLOT, QTILE = 128 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * SLOT + 2 * QTILE];
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
    const int ksw = (dpp ^ ((drow >> 1) & 7)) * 16, vsw_ = (dpp ^ (4 * ((drow >> 1) & 1))) * 16;
    constexpr int OOB = (int)0x80000000;
    // everything of item `itm` that block / Q requests need (wave-uniform values + the table row in a VGPR)
    struct Src { i32x4 rs_k, rs_v, rs_q; int kvo, vvo, qvo, qrow0, nblk, kvl; };
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
      s.rs_q = make_rsrc((const bf16*)a.q + (size_t)b * Lq * C, Lq * C * 2);
      s.kvo = (drow * C + head * 64) * 2 + ksw;
      s.vvo = (drow * C + head * 64) * 2 + vsw_;
      s.qvo = s.kvo;
      s.qrow0 = qb128 * 128;
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
    auto issue_q = [&](const Src& s, int par) __attribute__((always_inline)) {
      const unsigned dst = lds0 + QOFF + par * QTILE + lw * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = s.qrow0 + drow + 32 * i < Lq;
        dma16(s.rs_q, ok ? s.qvo : OOB, (s.qrow0 + 32 * i) * C * 2, dst + i * 4096);
      }
    };
    int item = item_at(0);
    if (item < 0) return;
    Src cur = open_item(item);
    issue_q(cur, 0);
#pragma unroll 1
    for (int j = 0; j < 3 && j < cur.nblk; ++j) issue_block(cur, j);
#pragma unroll 1
    for (int it = 0;; ++it) {
      const int nblk = cur.nblk;
      // requests so far, in order: Q tile, blocks 0 .. min(nblk, 3) - 1.  barrier_j needs blocks <= j + 1 landed.
#pragma unroll 1
      for (int j = 0; j < nblk; ++j) {
        if (j + 2 < nblk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // block j + 2 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // barrier_j: block j + 1 landed; block j - 1 released
        if (j + 3 < nblk) issue_block(cur, j + 3);
      }
      const int next = item_at(it + 1);
      Src nxt = cur;
      if (next >= 0) {
        nxt = open_item(next);                     // (nothing of this item is in flight any more)
        issue_q(nxt, (it + 1) & 1);
      }
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
  const int r = lane & 31, h = lane >> 5;
  const int qw = wave & 1, st = wave >> 1;
  const int kb0 = (64 * st + r) * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  const int grp = lane >> 4, hh = grp >> 1, q4 = (lane & 15) >> 2, pcol = (lane & 3) * 4 + 16 * (grp & 1);
  const int vb0 = 128 * 128 + (64 * st + 4 * hh + q4) * 128 + pcol * 2, vsw = (q4 >> 1) & 1;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto vtr = [&](const unsigned char* slot, int tokbase, int dt) __attribute__((always_inline)) {
    const unsigned char* p0 = slot + vb0 + tokbase * 128 + ((dt ^ vsw) * 64);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 128));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
  // A operand of the row-sum MFMA (see pv below): a 16x16x32 MFMA reads the P fragment of a 32x32x16 B operand (lane =
  // (query n, key half kb)) as k-block 2 kb + n / 16 of column n % 16; rows 4 g .. 4 g + 3 of A select the k-blocks with
  // parity g & 1, so that D row-group g of column n' holds sum_k P[k][n' + 16 (g & 1)] -- and D row-group g is what lanes
  // 16 g .. 16 g + 15 receive: every lane gets the row sum of its own query lane & 31.
  bf16x8 sel16;
#pragma unroll
  for (int e = 0; e < 8; ++e) sel16[e] = f2bf((((lane >> 4) ^ ((lane & 15) >> 2)) & 1) ? 0.f : 1.f);
  typedef __attribute__((ext_vector_type(4))) float f32x4_;

  int item = item_at(0);
  WS_STAMP_DECL
#pragma unroll 1
  for (int it = 0; item >= 0; ++it) {
    const int pair = item >> 16, qb128 = item & 0xffff;
    const int b = pair / a.heads, head = pair - b * a.heads;
    const int qw0 = qb128 * 128 + qw * 64;          // lane r of query block x: row qw0 + 32 x + r
    // per-item copy of the lane id for everything OUTSIDE the block loop (Q fragment addresses, mask, epilogue): hipcc
    // otherwise hoists ~40 lane-constant address registers out of the item loop and spills them to scratch around the
    // block loop -- every reload is a scratch load + vmcnt(0) in the item's prologue / epilogue (stamped: ~8 K cycles)
    int le = lane;
    asm volatile("" : "+v"(le));
    const int r_ = le & 31, h_ = le >> 5;
    const int trow = qb128 >> d.tshift;
    const int nblk = __builtin_amdgcn_readfirstlane(a.kv_num[trow]) << d.tshift;
    const int last_e = __builtin_amdgcn_readfirstlane(a.kv_idx[(size_t)trow * a.tab_cols + ((nblk - 1) >> d.tshift)]);
    const int last_key0 = ((last_e << d.tshift) + ((nblk - 1) & tmask)) * 128 + 64 * st;

    f32x16 o[2][2];
    f32x4_ osum[2];                                 // row sums of P of query block x (all four registers hold the lane's row)
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][0][i] = 0.f; o[0][1][i] = 0.f; o[1][0][i] = 0.f; o[1][1][i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { osum[0][i] = 0.f; osum[1][i] = 0.f; }

    auto load_k = [&](bf16x8 (&kf)[4], const unsigned char* S0, int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(S0 + ((kb0 ^ (ks * 32)) + kt * 4096));
    };
    auto load_v = [&](bf16x8 (&vf)[2][2], const unsigned char* S0, int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[s2][dt] = vtr(S0, kt * 32 + 16 * s2, dt);
    };
    // mask of the list's last block as start values of S^T (x, kt): 0 (allowed) / -1e30.  An accumulator register group
    // rr = 4 g .. 4 g + 3 holds 4 consecutive, 4-aligned keys: one frame (P is a power of two >= 4) and one side of Lk (a
    // multiple of 8), so the mask is evaluated once per group.
    auto bias = [&](int x, int kt) __attribute__((always_inline)) {
      f32x16 m;
      const int qrow = qw0 + 32 * x + r_;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key = last_key0 + 32 * kt + 8 * g + 4 * h_;
        const float v = (key < Lk && qrow < Lq && tok_allowed<MODE>(qrow, key, d.pshift, a.T, d.qf_off)) ? 0.f : NEG_BIG;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[4 * g + k] = v;
      }
      return m;
    };
    // the three stages of a pipeline slot
    auto qk = [&](f32x16& s, bf16x8 (&kf)[4], bf16x8 (&q)[4], auto masked_, int x, int kt) __attribute__((always_inline)) {
      if constexpr (decltype(masked_)::value) {
        s = bias(x, kt);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = mfma32(kf[ks], q[ks], s);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = mfma32(kf[ks], q[ks], ks == 0 ? zero16 : s);
      }
    };
    auto sm = [&](f32x16& s, bf16x8 (&pb)[2], int x) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[s2][e] = f2bf(__builtin_amdgcn_exp2f(s[8 * s2 + e]));
    };
    // P.V of a micro-step + its row sums ON THE MATRIX PIPE (a 4-pass 16x16x32 MFMA against the selector `sel16`): the 16
    // adds per lane and micro-step they replace (hipcc pairs them into v_pk_add_f32, ~4x the issue cost of a plain add
    // beside MFMAs) were a third of the VALU issue time of the loop, and the matrix pipe has the room (32 full + 8 half
    // MFMAs per block).
    auto pv = [&](bf16x8 (&vf)[2][2], bf16x8 (&pb)[2], int x) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[x][dt] = mfma32(vf[s2][dt], pb[s2], o[x][dt]);
        osum[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel16, pb[s2], osum[x], 0, 0, 0);
      }
    };

    __syncthreads();                               // barrier_0: Q tile, blocks 0 and 1 landed
    WS_STAMP(0)
    bf16x8 qf[2][4];
    {
      const unsigned char* Qt = smem + QOFF + (it & 1) * QTILE;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int row = 64 * qw + 32 * x + r_;
          qf[x][ks] = *(const bf16x8*)(Qt + row * 128 + (((2 * ks + h_) ^ ((row >> 1) & 7)) << 4));   // (q carries log2(e)/8)
        }
    }
    // register sets: kA = K fragments of half kt = 0, kB of kt = 1; vA = V^T fragments of kt = 0, vB of kt = 1;
    // sA / sB = S^T of the even / odd micro-step in flight; pA / pB = packed P of the even / odd micro-step
    bf16x8 kA[4], kB[4], vA[2][2], vB[2][2], pA[2], pB[2];
    f32x16 sA, sB;
    load_k(kA, smem, 0);
    load_k(kB, smem, 1);
    load_v(vA, smem, 0);
    // pipeline fill: S^T of micro-steps 0 = (0, 0, x = 0) and 1 = (0, 0, x = 1), softmax of micro-step 0
    if (nblk == 1) {
      qk(sA, kA, qf[0], std::true_type{}, 0, 0);
      qk(sB, kA, qf[1], std::true_type{}, 1, 0);
    } else {
      qk(sA, kA, qf[0], std::false_type{}, 0, 0);
      qk(sB, kA, qf[1], std::false_type{}, 1, 0);
    }
    sm(sA, pA, 0);
    WS_SETTLE
    WS_STAMP(1)

    // One block = four slots.  Slot u of block j:  QK of micro-step (u + 2)  |  softmax of (u + 1)  |  PV of u, with
    //   micro-step u = (kt = u >> 1, x = u & 1) of block j, u = 4, 5 = (0, 0), (0, 1) of block j + 1.
    // NEXT: 0 = block j + 1 exists and is not the last, 1 = block j + 1 is the last (its S^T starts from the mask),
    //       2 = block j is the last (its kt = 1 S^T starts from the mask; nothing follows).
    auto block = [&](auto next_, int j) __attribute__((always_inline)) {
      constexpr int NEXT = decltype(next_)::value;
      const unsigned char* S0 = smem + (j & 3) * SLOT;
      const unsigned char* S1 = smem + ((j + 1) & 3) * SLOT;
      if (j > 0) __syncthreads();                  // barrier_j: block j + 1 landed, block j - 1 released
      // Issue order of a slot, pinned with sched_group_barrier: an in-order wave keeps the matrix pipe busy only while
      // the VALU work between two MFMAs stays under ~24 cycles of issue, so the slot's 16 exponentials and 8 converts
      // are dealt out two / one at a time between its 10 MFMAs (hipcc's own order bunches 5-8 exponentials, during
      // which the pipe drains), and the 12 fragment reads of a half block ride in the first gaps.
#define WS_PIN_GAP(READS, NEXP, CVT)                                                   \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
  if (READS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                        \
  __builtin_amdgcn_sched_group_barrier(0x400, NEXP, 0);                                \
  if (CVT) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
#define WS_PIN(READS)                                                                  \
  WS_PIN_GAP(READS, 2, 1) WS_PIN_GAP(READS, 2, 1) WS_PIN_GAP(READS, 2, 1) WS_PIN_GAP(READS, 2, 1) WS_PIN_GAP(READS, 2, 1) \
  WS_PIN_GAP(READS, 2, 1) WS_PIN_GAP(0, 1, 1) WS_PIN_GAP(0, 1, 1) WS_PIN_GAP(0, 1, 0) WS_PIN_GAP(0, 1, 0)
      // slot 0:  QK(1, 0) | softmax(0, 1) | PV(0, 0)          reads: vB <- V(j, kt 1), kA <- K(j + 1, kt 0)
      load_v(vB, S0, 1);
      if constexpr (NEXT != 2) load_k(kA, S1, 0);
      qk(sA, kB, qf[0], std::integral_constant<bool, NEXT == 2>{}, 0, 1);
      sm(sB, pB, 1);
      pv(vA, pA, 0);
      if constexpr (NEXT == 0) { WS_PIN(1) }
      // slot 1:  QK(1, 1) | softmax(1, 0) | PV(0, 1)
      qk(sB, kB, qf[1], std::integral_constant<bool, NEXT == 2>{}, 1, 1);
      sm(sA, pA, 0);
      pv(vA, pB, 1);
      if constexpr (NEXT == 0) { WS_PIN(0) }
      // slot 2:  QK(next 0, 0) | softmax(1, 1) | PV(1, 0)     reads: vA <- V(j + 1, kt 0), kB <- K(j + 1, kt 1)
      if constexpr (NEXT != 2) {
        load_v(vA, S1, 0);
        load_k(kB, S1, 1);
        qk(sA, kA, qf[0], std::integral_constant<bool, NEXT == 1>{}, 0, 0);
      }
      sm(sB, pB, 1);
      pv(vB, pA, 0);
      if constexpr (NEXT == 0) { WS_PIN(1) }
      // slot 3:  QK(next 0, 1) | softmax(next 0, 0) | PV(1, 1)
      if constexpr (NEXT != 2) {
        qk(sB, kA, qf[1], std::integral_constant<bool, NEXT == 1>{}, 1, 0);
        sm(sA, pA, 0);
      }
      pv(vB, pB, 1);
      if constexpr (NEXT == 0) { WS_PIN(0) }
    };
    WS_STAMP(2)
#pragma unroll 1
    for (int j = 0; j + 2 < nblk; ++j) block(std::integral_constant<int, 0>{}, j);
    WS_SETTLE
    WS_STAMP(3)
    if (nblk >= 2) block(std::integral_constant<int, 1>{}, nblk - 2);
    block(std::integral_constant<int, 2>{}, nblk - 1);
    WS_SETTLE
    WS_STAMP(4)

    // ---- epilogue: wave (qw, st) finishes query block x = st; the other one's partial (O, l) goes to its SIMD-pair
    // partner (qw, 1 - st) through ring slot 3 (8 KB per wave) / the item's Q tile (row sums)
    const float l[2] = {osum[0][0], osum[1][0]};
    float* red = (float*)(smem + 3 * SLOT) + wave * 2048 + le;        // this wave's 32 x 64 floats
    float* lred = (float*)(smem + QOFF + (it & 1) * QTILE) + le;      // [4 waves][64]
    __syncthreads();                               // E1: every compute wave is done with the ring and the Q tile
    WS_STAMP(5)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      red[i * 64] = st ? o[0][0][i] : o[1][0][i];
      red[(16 + i) * 64] = st ? o[0][1][i] : o[1][1][i];
    }
    lred[wave * 64] = st ? l[0] : l[1];
    __syncthreads();                               // E2
    {
      const int pw = wave ^ 2;                     // partner: same query rows, other key half
      const float* pr = (const float*)(smem + 3 * SLOT) + pw * 2048 + le;
      f32x16 oa = st ? o[1][0] : o[0][0], ob = st ? o[1][1] : o[0][1];
#pragma unroll
      for (int i = 0; i < 16; ++i) { oa[i] += pr[i * 64]; ob[i] += pr[(16 + i) * 64]; }
      const float lx = (st ? l[1] : l[0]) + lred[pw * 64];            // (the MFMA sums run over all 16 keys of a k-step)
      const float inv = (lx > 0.f) ? __builtin_amdgcn_rcpf(lx) : 0.f;
      const int q0 = qw0 + 32 * st;
      // wave-private transpose through the partner's region (only this wave reads it, and it has just done so):
      // O^T (lane = query row) -> [row][64 ch] bf16 rows of 128 B, 16-byte parts XOR-swizzled with the row
      unsigned char* ot = smem + 3 * SLOT + pw * 8192;
      unsigned char* otw = ot + r_ * 128 + 8 * h_;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 ov;
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = f2bf((dt ? ob : oa)[4 * g + k] * inv);
          const int part = dt * 4 + g;
          *(bf16x4*)(otw + ((part ^ (r_ & 7)) << 4)) = ov;
        }
      if (a.lse && h_ == 0 && q0 + r_ < Lq)
        a.lse[(size_t)(b * a.heads + head) * Lq + q0 + r_] = __builtin_amdgcn_logf(fmaxf(lx, 1e-30f));   // v_log_f32 = log2
      const int row0 = le >> 3, part = le & 7;
      bf16* og = (bf16*)a.out + ((size_t)b * Lq + q0 + row0) * C + head * 64 + part * 8;
      const unsigned char* otr = ot + row0 * 128 + ((part ^ row0) << 4);       // rows row0 + 8 i: (row & 7) == row0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 v = *(const u32x4*)(otr + i * 1024);
        if (q0 + row0 + 8 * i < Lq) *(u32x4*)(og + (size_t)(8 * i) * C) = v;
      }
    }
    item = item_at(it + 1);
    WS_STAMP(6)
  }
#ifdef ATTN_STAMP
  if (blockIdx.x == 0 && lane == 0 && a.dkv_part) {
    unsigned long long* dst = (unsigned long long*)a.dkv_part + wave * 8;
    for (int i = 0; i < 8; ++i) dst[i] = st_acc[i];
  }
#endif
#endif
}
