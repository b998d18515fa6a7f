// LDS-DMA helpers shared by the gfx950 kernels that stage tiles with `buffer_load_dwordx4 ... offen lds`.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// [0] = zeros (spatial padding, dgrad frame padding), [1] = ones (forward temporal padding)
static __device__ __attribute__((aligned(64))) const unsigned short oniris_fill_rows[2][32] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80,
     0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80,
     0x3F80, 0x3F80, 0x3F80, 0x3F80}};

typedef __attribute__((ext_vector_type(4))) int i32x4;

// raw buffer resource (stride 0, 32-bit num_records, gfx950 dword-3 flags)
__device__ __forceinline__ i32x4 make_rsrc(const void* p, int bytes) {
  const unsigned long long u = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((u >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000;
  return r;
}

// One LDS-DMA instruction: 16 B per active lane, source = rsrc base + soff + voff (per lane; beyond num_records -> 0),
// destination = LDS byte address `lds` + 16 * lane.  Issued through inline asm ON PURPOSE: hipcc orders every later
// LDS access and every later use of an ordinary load behind a builtin LDS-DMA with `s_waitcnt vmcnt(0)`, which
// serialises the epilogue stores and drains the prefetch.  The kernel waits for its DMA itself (dma_wait() before the
// barrier that publishes a buffer); hipcc's own counted waits stay conservative (it sees fewer VMEM ops than exist).
__device__ __forceinline__ void dma16(const i32x4& rs, int voff, int soff, unsigned lds) {
  unsigned keep;
  soff = __builtin_amdgcn_readfirstlane(soff);          // wave-uniform by construction; make it an SGPR for sure
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

