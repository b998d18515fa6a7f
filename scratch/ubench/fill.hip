// How fast can ONE CU pull bytes, and does it matter which path they take?  Every workgroup (one per CU, 8 waves) streams
// its own region through (a) LDS-DMA (`buffer_load ... lds`, 1 KB per wave instruction, DEPTH instructions in flight per
// wave), (b) ordinary 16-byte loads into registers (the same bytes, DEPTH in flight, results folded into one register),
// (c) half of the waves each.  Region per workgroup: small (stays in L2 / Infinity Cache) or large (HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long u = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((u >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma16(const i32x4& rs, int voff, int soff, unsigned lds) {
  unsigned keep;
  soff = __builtin_amdgcn_readfirstlane(soff);
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// mode 0: all waves DMA; 1: all waves register loads; 2: waves 0-3 DMA, 4-7 register loads
template <int DEPTH>
__global__ __launch_bounds__(512) void k_fill(const unsigned char* base, size_t region, int iters, int mode, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * DEPTH * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* reg = base + (size_t)blockIdx.x * region;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + wave * DEPTH * 1024;
  const bool dma = mode == 0 || (mode == 2 && wave < 4);
  const int nw = 8;                                   // every wave walks its 1/8 of the region, 1 KB per instruction
  const size_t per = region / nw;
  unsigned acc = 0;
  if (dma) {
    const i32x4 rs = make_rsrc(reg + wave * per, (unsigned)per);
    for (int it = 0; it < iters; ++it) {
      for (size_t o = 0; o < per; o += DEPTH * 1024) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) dma16(rs, lane * 16, (int)(o + k * 1024), lds0 + k * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  } else {
    const u32x4* p = (const u32x4*)(reg + wave * per) + lane;
    for (int it = 0; it < iters; ++it) {
      for (size_t o = 0; o < per / 16; o += DEPTH * 64) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = __builtin_nontemporal_load(p + o + k * 64) ;
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int DEPTH>
static void run(const unsigned char* d, size_t region, int mode, unsigned* sink, const char* what) {
  const int nwg = 256;
  const size_t target = (size_t)6 << 30;               // ~6 GB per timed launch
  int iters = (int)(target / (region * nwg)); if (iters < 1) iters = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_fill<DEPTH>, dim3(nwg), dim3(512), 0, 0, d, region, 1, mode, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_fill<DEPTH>, dim3(nwg), dim3(512), 0, 0, d, region, iters, mode, sink);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)region * nwg * iters * (mode == 2 ? 1.0 : 1.0);
  printf("%-28s depth %2d  region/WG %6zu KB: %7.2f TB/s chip, %6.1f GB/s per CU\n", what, DEPTH, region >> 10, bytes / ms / 1e9,
         bytes / ms / 1e6 / nwg);
}
int main() {
  unsigned char* d; hipMalloc(&d, (size_t)2 << 30); hipMemset(d, 1, (size_t)2 << 30);
  unsigned* sink; hipMalloc(&sink, 64);
  const char* names[3] = {"LDS-DMA (8 waves)", "register loads (8 waves)", "4 waves DMA + 4 waves regs"};
  for (size_t region : {(size_t)64 << 10, (size_t)1 << 20, (size_t)8 << 20}) {
    for (int mode = 0; mode < 3; ++mode) {
      run<2>(d, region, mode, sink, names[mode]);
      run<4>(d, region, mode, sink, names[mode]);
      run<8>(d, region, mode, sink, names[mode]);
    }
    printf("\n");
  }
  return 0;
}
