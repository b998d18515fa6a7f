// What can an elementwise pass with act_bwd's traffic mix reach on gfx950?  3 tensors read + 1 written, 16 bytes per lane, 268 MB
// tensors (the 64x64 level of the gym net at B = 8: 1024 frames x 4096 pixels x 32 channels bf16).  Variants:
//   G  = 16-byte groups per thread (independent loads in flight per lane: 3 * G)
//   NT = non-temporal loads / stores (streamed once: no point in keeping the lines in L2 / the Infinity Cache)
//   MATH = the silu-derivative arithmetic of act_bwd (8 x exp + rcp per group) or a bare add
// and a plain copy (1 read + 1 write) as the yardstick.  hipcc --offload-arch=gfx950 -O3 ew_mix.hip -o ew_mix
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
typedef unsigned short bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ inline unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

template <int G, bool NT, bool MATH>
__global__ __launch_bounds__(256) void k_mix(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, const bf16x8* __restrict__ c,
                                             bf16x8* __restrict__ o, size_t n) {
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  bf16x8 va[G], vb[G], vc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t i = base + g * stride;
    if (i < n) {
      if (NT) { va[g] = __builtin_nontemporal_load(a + i); vb[g] = __builtin_nontemporal_load(b + i); vc[g] = __builtin_nontemporal_load(c + i); }
      else { va[g] = a[i]; vb[g] = b[i]; vc[g] = c[i]; }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t i = base + g * stride;
    if (i < n) {
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float x = bf2f(vc[g][k]);
        float v;
        if (MATH) {
          const float s = 1.f / (1.f + __expf(-x));
          v = bf2f(va[g][k]) * (s * (1.f + x * (1.f - s))) * 1.6778523f + bf2f(vb[g][k]);
        } else {
          v = bf2f(va[g][k]) + bf2f(vb[g][k]) + x;
        }
        r[k] = f2bf(v);
      }
      if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
    }
  }
}

// the same with act_bwd's index arithmetic: gid -> (pixel, channel group) by a 64-bit division by the RUN-TIME group count
template <bool NT, int DIVMODE>
__global__ __launch_bounds__(256) void k_mix_div(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, const bf16x8* __restrict__ c,
                                                 bf16x8* __restrict__ o, long long npix, int G, unsigned long long magic) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  long long pix; int cg;
  if (DIVMODE == 0) { pix = gid / G; cg = (int)(gid % G); }
  else { pix = (long long)(((unsigned long long)gid * magic) >> 40); cg = (int)(gid - pix * G); }     // magic = ceil(2^40 / G)
  if (pix >= npix) return;
  const size_t i = (size_t)pix * G + cg;
  bf16x8 va, vb, vc;
  if (NT) { va = __builtin_nontemporal_load(a + i); vb = __builtin_nontemporal_load(b + i); vc = __builtin_nontemporal_load(c + i); }
  else { va = a[i]; vb = b[i]; vc = c[i]; }
  bf16x8 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float x = bf2f(vc[k]);
    const float s = 1.f / (1.f + __expf(-x));
    r[k] = f2bf(bf2f(va[k]) * (s * (1.f + x * (1.f - s))) * 1.6778523f + bf2f(vb[k]));
  }
  if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const bf16x8* __restrict__ a, bf16x8* __restrict__ o, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), o + i); else o[i] = a[i]; }
}

template <typename F>
static float timeit(F f, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  const size_t elems = (size_t)1024 * 4096 * 32, n = elems / 8, bytes = elems * 2;
  bf16x8 *a, *b, *c, *o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes);
  hipMemset(a, 0x3c, bytes); hipMemset(b, 0x3c, bytes); hipMemset(c, 0x3c, bytes);
  const int rep = 30;
#define RUN(G, NT, MATH) { const unsigned grid = (unsigned)((n + 256 * G - 1) / (256 * G)); \
    const float us = timeit([&] { hipLaunchKernelGGL((k_mix<G, NT, MATH>), dim3(grid), dim3(256), 0, 0, a, b, c, o, n); }, rep); \
    printf("3R+1W  G=%d %s %s: %7.1f us  %.2f TB/s\n", G, NT ? "nt   " : "plain", MATH ? "silu'" : "add  ", us, 4.0 * bytes / us * 1e-6); }
  RUN(1, false, false) RUN(1, false, true) RUN(2, false, true) RUN(4, false, true)
  RUN(1, true, false) RUN(1, true, true) RUN(2, true, true) RUN(4, true, true)
  for (int G : {4, 12}) {
    const long long npix = (long long)(n / G);
    const unsigned grid = (unsigned)((n + 255) / 256);
    const unsigned long long magic = ((1ull << 40) + G - 1) / G;
    float us = timeit([&] { hipLaunchKernelGGL((k_mix_div<true, 0>), dim3(grid), dim3(256), 0, 0, a, b, c, o, npix, G, magic); }, rep);
    printf("3R+1W nt silu' + 64-bit div/mod by G=%2d: %7.1f us  %.2f TB/s\n", G, us, 4.0 * bytes / us * 1e-6);
    us = timeit([&] { hipLaunchKernelGGL((k_mix_div<true, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, o, npix, G, magic); }, rep);
    printf("3R+1W nt silu' + multiply-shift   G=%2d: %7.1f us  %.2f TB/s\n", G, us, 4.0 * bytes / us * 1e-6);
  }
  { const unsigned grid = (unsigned)((n + 255) / 256);
    float us = timeit([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, o, n); }, rep);
    printf("copy 1R+1W plain: %7.1f us  %.2f TB/s\n", us, 2.0 * bytes / us * 1e-6);
    us = timeit([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, o, n); }, rep);
    printf("copy 1R+1W nt:    %7.1f us  %.2f TB/s\n", us, 2.0 * bytes / us * 1e-6); }
  return 0;
}
