// Do MFMA 32x32x16 bf16 and VALU (v_exp_f32 / v_fma_f32) work of the SAME SIMD overlap on gfx950?
// One workgroup per CU, W waves per SIMD; each wave runs NIT iterations of: M MFMAs, E exps, F fmas (independent chains).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int M, int E, int F>
__global__ __launch_bounds__(512) void k(float* out, int nit, int wave_role_split) {
  const int wave = threadIdx.x >> 6;
  // role split: waves with (wave & 4) == 0 do the MFMA part only, the others the VALU part only (two waves per SIMD)
  const bool do_m = !wave_role_split || (wave & 4) == 0, do_v = !wave_role_split || (wave & 4) != 0;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float e[8], f[8];
  for (int i = 0; i < 8; ++i) { e[i] = -0.001f * (threadIdx.x + i); f[i] = 0.5f + i; }
  for (int it = 0; it < nit; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int q = 0; q < E; ++q) e[q & 7] = __builtin_amdgcn_exp2f(e[q & 7]);
#pragma unroll
      for (int q = 0; q < F; ++q) f[q & 7] = __builtin_fmaf(f[q & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += e[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M, int E, int F>
static float run(const char* name, int threads, int split, float* out) {
  const int nit = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<M, E, F>), dim3(256), dim3(threads), 0, 0, out, nit, split);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<M, E, F>), dim3(256), dim3(threads), 0, 0, out, nit, split);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s threads=%3d split=%d : %8.1f us  (%.0f ns per iteration)\n", name, threads, split, ms * 1e3, ms * 1e6 / nit);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
  // one wave per SIMD (256 threads = 4 waves)
  run<16, 0, 0>("16 MFMA", 256, 0, out);
  run<0, 32, 0>("32 exp", 256, 0, out);
  run<0, 0, 104>("104 fma", 256, 0, out);
  run<16, 32, 0>("16 MFMA + 32 exp, same wave", 256, 0, out);
  run<16, 0, 104>("16 MFMA + 104 fma, same wave", 256, 0, out);
  run<16, 32, 104>("16 MFMA + 32 exp + 104 fma, same wave", 256, 0, out);
  // two waves per SIMD (512 threads = 8 waves): both do everything
  run<16, 32, 104>("2 waves/SIMD, each MFMA+exp+fma", 512, 0, out);
  // two waves per SIMD with split roles: one only MFMA, the other only VALU
  run<16, 32, 104>("2 waves/SIMD, roles split (MFMA | VALU)", 512, 1, out);
  run<16, 32, 0>("2 waves/SIMD, roles split (MFMA | exp)", 512, 1, out);
  run<16, 0, 104>("2 waves/SIMD, roles split (MFMA | fma)", 512, 1, out);
  return 0;
}
