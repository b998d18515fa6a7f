// Which bf16 MFMA shape delivers more FLOP/s on RANDOM data under the chip's power management: 32x32x16 (2x2 register
// blocking) or 16x16x32 (4x4 blocking)?  Same LDS bytes per FLOP (every operand fragment is re-read from LDS, 16 B per
// lane), same accumulator count (64 registers), 8 waves per workgroup, one workgroup per CU, long enough to settle the clock.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip ; run: ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const unsigned short* src, float* out, int nit) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 1024];      // 64 KB of operand data
  for (int i = threadIdx.x; i < 32 * 1024 / 8; i += 512) *(uint4*)(lds + i * 8) = *(const uint4*)(src + (size_t)blockIdx.x * 32 * 1024 % (1 << 20) + i * 8);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* base = lds + wave * 2048 + lane * 8;                  // conflict-free: lane-linear 16-byte reads
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int st = 0; st < 8; ++st) {                                         // 8 k-steps of 16: 4 fragment reads, 4 MFMAs each
        bf16x8 A[2], B[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { A[a] = *(const bf16x8*)(base + ((st * 4 + a) * 512 + it * 64) % 14336); B[a] = *(const bf16x8*)(base + ((st * 4 + 2 + a) * 512 + it * 64) % 14336); }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], B[b], acc[a][b], 0, 0, 0);
      }
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
  } else {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.f;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {                                         // 4 k-steps of 32: 8 fragment reads, 16 MFMAs each
        bf16x8 A[4], B[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { A[a] = *(const bf16x8*)(base + ((st * 8 + a) * 512 + it * 64) % 14336); B[a] = *(const bf16x8*)(base + ((st * 8 + 4 + a) * 512 + it * 64) % 14336); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[a], B[b], acc[a][b], 0, 0, 0);
      }
    }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) s += acc[a][b][i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE>
static void run(const char* name, const unsigned short* src, float* out) {
  const int nit = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(512), 0, 0, src, out, nit);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(512), 0, 0, src, out, nit);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 20.0 * 256 * 8 * (double)nit * 32 * 32768.0;            // per wave-iteration: 32 x 32 KFLOP either way
  printf("%-34s %8.2f ms  %7.0f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned short* src; float* out;
  hipMalloc(&src, (1 << 20) * 2 + 65536 * 2); hipMalloc(&out, 256 * 512 * sizeof(float));
  unsigned short* h = (unsigned short*)malloc((1 << 20) * 2 + 65536 * 2);
  for (int z = 0; z < 2; ++z) {
    srand(1);
    for (int i = 0; i < (1 << 20) + 65536; ++i) {                              // random bf16 in about [-2, 2] / zeros
      const float f = z ? 0.f : ((rand() & 0xffff) / 16384.f - 2.f);
      unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(src, h, (1 << 20) * 2 + 65536 * 2, hipMemcpyHostToDevice);
    printf("%s operands\n", z ? "all-zero" : "random");
    run<32>("  32x32x16, 2x2 blocking", src, out);
    run<16>("  16x16x32, 4x4 blocking", src, out);
    run<32>("  32x32x16, 2x2 blocking (again)", src, out);
    run<16>("  16x16x32, 4x4 blocking (again)", src, out);
  }
  return 0;
}
