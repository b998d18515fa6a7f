// What does a kernel boundary cost on gfx950?  N dependent launches in one stream of kernels that do (almost) nothing:
// time per launch = dispatch + ramp + drain + end-of-kernel release.  Variants: threads per workgroup, static LDS, a kernel
// that dirties L2 (writes MB bytes) before it ends.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSB>
__global__ __launch_bounds__(512) void k_empty(float* out, int n_write) {
  __shared__ unsigned char s[LDSB > 0 ? LDSB : 4];
  if (LDSB > 0 && threadIdx.x == 0) s[0] = 1;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < n_write; ++k) out[i + (size_t)k * gridDim.x * blockDim.x] = (float)k;
}
template <typename K>
static float run(K kern, dim3 g, dim3 b, float* d, int nw, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, g, b, 0, 0, d, nw);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, g, b, 0, 0, d, nw);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}
int main() {
  float* d; hipMalloc(&d, (size_t)1 << 30);
  const int n = 2000;
  printf("1 WG x 64 threads, no LDS:                 %.2f us per launch\n", run(k_empty<0>, dim3(1), dim3(64), d, 0, n));
  printf("256 WG x 512 threads, no LDS:              %.2f us per launch\n", run(k_empty<0>, dim3(256), dim3(512), d, 0, n));
  printf("256 WG x 512 threads, 64 KB LDS:           %.2f us per launch\n", run(k_empty<65536>, dim3(256), dim3(512), d, 0, n));
  printf("2048 WG x 256 threads, no LDS:             %.2f us per launch\n", run(k_empty<0>, dim3(2048), dim3(256), d, 0, n));
  for (int mb : {1, 8, 32, 128}) {
    const int nw = mb * (1 << 20) / 4 / (256 * 512);
    printf("256 WG x 512 threads writing %3d MB:        %.2f us per launch (%.2f us at 5 TB/s)\n", mb, run(k_empty<0>, dim3(256), dim3(512), d, nw, 500), mb / 5.0);
  }
  return 0;
}
