// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long cycles, int* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
  int* d; hipMalloc(&d, 64);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned long long cyc = 10000000ULL;     // 100 MHz counter -> 100 ms?  (s_memtime runs at 100 MHz)
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, s);
      for (int k = 0; k < 4; ++k)
        hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, cyc / 100, d);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("flags=%d rep=%d: 4 kernels of 32 WGs: %.3f ms  (err %s)\n", flags, rep, ms, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
