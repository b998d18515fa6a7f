// Shared device/host helpers for liboniris_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ONIRIS_OK 0
#define ONIRIS_EINVAL (-1)
#define ONIRIS_ELAUNCH (-2)
#define ONIRIS_EUNSUPPORTED (-3)

// thread-local last error (C-ABI: oniris_last_error)
void oniris_set_error(const char* fmt, ...);

#define ONIRIS_CHECK_ARG(cond, ...)                \
  do {                                             \
    if (!(cond)) {                                 \
      oniris_set_error(__VA_ARGS__);               \
      return ONIRIS_EINVAL;                        \
    }                                              \
  } while (0)

#define ONIRIS_LAUNCH_CHECK()                                            \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      oniris_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,    \
                       hipGetErrorString(e_));                           \
      return ONIRIS_ELAUNCH;                                             \
    }                                                                    \
  } while (0)

// Measurement aid (oniris_profile_arm, misc.cpp): when a start / stop event pair is armed, the next kernel this thread
// launches through oniris_launch records its OWN begin and end into them (hipExtLaunchKernel: the timestamps of the
// dispatch itself, what rocprofv3 reports) -- bracketing events on the stream also time the ~2 us of kernel boundary.
extern thread_local hipEvent_t oniris_prof_ev[2];

// Dispatch census (oniris_census_*, misc.cpp; tests/conftest.py + tests/test_zz_dispatch_coverage.py): while it is on, every
// kernel launch of the library notes WHICH instantiation went out (the kernel handle, resolved to its demangled name when the
// list is read) plus a tag for variants that are chosen at run time inside one instantiation (non-temporal output stores).
// The test suite uses it to prove that what the timed regions of bench.py launch is what the oracle-comparing tests launched.
extern int oniris_census_on;
void oniris_census_note(const void* kernel_handle, const char* tag);
#define ONIRIS_KLAUNCH(kern, grid, block, shmem, stream, ...)                            \
  do {                                                                                   \
    if (oniris_census_on) oniris_census_note((const void*)(kern), nullptr);              \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                   \
  } while (0)

template <typename K, typename... A>
static inline void oniris_launch_tagged(const char* tag, K kern, dim3 grid, dim3 block, hipStream_t stream, A... args) {
  if (oniris_census_on) oniris_census_note((const void*)kern, tag);
  if (oniris_prof_ev[0]) {
    hipExtLaunchKernelGGL(kern, grid, block, 0, stream, oniris_prof_ev[0], oniris_prof_ev[1], 0, args...);
    oniris_prof_ev[0] = oniris_prof_ev[1] = nullptr;
  } else {
    hipLaunchKernelGGL(kern, grid, block, 0, stream, args...);
  }
}
template <typename K, typename... A>
static inline void oniris_launch(K kern, dim3 grid, dim3 block, hipStream_t stream, A... args) {
  oniris_launch_tagged(nullptr, kern, grid, block, stream, args...);
}

// Streaming loads / stores of the HBM-bound passes (elementwise.hip, gconv_bwd_fused): tensors far larger than the caches are
// read and written exactly once per pass -- the non-temporal policy (no retention in L2 / the Infinity Cache) measured +10 %
// on act_bwd's 3-reads-1-write mix at 268 MB per tensor (scratch/ubench/ew_mix.hip: 5.84 -> 6.45 TB/s).  NT is chosen on the
// host per launch: tensors of at least oniris_ew_nt_bytes() bytes (ONIRIS_EW_NT_MB, default 96; smaller ones may still be in
// the 256 MB Infinity Cache when their consumer runs).
long long oniris_ew_nt_bytes(void);
template <bool NT, typename V>
__device__ __forceinline__ V ldv(const V* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename V>
__device__ __forceinline__ void stv(V* p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Persistent launches (one workgroup per CU: conv_glds.h, conv1x1_glds.h; the attention work lists are sized by the caller):
// the CU count of the current device minus oniris_set_cu_reserve()'s k -- CUs left to the collective library's kernels while
// a gradient exchange is in flight (a persistent workgroup that finds its CU's LDS taken waits for a whole other workgroup
// to finish: the launch takes up to twice as long).  misc.cpp.
extern int oniris_cu_reserve;
int oniris_persistent_wgs(void);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int roundup(int a, int b) { return cdiv(a, b) * b; }

#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// MFMA 32x32x16 bf16:  D[i][j] += sum_k A[i][k] * B[k][j]
//   lane l: A[i = l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][j = l&31], e = 0..7
//   D: lane holds col j = l&31, rows i = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// sigmoid through the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp, far below the bf16 rounding of every
// consumer) instead of __expf + an IEEE division (~10 more VALU ops per element in the HBM-bound elementwise passes)
__device__ __forceinline__ float sigmoid_fast(float z) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
#endif
