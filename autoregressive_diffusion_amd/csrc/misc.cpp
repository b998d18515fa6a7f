// Host-only pieces of liboniris_hip.so: error string, ABI version, mask tables, RCCL helpers.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "common.h"
#include "../../include/oniris.h"

static thread_local char g_err[512] = "";

void oniris_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* oniris_last_error(void) { return g_err; }
extern "C" int oniris_abi_version(void) { return 1; }

// ---- mask tables (reference: edm2/attention/attention_masking.py:27-53, 64-90); 128 = flex default block
static const int kFlexBlock = 128;

extern "C" int oniris_train_mask(int n_frames, int image_size, int32_t* num, int32_t* idx, int* block_size) {
  ONIRIS_CHECK_ARG(n_frames > 0 && image_size > 0, "train_mask: bad sizes");
  int nb = n_frames, blk = image_size;
  if (image_size < kFlexBlock) {
    if ((n_frames * image_size) % kFlexBlock != 0) return 0;      // the reference returns None
    nb = n_frames * image_size / kFlexBlock;
    blk = kFlexBlock;
  }
  if (block_size) *block_size = blk;
  const int n2 = 2 * nb;
  if (num)
    for (int i = 0; i < n2; ++i) num[i] = (i % nb) + 1;
  if (idx) {
    memset(idx, 0, sizeof(int32_t) * (size_t)n2 * n2);
    for (int i = 0; i < nb; ++i) {
      for (int j = 0; j <= i; ++j) idx[(size_t)i * n2 + j] = j;                 // clean row: clean blocks 0..i
      for (int j = 0; j < i; ++j) idx[(size_t)(nb + i) * n2 + j] = j;           // noisy row: clean blocks 0..i-1
      idx[(size_t)(nb + i) * n2 + i] = nb + i;                                  //            + its own noisy block
    }
  }
  return n2;
}

extern "C" int oniris_infer_mask(int n_frames, int image_size, int32_t* num, int32_t* idx, int* block_size) {
  ONIRIS_CHECK_ARG(n_frames > 0 && image_size > 0, "infer_mask: bad sizes");
  if (n_frames * image_size < kFlexBlock) return 0;                 // score_mod fall-back
  int nb = n_frames, blk = image_size;
  if (image_size < kFlexBlock) {
    if ((n_frames * image_size) % kFlexBlock != 0) return 0;        // dense create_block_mask fall-back
    nb = n_frames * image_size / kFlexBlock;
    blk = kFlexBlock;
  }
  if (block_size) *block_size = blk;
  if (num)
    for (int i = 0; i < nb; ++i) num[i] = i + 1;
  if (idx) {
    memset(idx, 0, sizeof(int32_t) * (size_t)nb * nb);
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j <= i; ++j) idx[(size_t)i * nb + j] = j;
  }
  return nb;
}

extern "C" int oniris_mask_transpose(int n_rows, int n_cols, const int32_t* kv_num, const int32_t* kv_idx,
                                     int32_t* q_num, int32_t* q_idx) {
  ONIRIS_CHECK_ARG(n_rows > 0 && n_cols > 0 && kv_num && kv_idx && q_num && q_idx, "mask_transpose: bad arguments");
  for (int c = 0; c < n_cols; ++c) q_num[c] = 0;
  memset(q_idx, 0, sizeof(int32_t) * (size_t)n_cols * n_rows);
  for (int r = 0; r < n_rows; ++r)
    for (int j = 0; j < kv_num[r]; ++j) {
      const int c = kv_idx[(size_t)r * n_cols + j];
      ONIRIS_CHECK_ARG(c >= 0 && c < n_cols, "mask_transpose: index out of range");
      q_idx[(size_t)c * n_rows + q_num[c]++] = r;
    }
  return ONIRIS_OK;
}

// ---- RCCL helpers
extern "C" int oniris_comm_unique_id(void* id128) {
  ONIRIS_CHECK_ARG(id128, "comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) { oniris_set_error("ncclGetUniqueId failed"); return ONIRIS_ELAUNCH; }
  memcpy(id128, &id, 128);
  return ONIRIS_OK;
}
extern "C" int oniris_comm_init(void** comm, int rank, int world, const void* id128) {
  ONIRIS_CHECK_ARG(comm && id128 && world > 0 && rank >= 0 && rank < world, "comm_init: bad arguments");
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t c;
  ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) { oniris_set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); return ONIRIS_ELAUNCH; }
  *comm = (void*)c;
  return ONIRIS_OK;
}
extern "C" int oniris_comm_allreduce_sum_f32(void* comm, float* buf, size_t count, oniris_stream_t stream) {
  ONIRIS_CHECK_ARG(comm && buf, "comm_allreduce: null");
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  if (r != ncclSuccess) { oniris_set_error("ncclAllReduce: %s", ncclGetErrorString(r)); return ONIRIS_ELAUNCH; }
  return ONIRIS_OK;
}
extern "C" int oniris_comm_destroy(void* comm) {
  if (comm) ncclCommDestroy((ncclComm_t)comm);
  return ONIRIS_OK;
}

// sizes of the argument structs, so a binding (ctypes, cgo, JNI ...) can verify its mirror of include/oniris.h
extern "C" int oniris_struct_sizes(int32_t* out4) {
  ONIRIS_CHECK_ARG(out4, "struct_sizes: null");
  out4[0] = (int32_t)sizeof(OnirisWeightDesc); out4[1] = (int32_t)sizeof(OnirisConvArgs);
  out4[2] = (int32_t)sizeof(OnirisWgradArgs);  out4[3] = (int32_t)sizeof(OnirisAttnArgs);
  return ONIRIS_OK;
}
