#include <cstdlib>
// Host-only pieces of liboniris_hip.so: error string, ABI version, mask tables, attention schedule.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <cxxabi.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
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

thread_local hipEvent_t oniris_prof_ev[2] = {nullptr, nullptr};
extern "C" int oniris_profile_arm(void* start_event, void* stop_event) {
  ONIRIS_CHECK_ARG(start_event && stop_event, "profile_arm: null event");
  oniris_prof_ev[0] = (hipEvent_t)start_event;
  oniris_prof_ev[1] = (hipEvent_t)stop_event;
  return ONIRIS_OK;
}
extern "C" int oniris_profile_disarm(void) {          // 1: the pair was still armed (no launch consumed it), 0: consumed
  const int armed = oniris_prof_ev[0] != nullptr;
  oniris_prof_ev[0] = oniris_prof_ev[1] = nullptr;
  return armed;
}
extern "C" int oniris_abi_version(void) { return 14; }

// Size from which a tensor is streamed with non-temporal accesses.  ONIRIS_EW_NT_MB (default 96; <= 0: never) seeds it;
// oniris_set_ew_nt_bytes replaces it at run time (the tests force every NT instantiation onto small oracle-sized tensors).
static long long g_ew_nt_bytes = -1;
long long oniris_ew_nt_bytes(void) {
  if (g_ew_nt_bytes < 0) {
    const char* e = getenv("ONIRIS_EW_NT_MB");
    const long long mb = e ? atoll(e) : 96;
    g_ew_nt_bytes = mb <= 0 ? (1LL << 62) : mb * (1LL << 20);          // 0 / negative: never
  }
  return g_ew_nt_bytes;
}
extern "C" long long oniris_set_ew_nt_bytes(long long bytes) {      // returns the previous threshold; bytes < 0: "never"
  const long long old = oniris_ew_nt_bytes();
  g_ew_nt_bytes = bytes < 0 ? (1LL << 62) : bytes;
  return old;
}

// ---- dispatch census (common.h): (kernel handle, tag) -> launches
int oniris_census_on = 0;
static std::mutex g_census_mu;
static std::map<std::pair<const void*, std::string>, long long> g_census;
void oniris_census_note(const void* h, const char* tag) {
  std::lock_guard<std::mutex> lock(g_census_mu);
  ++g_census[std::make_pair(h, std::string(tag ? tag : ""))];
}
extern "C" int oniris_census(int on) {                 // 1: clear the list and start noting, 0: stop (the list stays readable)
  std::lock_guard<std::mutex> lock(g_census_mu);
  if (on) g_census.clear();
  oniris_census_on = on ? 1 : 0;
  return ONIRIS_OK;
}
static std::string census_name(const void* h) {
  const char* raw = hipKernelNameRefByPtr(h, nullptr);
  Dl_info info;
  if ((!raw || !*raw) && dladdr(h, &info) && info.dli_sname) raw = info.dli_sname;
  (void)hipGetLastError();
  if (!raw || !*raw) {
    char buf[32];
    snprintf(buf, sizeof(buf), "kernel@%p", h);
    return buf;
  }
  int status = 0;
  char* dem = abi::__cxa_demangle(raw, nullptr, nullptr, &status);
  std::string name = (status == 0 && dem) ? dem : raw;
  free(dem);
  const size_t paren = name.rfind('(');                 // "void k<..>(Args)" -> "k<..>"
  if (paren != std::string::npos && name.back() == ')') name.resize(paren);
  if (name.rfind("void ", 0) == 0) name.erase(0, 5);
  const std::string stub = "__device_stub__";
  const size_t at = name.find(stub);
  if (at != std::string::npos) name.erase(at, stub.size());
  return name;
}
extern "C" long long oniris_census_read(char* buf, long long cap) {   // "<launches>\t<kernel>[ [tag]]\n" per entry; returns the bytes needed
  std::map<std::string, long long> out;
  {
    std::lock_guard<std::mutex> lock(g_census_mu);
    for (const auto& kv : g_census) {
      std::string key = census_name(kv.first.first);
      if (!kv.first.second.empty()) key += " [" + kv.first.second + "]";
      out[key] += kv.second;
    }
  }
  std::string text;
  for (const auto& kv : out) text += std::to_string(kv.second) + "\t" + kv.first + "\n";
  const long long need = (long long)text.size() + 1;
  if (buf && cap > 0) {
    const long long n = need <= cap ? need - 1 : cap - 1;
    memcpy(buf, text.data(), (size_t)n);
    buf[n] = 0;
  }
  return need;
}

int oniris_cu_reserve = 0;
int oniris_persistent_wgs(void) {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
      ncu = 256;
  }
  const int n = ncu - oniris_cu_reserve;
  return n < 8 ? 8 : n;
}
extern "C" int oniris_set_cu_reserve(int k) {          // returns the previous value; k is rounded up to whole XCD octets
  ONIRIS_CHECK_ARG(k >= 0 && k <= 128, "set_cu_reserve: 0 <= k <= 128");
  const int old = oniris_cu_reserve;
  oniris_cu_reserve = (k + 7) / 8 * 8;
  return old;
}

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

// ---- static balanced schedule of attention work items over persistent workgroups (see include/oniris.h)
extern "C" int oniris_attn_schedule(int n_pairs, int n_blocks, const int32_t* weight, int n_wg, int32_t* sched,
                                    int n_slots) {
  ONIRIS_CHECK_ARG(n_pairs > 0 && n_blocks > 0 && n_blocks < 65536 && n_pairs < 32768 && weight && n_wg > 0,
                   "attn_schedule: bad arguments");
  int ng = 1;
  for (int g = 8; g > 1; g >>= 1)
    if (n_pairs % g == 0 && n_wg % g == 0) { ng = g; break; }
  std::vector<int> order(n_blocks);
  for (int i = 0; i < n_blocks; ++i) order[i] = i;
  // heaviest block first (stable: equal weights keep block order)
  for (int i = 1; i < n_blocks; ++i) {
    const int v = order[i];
    int j = i - 1;
    while (j >= 0 && weight[order[j]] < weight[v]) { order[j + 1] = order[j]; --j; }
    order[j + 1] = v;
  }
  std::vector<long long> load(n_wg, 0);
  std::vector<std::vector<int32_t>> lists(n_wg);
  for (int g = 0; g < ng; ++g) {
    // the group's items: every block of every pair of the group, heaviest first (pairs interleaved)
    for (int bi = 0; bi < n_blocks; ++bi) {
      const int blk = order[bi];
      for (int p = g; p < n_pairs; p += ng) {
        int best = -1;
        for (int w = g; w < n_wg; w += ng)
          if (best < 0 || load[w] < load[best]) best = w;
        load[best] += weight[blk];
        lists[best].push_back((int32_t)((p << 16) | blk));
      }
    }
  }
  int need = 1;
  for (int w = 0; w < n_wg; ++w) need = std::max(need, (int)lists[w].size());
  if (!sched) return need;
  ONIRIS_CHECK_ARG(n_slots >= need, "attn_schedule: %d slots given, %d needed", n_slots, need);
  for (int w = 0; w < n_wg; ++w)
    for (int k = 0; k < n_slots; ++k) sched[(size_t)w * n_slots + k] = k < (int)lists[w].size() ? lists[w][k] : -1;
  return need;
}

// sizes of the argument structs, so a binding (ctypes, cgo, JNI ...) can verify its mirror of include/oniris.h
extern "C" int oniris_struct_sizes(int32_t* out4) {
  ONIRIS_CHECK_ARG(out4, "struct_sizes: null");
  out4[0] = (int32_t)sizeof(OnirisWeightDesc); out4[1] = (int32_t)sizeof(OnirisConvArgs);
  out4[2] = (int32_t)sizeof(OnirisWgradArgs);  out4[3] = (int32_t)sizeof(OnirisAttnArgs);
  return ONIRIS_OK;
}
