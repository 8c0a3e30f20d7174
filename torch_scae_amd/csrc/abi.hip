// ABI bookkeeping of libscae_hip.so (see include/scae_hip.h).
#include "common.h"

#include <string.h>

#include <new>

extern "C" int scae_abi_version(void) { return SCAE_ABI_VERSION; }

extern "C" const char *scae_error_string(int code) {
  if (code == SCAE_OK) return "ok";
  if (code == SCAE_ERR_BAD_ARG) return "scae: null pointer or non-positive size";
  if (code == SCAE_ERR_UNSUPPORTED) return "scae: shape outside this build's kernel limits";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "scae: unknown error";
}

// ---- launch lists (common.h: scae::launch) ------------------------------------------------
// The recordings that are open, each bound to ONE stream; a launch is appended to the
// recording of its own stream only.  This table is the bookkeeping of the handles the callers
// hold (like an allocator's), not state an entry point's result depends on.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

namespace {
struct Launch {
  const void *fn;
  dim3 grid, block;
  size_t lds;
  std::vector<unsigned long long> blob;   // the arguments, each at a 16-byte boundary
  std::vector<size_t> at;                 // byte offsets into blob
  int lane;                               // 0: the recording's stream, 1: its side stream
  int after;                              // >= 0: a SYNC record instead of a launch -- lane
                                          // `lane` waits for what lane `after` has been given
};
struct List {
  hipStream_t stream[2];   // [1]: the side lane's stream (null: none)
  bool open;
  std::vector<Launch> launches;
  int n_kernels;
  std::vector<hipEvent_t> events;   // one per sync record, made by the first two-stream run
};
std::atomic<int> g_open{0};     // (scae::launch's fast path: nothing records)
std::mutex g_mu;                // (forward and backward launches come from different host threads)
std::vector<List *> g_lists;    // the open recordings
thread_local int t_launch_err = 0;

int lane_of(const List *l, hipStream_t st) {
  if (st == l->stream[0]) return 0;
  return l->stream[1] && st == l->stream[1] ? 1 : -1;
}
}  // namespace

namespace scae_rec {
bool recording() { return g_open.load(std::memory_order_relaxed) > 0; }
void append(const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, void *const *args,
            const size_t *sizes, int n) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (List *list : g_lists) {
    const int lane = lane_of(list, st);
    if (lane < 0) continue;
    Launch l{fn, grid, block, lds, {}, {}, lane, -1};
    size_t bytes = 0;
    for (int i = 0; i < n; ++i) {
      l.at.push_back(bytes);
      bytes += (sizes[i] + 15) & ~(size_t)15;
    }
    l.blob.assign((bytes + 7) / 8 + 2, 0ull);
    // (the vector's storage is 16-byte aligned by the allocator for these sizes)
    for (int i = 0; i < n; ++i)
      memcpy(reinterpret_cast<unsigned char *>(l.blob.data()) + l.at[i], args[i], sizes[i]);
    list->launches.push_back(std::move(l));
    ++list->n_kernels;
  }
}
// the hipError_t of the calling thread's first failed hipLaunchKernel (scae::launch), once
void note_launch_error(int e) {
  if (!t_launch_err) t_launch_err = e;   // (the first failure of a launcher that issues several)
}
int take_launch_error() {
  const int e = t_launch_err;
  t_launch_err = 0;
  return e;
}
}  // namespace scae_rec

static void close_list(List *l) {   // g_mu held
  if (!l->open) return;
  l->open = false;
  g_lists.erase(std::remove(g_lists.begin(), g_lists.end(), l), g_lists.end());
  g_open.fetch_sub(1);
}

extern "C" void *scae_launch_list_begin(void *stream) {
  List *l = new (std::nothrow) List{{(hipStream_t)stream, nullptr}, true, {}, 0, {}};
  if (!l) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  g_lists.push_back(l);
  g_open.fetch_add(1);
  return l;
}
extern "C" int scae_launch_list_side_stream(void *list, void *side_stream) {
  SCAE_REQUIRE(list && side_stream);
  std::lock_guard<std::mutex> lock(g_mu);
  List *l = static_cast<List *>(list);
  if (!l->open || l->stream[1] || (hipStream_t)side_stream == l->stream[0]) return SCAE_ERR_BAD_ARG;
  l->stream[1] = (hipStream_t)side_stream;
  return SCAE_OK;
}
extern "C" int scae_launch_list_order(void *later_stream, void *earlier_stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  int n = 0;
  for (List *l : g_lists) {
    const int a = lane_of(l, (hipStream_t)later_stream), b = lane_of(l, (hipStream_t)earlier_stream);
    if (a < 0 || b < 0 || a == b) continue;
    ++n;
    // (the same edge again with nothing given to the earlier lane since: already implied)
    bool implied = false;
    for (auto it = l->launches.rbegin(); it != l->launches.rend(); ++it) {
      if (it->after >= 0) {
        if (it->lane == a && it->after == b) implied = true;
        if (implied) break;
        continue;
      }
      if (it->lane == b) break;
    }
    if (!implied) l->launches.push_back(Launch{nullptr, dim3(), dim3(), 0, {}, {}, a, b});
  }
  return n;
}
extern "C" int scae_launch_list_end(void *list) {
  SCAE_REQUIRE(list);
  std::lock_guard<std::mutex> lock(g_mu);
  List *l = static_cast<List *>(list);
  if (!l->open) return SCAE_ERR_BAD_ARG;
  close_list(l);
  return SCAE_OK;
}
extern "C" int scae_launch_list_size(const void *list) {
  if (!list) return 0;
  std::lock_guard<std::mutex> lock(g_mu);
  return static_cast<const List *>(list)->n_kernels;
}
extern "C" int scae_launch_list_side_size(const void *list) {
  if (!list) return 0;
  std::lock_guard<std::mutex> lock(g_mu);
  int n = 0;
  for (const Launch &l : static_cast<const List *>(list)->launches) n += l.after < 0 && l.lane == 1;
  return n;
}
extern "C" int scae_launch_list_run2(const void *list, void *stream, void *side_stream) {
  SCAE_REQUIRE(list);
  List *ls = const_cast<List *>(static_cast<const List *>(list));
  if (ls->open) return SCAE_ERR_BAD_ARG;   // (still recording: it would record itself)
  hipStream_t st[2] = {(hipStream_t)stream, side_stream ? (hipStream_t)side_stream : (hipStream_t)stream};
  const bool two = st[0] != st[1];
  size_t ev = 0;
  for (const Launch &l : ls->launches) {
    if (l.after >= 0) {   // lane l.lane waits for what lane l.after has been given so far
      if (!two) continue;   // (one stream: its own order already says so)
      if (ev == ls->events.size()) {
        hipEvent_t e;
        hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (rc != hipSuccess) return (int)rc;
        ls->events.push_back(e);
      }
      hipError_t rc = hipEventRecord(ls->events[ev], st[l.after]);
      if (rc == hipSuccess) rc = hipStreamWaitEvent(st[l.lane], ls->events[ev], 0);
      if (rc != hipSuccess) return (int)rc;
      ++ev;
      continue;
    }
    void *ptrs[64];
    if (l.at.size() > 64) return SCAE_ERR_UNSUPPORTED;
    unsigned char *base =
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(l.blob.data()));
    for (size_t i = 0; i < l.at.size(); ++i) ptrs[i] = base + l.at[i];
    hipError_t e = hipLaunchKernel(l.fn, l.grid, l.block, ptrs, l.lds, st[l.lane]);
    if (e != hipSuccess) return (int)e;
  }
  return SCAE_OK;
}
// One run of the list with a timing event in front of and behind every launch (on the launch's
// own stream): out[2 i], out[2 i + 1] = start / end of launch i in microseconds after the
// first event -- the step's timeline WITH its overlap (a kernel trace serialises the dispatches).
// Synchronises both streams; the event pairs add a few microseconds per launch.
extern "C" int scae_launch_list_timeline(const void *list, void *stream, void *side_stream,
                                         float *out_us, int n_out) {
  SCAE_REQUIRE(list && out_us);
  List *ls = const_cast<List *>(static_cast<const List *>(list));
  if (ls->open || n_out < 2 * ls->n_kernels) return SCAE_ERR_BAD_ARG;
  hipStream_t st[2] = {(hipStream_t)stream, side_stream ? (hipStream_t)side_stream : (hipStream_t)stream};
  const bool two = st[0] != st[1];
  std::vector<hipEvent_t> ev(2 * ls->n_kernels), sy;
  for (auto &e : ev)
    if (hipEventCreate(&e) != hipSuccess) return SCAE_ERR_UNSUPPORTED;
  int k = 0;
  hipError_t rc = hipSuccess;
  for (const Launch &l : ls->launches) {
    if (l.after >= 0) {
      if (!two) continue;
      hipEvent_t e;
      rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (rc != hipSuccess) break;
      sy.push_back(e);
      rc = hipEventRecord(e, st[l.after]);
      if (rc == hipSuccess) rc = hipStreamWaitEvent(st[l.lane], e, 0);
      if (rc != hipSuccess) break;
      continue;
    }
    void *ptrs[64];
    unsigned char *base =
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(l.blob.data()));
    for (size_t i = 0; i < l.at.size() && i < 64; ++i) ptrs[i] = base + l.at[i];
    rc = hipEventRecord(ev[2 * k], st[l.lane]);
    if (rc == hipSuccess) rc = hipLaunchKernel(l.fn, l.grid, l.block, ptrs, l.lds, st[l.lane]);
    if (rc == hipSuccess) rc = hipEventRecord(ev[2 * k + 1], st[l.lane]);
    if (rc != hipSuccess) break;
    ++k;
  }
  if (rc == hipSuccess) rc = hipStreamSynchronize(st[0]);
  if (rc == hipSuccess && two) rc = hipStreamSynchronize(st[1]);
  for (int i = 0; rc == hipSuccess && i < 2 * k; ++i) {
    float ms = 0.f;
    rc = hipEventElapsedTime(&ms, ev[0], ev[i]);
    out_us[i] = ms * 1e3f;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  for (auto &e : sy) (void)hipEventDestroy(e);
  return rc == hipSuccess ? SCAE_OK : (int)rc;
}
extern "C" int scae_launch_list_lane(const void *list, int i) {   // lane of launch i, or -1
  if (!list) return -1;
  std::lock_guard<std::mutex> lock(g_mu);
  int k = 0;
  for (const Launch &l : static_cast<const List *>(list)->launches) {
    if (l.after >= 0) continue;
    if (k++ == i) return l.lane;
  }
  return -1;
}
extern "C" int scae_launch_list_run(const void *list, void *stream) {
  return scae_launch_list_run2(list, stream, nullptr);
}
extern "C" void scae_launch_list_free(void *list) {
  if (!list) return;
  List *l = static_cast<List *>(list);
  {
    std::lock_guard<std::mutex> lock(g_mu);
    close_list(l);
  }
  for (hipEvent_t e : l->events) (void)hipEventDestroy(e);
  delete l;
}
