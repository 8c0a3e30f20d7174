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
};
struct List {
  hipStream_t stream;
  bool open;
  std::vector<Launch> launches;
};
std::atomic<int> g_open{0};     // (scae::launch's fast path: nothing records)
std::mutex g_mu;                // (forward and backward launches come from different host threads)
std::vector<List *> g_lists;    // the open recordings
thread_local int t_launch_err = 0;
}  // namespace

namespace scae_rec {
bool recording() { return g_open.load(std::memory_order_relaxed) > 0; }
void append(const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, void *const *args,
            const size_t *sizes, int n) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (List *list : g_lists) {
    if (list->stream != st) continue;
    Launch l{fn, grid, block, lds, {}, {}};
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
  }
}
// the hipError_t of the calling thread's last failed hipLaunchKernel (scae::launch), once
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
  List *l = new (std::nothrow) List{(hipStream_t)stream, true, {}};
  if (!l) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  g_lists.push_back(l);
  g_open.fetch_add(1);
  return l;
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
  return (int)static_cast<const List *>(list)->launches.size();
}
extern "C" int scae_launch_list_run(const void *list, void *stream) {
  SCAE_REQUIRE(list);
  const List *ls = static_cast<const List *>(list);
  if (ls->open) return SCAE_ERR_BAD_ARG;   // (still recording: it would record itself)
  for (const Launch &l : ls->launches) {
    void *ptrs[64];
    if (l.at.size() > 64) return SCAE_ERR_UNSUPPORTED;
    unsigned char *base =
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(l.blob.data()));
    for (size_t i = 0; i < l.at.size(); ++i) ptrs[i] = base + l.at[i];
    hipError_t e = hipLaunchKernel(l.fn, l.grid, l.block, ptrs, l.lds, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return SCAE_OK;
}
extern "C" void scae_launch_list_free(void *list) {
  if (!list) return;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    close_list(static_cast<List *>(list));
  }
  delete static_cast<List *>(list);
}
