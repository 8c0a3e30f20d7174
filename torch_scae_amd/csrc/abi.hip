// ABI bookkeeping of libscae_hip.so (see include/scae_hip.h).
#include "common.h"

#include <string.h>

extern "C" int scae_abi_version(void) { return SCAE_ABI_VERSION; }

extern "C" const char *scae_error_string(int code) {
  if (code == SCAE_OK) return "ok";
  if (code == SCAE_ERR_BAD_ARG) return "scae: null pointer or non-positive size";
  if (code == SCAE_ERR_UNSUPPORTED) return "scae: shape outside this build's kernel limits";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "scae: unknown error";
}

// ---- launch lists (common.h: scae::launch) ------------------------------------------------
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
  std::vector<Launch> launches;
};
std::atomic<bool> g_on{false};
std::mutex g_mu;       // (forward and backward launches come from different host threads)
List *g_list = nullptr;
}  // namespace

namespace scae_rec {
bool recording() { return g_on.load(std::memory_order_relaxed); }
void append(const void *fn, dim3 grid, dim3 block, size_t lds, void *const *args,
            const size_t *sizes, int n) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_list) return;
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
  g_list->launches.push_back(std::move(l));
}
}  // namespace scae_rec

extern "C" int scae_launch_list_begin(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_list) return SCAE_ERR_BAD_ARG;   // one recording at a time
  g_list = new List();
  g_on.store(true);
  return SCAE_OK;
}
extern "C" void *scae_launch_list_end(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  List *l = g_list;
  g_list = nullptr;
  g_on.store(false);
  return l;
}
extern "C" int scae_launch_list_size(const void *list) {
  return list ? (int)static_cast<const List *>(list)->launches.size() : 0;
}
extern "C" int scae_launch_list_run(const void *list, void *stream) {
  SCAE_REQUIRE(list);
  for (const Launch &l : static_cast<const List *>(list)->launches) {
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
extern "C" void scae_launch_list_free(void *list) { delete static_cast<List *>(list); }
