// Device allocations of the library (pf_malloc / pf_free): big arrays are PHYSICALLY CONTIGUOUS.
//
// Round-4 finding (tools/stream_vs_memory_probe.py, tools/vram_region_probe.hip; profiles/r04/alloc_*.log): on MI355X the
// bandwidth a kernel gets depends on where its arrays lie PHYSICALLY, by up to 20 % (a 512-plane column sweep: 5.1 to 6.2
// TB/s; a plain device copy: 4.5 to 5.2 TB/s), and hipMalloc does not give the same physical layout twice: a freed block
// is not the next one handed out, so one process alternated between 2.31 and 2.55 ms per spectral 512^3 step from one
// handle to the next -- on the SAME virtual addresses and the SAME stream.  That is the "allocation lottery" of rounds 2-3;
// it is not visible to user space (no physical addresses) and no layout or stream choice of the library changes it.
// Three policies were measured (PFHIP_ALLOC):
//   contiguous  hipExtMallocWithFlags(hipDeviceMallocContiguous): one physical run per array.  Every handle of every
//               process gets the same time to 0.3 % (spectral 512^3: 2.25 ms chunked, 2.45 whole-box), at or slightly
//               better than hipMalloc's slow state for every workload.  DEFAULT: reproducible, and layout tuning means
//               something again (virtual offsets within an array are physical offsets).
//   plain       hipMalloc: two states per workload (spectral 2.06 / 2.27 chunked), chosen by the allocator.
//   scatter[:KiB]  the array assembled from 2 MiB (or KiB) physical pieces (hipMemCreate) mapped into one reserved range
//               in shuffled order: a distribution instead of two states (spectral 2.03 ... 2.60, BM3 0.72 ... 0.75 ms);
//               its best cases are the best seen, but which case a handle gets is again the driver's choice of pieces.
//               Pieces below 2 MiB (the page-table fragment the TLBs hold in one entry) cost 2-4x.
// Allocations below 32 MiB are plain hipMallocs, and so is everything of the BE-parity mode (csrc/fem_be.hip calls hipMalloc
// itself): its band and dense blocks showed no lottery, and on contiguous memory its steps were 16 % SLOWER (BM3 186 vs
// 160 ms, BM2 58.5 vs 50.2: profiles/r04/alloc_fem_be.log).  If the contiguous (or virtual-memory) calls fail the array is a plain
// hipMalloc and pf_alloc_describe() says so -- results never depend on the placement.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pfhip_internal.h"

namespace pfhip {
namespace {

struct Block {
  size_t bytes;   // mapped size (a multiple of the piece size)
};
std::mutex g_mu;
std::map<void*, Block> g_blocks;       // scattered blocks alive
std::string g_fallback;                // why the last scattered allocation fell back to hipMalloc ("" = it did not)
constexpr size_t kScatterMin = 32u << 20;

struct Policy {
  int mode = 1;            // 0 plain, 1 contiguous (default), 2 scatter
  size_t piece = 2u << 20;
};
Policy policy() {
  Policy p;
  const char* e = getenv("PFHIP_ALLOC");
  if (!e || !*e) return p;
  if (!strcmp(e, "plain")) p.mode = 0;
  else if (!strcmp(e, "contiguous")) p.mode = 1;
  else if (!strncmp(e, "scatter", 7)) {
    p.mode = 2;
    const long kib = e[7] == ':' ? atol(e + 8) : 0;
    if (kib >= 64 && kib <= (1 << 20) && (kib & (kib - 1)) == 0) p.piece = (size_t)kib << 10;
  }
  return p;
}

hipError_t scatter_alloc(void** out, size_t bytes, size_t piece) {
  const size_t n = (bytes + piece - 1) / piece, total = n * piece;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  hipError_t r = hipGetDevice(&prop.location.id);
  if (r != hipSuccess) return r;
  hipDeviceptr_t va = nullptr;
  r = hipMemAddressReserve(&va, total, 0, nullptr, 0);
  if (r != hipSuccess) return r;
  // a fixed pseudo-random order (the physical address of each piece is the driver's choice; the order only has to be
  // unrelated to the order the driver hands the pieces out in)
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  for (size_t i = n - 1; i > 0 && n > 1; --i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    std::swap(order[i], order[(size_t)((st >> 33) % (i + 1))]);
  }
  size_t mapped = 0;
  for (; mapped < n; ++mapped) {
    hipMemGenericAllocationHandle_t hnd;
    r = hipMemCreate(&hnd, piece, &prop, 0);
    if (r != hipSuccess) break;
    r = hipMemMap((hipDeviceptr_t)((char*)va + order[mapped] * piece), piece, 0, hnd, 0);
    (void)hipMemRelease(hnd);   // the mapping keeps the piece alive; an unmapped piece is freed here
    if (r != hipSuccess) break;
  }
  if (r == hipSuccess) {
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    r = hipMemSetAccess(va, total, &acc, 1);
  }
  if (r != hipSuccess) {
    for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((hipDeviceptr_t)((char*)va + order[i] * piece), piece);
    (void)hipMemAddressFree(va, total);
    (void)hipGetLastError();
    return r;
  }
  *out = (void*)va;
  std::lock_guard<std::mutex> lk(g_mu);
  g_blocks[*out] = Block{total};
  return hipSuccess;
}

}  // namespace

// TEST HOOK (pfk_set_tuning key 11): fill every new allocation with this byte (-1 = off).  Fresh processes get zero pages
// from the driver, recycled memory is anything: a kernel that reads what nobody wrote passes every test of a fresh process
// and fails in the field (csrc/fem_be.hip did, round 4).  tests/test_gpu_parity.py runs every scheme with a finite pattern.
int g_alloc_fill = -1;
void pf_alloc_set_fill(int byte) { g_alloc_fill = byte < 0 ? -1 : (byte & 0xFF); }

static hipError_t pf_malloc_bytes_raw(void** p, size_t bytes);
hipError_t pf_malloc_bytes(void** p, size_t bytes) {
  hipError_t r = pf_malloc_bytes_raw(p, bytes);
  if (r == hipSuccess && g_alloc_fill >= 0) {
    r = hipMemset(*p, g_alloc_fill, bytes);
    if (r == hipSuccess) r = hipDeviceSynchronize();   // (the library initialises on its own non-blocking streams)
  }
  return r;
}

static hipError_t pf_malloc_bytes_raw(void** p, size_t bytes) {
  *p = nullptr;
  const Policy pol = policy();
  if (bytes >= kScatterMin && pol.mode == 2) {
    const hipError_t r = scatter_alloc(p, bytes, pol.piece);
    if (r == hipSuccess) return r;
    if (r == hipErrorOutOfMemory) return r;
    std::lock_guard<std::mutex> lk(g_mu);
    g_fallback = std::string("scattered allocation unavailable (") + hipGetErrorString(r) + "): plain hipMalloc";
    *p = nullptr;
  } else if (bytes >= kScatterMin && pol.mode == 1) {
    const hipError_t r = hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
    if (r == hipSuccess) return r;
    (void)hipGetLastError();
    *p = nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    g_fallback = std::string("contiguous allocation of ") + std::to_string(bytes >> 20) + " MiB unavailable (" + hipGetErrorString(r) + "): plain hipMalloc";
  }
  return hipMalloc(p, bytes);
}

hipError_t pf_free(void* p) {
  if (!p) return hipSuccess;
  size_t total = 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_blocks.find(p);
    if (it != g_blocks.end()) {
      total = it->second.bytes;
      g_blocks.erase(it);
    }
  }
  if (!total) return hipFree(p);
  hipError_t r = hipDeviceSynchronize();   // hipFree's implicit synchronisation, which hipMemUnmap does not have
  if (r != hipSuccess) return r;
  r = hipMemUnmap((hipDeviceptr_t)p, total);
  if (r != hipSuccess) return r;
  return hipMemAddressFree((hipDeviceptr_t)p, total);
}

// one clause for the status strings of the handles
std::string pf_alloc_describe() {
  const Policy pol = policy();
  std::lock_guard<std::mutex> lk(g_mu);
  if (pol.mode != 0 && !g_fallback.empty()) return "alloc: WARNING " + g_fallback;
  if (pol.mode == 2) return "alloc: big arrays scattered in " + std::to_string(pol.piece >> 10) + " KiB pieces (PFHIP_ALLOC)";
  return pol.mode == 1 ? "alloc: big arrays physically contiguous" : "alloc: plain hipMalloc (PFHIP_ALLOC)";
}

}  // namespace pfhip
