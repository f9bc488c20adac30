// Tile store and data plane of the C-ABI: library-owned HBM blobs and the collectives between the
// one-process-per-GPU workers, directly over RCCL (xGMI).
//
// Reference counterparts: the tile store of a worker, `Worker._blobs` with create / get / update / destroy
// (spartan/worker.py:70,126-185, spartan/blob_ctx.py:103-254) and the ZeroMQ `get` / `update` exchange between
// workers (spartan/blob_ctx.py:163-179, spartan/rpc/zeromq.py).  A host that brings no device allocator and no
// communication layer of its own (INTEGRATION.md's ctypes host) gets both from here; so does the Python host in this
// repository: its tiles are blobs of this store (spartan_amd/devarray.py) and its data plane between ranks is the
// collectives below (spartan_amd/comm.py).
//
// RCCL is bound at run time (dlopen): the library keeps loading on a host without RCCL.  The copy that is bound is
// the one installed beside the HIP runtime THIS library runs on, and a copy that is linked against another HIP
// runtime (a process may hold two: PyTorch bundles its own libamdhip64 and librccl) is refused -- device pointers,
// streams and events handed to a collective must belong to the runtime the collective runs on.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "sp_common.hpp"

// --------------------------------------------------------------------------------------------------------------
// blobs
// --------------------------------------------------------------------------------------------------------------
namespace {

struct Blob {
  void* ptr;
  size_t bytes;       // allocation size (rounded)
  int32_t dtype, ndim;
  int64_t shape[SP_BLOB_MAX_DIMS];
  int device;
};

std::mutex g_blob_mu;
std::unordered_map<uint64_t, Blob> g_blobs;
uint64_t g_next_blob = 1;
// freed allocations kept for re-use, keyed by (device, rounded size): tiles of an iteration loop have a handful
// of distinct sizes and hipFree synchronises the device
std::unordered_map<uint64_t, std::vector<void*>> g_pool;
size_t g_pool_bytes = 0;

size_t blob_round(size_t n) {
  if (n < 512) return 512;
  if (n < (1u << 21)) {          // powers of two up to 2 MiB
    size_t p = 512;
    while (p < n) p <<= 1;
    return p;
  }
  return (n + (1u << 21) - 1) & ~(size_t)((1u << 21) - 1);   // 2 MiB steps above
}

uint64_t pool_key(int device, size_t bytes) { return ((uint64_t)device << 56) ^ (uint64_t)bytes; }

int blob_lookup(uint64_t h, Blob* out) {
  std::lock_guard<std::mutex> lock(g_blob_mu);
  auto it = g_blobs.find(h);
  if (it == g_blobs.end()) SP_FAIL("unknown blob handle %llu", (unsigned long long)h);
  *out = it->second;
  return 0;
}

// contiguous row-major strides (elements)
void dense_strides(const int64_t* shape, int nd, int64_t* st) {
  int64_t s = 1;
  for (int i = nd - 1; i >= 0; --i) {
    st[i] = s;
    s *= shape[i];
  }
}

int check_box(const Blob& b, const int64_t* ul, const int64_t* lr, int64_t* ext, int64_t* count) {
  *count = 1;
  for (int i = 0; i < b.ndim; ++i) {
    const int64_t u = ul ? ul[i] : 0, l = lr ? lr[i] : b.shape[i];
    if (u < 0 || l > b.shape[i] || u > l) SP_FAIL("blob region [%lld, %lld) outside axis %d of length %lld", (long long)u, (long long)l, i, (long long)b.shape[i]);
    ext[i] = l - u;
    *count *= ext[i];
  }
  return 0;
}

bool box_is_whole(const Blob& b, const int64_t* ext) {
  for (int i = 0; i < b.ndim; ++i)
    if (ext[i] != b.shape[i]) return false;
  return true;
}

// One contiguous byte range?  (every axis before the first cut one has extent 1, every axis after it is whole:
// a run of rows of a matrix, any range of a 1-D blob.)
bool box_is_contiguous(const Blob& b, const int64_t* ext) {
  int i = 0;
  while (i < b.ndim && ext[i] == 1 && b.shape[i] != 1) ++i;     // leading single indices
  if (i < b.ndim) ++i;                                           // the one axis that may be a proper range
  for (; i < b.ndim; ++i)
    if (ext[i] != b.shape[i]) return false;
  return true;
}

}  // namespace

extern "C" int sp_blob_create(const int64_t* shape, int32_t ndim, int32_t dtype, uint64_t* handle) {
  if (!handle || (ndim > 0 && !shape)) SP_FAIL("sp_blob_create: NULL argument");
  if (ndim < 0 || ndim > SP_BLOB_MAX_DIMS) SP_FAIL("sp_blob_create: ndim %d not in 0..%d", ndim, SP_BLOB_MAX_DIMS);
  const size_t es = sp_dtype_size(dtype);
  if (!es) SP_FAIL("sp_blob_create: unknown dtype %d", dtype);
  Blob b;
  memset(&b, 0, sizeof b);
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) SP_FAIL("sp_blob_create: negative extent");
    b.shape[i] = shape[i];
    n *= (size_t)shape[i];
  }
  b.dtype = dtype;
  b.ndim = ndim;
  b.bytes = blob_round(n * es);
  SP_HIP(hipGetDevice(&b.device));
  {
    std::lock_guard<std::mutex> lock(g_blob_mu);
    auto it = g_pool.find(pool_key(b.device, b.bytes));
    if (it != g_pool.end() && !it->second.empty()) {
      b.ptr = it->second.back();
      it->second.pop_back();
      g_pool_bytes -= b.bytes;
    }
  }
  if (!b.ptr) {
    hipError_t e = hipMalloc(&b.ptr, b.bytes);
    if (e != hipSuccess) {
      // out of memory: give the pool back and try once more
      (void)hipGetLastError();
      sp_blob_trim();
      e = hipMalloc(&b.ptr, b.bytes);
    }
    if (e != hipSuccess) SP_FAIL("sp_blob_create: hipMalloc(%zu) failed: %s", b.bytes, hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lock(g_blob_mu);
  *handle = g_next_blob++;
  g_blobs[*handle] = b;
  return 0;
}

extern "C" int sp_blob_destroy(uint64_t h) {
  std::lock_guard<std::mutex> lock(g_blob_mu);
  auto it = g_blobs.find(h);
  if (it == g_blobs.end()) SP_FAIL("sp_blob_destroy: unknown blob handle %llu", (unsigned long long)h);
  // Work already enqueued on the blob runs before any later use of the memory as long as the host keeps to one
  // compute stream per device (the contract of this store, as of most caching allocators).
  g_pool[pool_key(it->second.device, it->second.bytes)].push_back(it->second.ptr);
  g_pool_bytes += it->second.bytes;
  g_blobs.erase(it);
  return 0;
}

extern "C" int sp_blob_trim(void) {
  std::unordered_map<uint64_t, std::vector<void*>> pool;
  {
    std::lock_guard<std::mutex> lock(g_blob_mu);
    pool.swap(g_pool);
    g_pool_bytes = 0;
  }
  for (auto& kv : pool)
    for (void* p : kv.second) SP_HIP(hipFree(p));
  return 0;
}

extern "C" int sp_blob_info(uint64_t h, void** d_ptr, int64_t* shape, int32_t* ndim, int32_t* dtype) {
  Blob b;
  if (blob_lookup(h, &b)) return 1;
  if (d_ptr) *d_ptr = b.ptr;
  if (ndim) *ndim = b.ndim;
  if (dtype) *dtype = b.dtype;
  if (shape)
    for (int i = 0; i < b.ndim; ++i) shape[i] = b.shape[i];
  return 0;
}

extern "C" int sp_blob_stats(int64_t* live_blobs, int64_t* pooled_bytes) {
  std::lock_guard<std::mutex> lock(g_blob_mu);
  if (live_blobs) *live_blobs = (int64_t)g_blobs.size();
  if (pooled_bytes) *pooled_bytes = (int64_t)g_pool_bytes;
  return 0;
}

// host <-> box of a blob.  A whole blob is one async copy; a proper box goes through a packed device staging
// buffer and the strided box-copy kernel (sp_slice_copy), so the host side is always one contiguous transfer.
static int blob_transfer(uint64_t h, void* host, const int64_t* ul, const int64_t* lr, bool to_device, void* stream) {
  Blob b;
  if (blob_lookup(h, &b)) return 1;
  if (!host) SP_FAIL("sp_blob transfer: NULL host pointer");
  int64_t ext[SP_BLOB_MAX_DIMS], count;
  if (check_box(b, ul, lr, ext, &count)) return 1;
  if (count == 0) return 0;
  const size_t es = sp_dtype_size(b.dtype);
  hipStream_t st = (hipStream_t)stream;
  if (box_is_whole(b, ext)) {
    if (to_device) SP_HIP(hipMemcpyAsync(b.ptr, host, (size_t)count * es, hipMemcpyHostToDevice, st));
    else SP_HIP(hipMemcpyAsync(host, b.ptr, (size_t)count * es, hipMemcpyDeviceToHost, st));
    return 0;
  }
  int64_t bst[SP_BLOB_MAX_DIMS], pst[SP_BLOB_MAX_DIMS];
  dense_strides(b.shape, b.ndim, bst);
  dense_strides(ext, b.ndim, pst);
  int64_t off = 0;
  for (int i = 0; i < b.ndim; ++i) off += (ul ? ul[i] : 0) * bst[i];
  char* box = (char*)b.ptr + (size_t)off * es;
  if (box_is_contiguous(b, ext)) {       // one byte range of the blob: a direct transfer, no staging
    if (to_device) SP_HIP(hipMemcpyAsync(box, host, (size_t)count * es, hipMemcpyHostToDevice, st));
    else SP_HIP(hipMemcpyAsync(host, box, (size_t)count * es, hipMemcpyDeviceToHost, st));
    return 0;
  }
  if (b.ndim > 4) SP_FAIL("sp_blob transfer: a proper box of a %d-d blob (boxes are supported up to 4-d)", b.ndim);
  uint64_t stage_h;
  if (sp_blob_create(&count, 1, b.dtype, &stage_h)) return 1;
  Blob stage;
  if (blob_lookup(stage_h, &stage)) return 1;
  int rc = 0;
  if (to_device) {
    SP_HIP(hipMemcpyAsync(stage.ptr, host, (size_t)count * es, hipMemcpyHostToDevice, st));
    rc = sp_slice_copy(box, bst, stage.ptr, pst, ext, b.ndim, (int32_t)es, stream);
  } else {
    rc = sp_slice_copy(stage.ptr, pst, box, bst, ext, b.ndim, (int32_t)es, stream);
    if (!rc) SP_HIP(hipMemcpyAsync(host, stage.ptr, (size_t)count * es, hipMemcpyDeviceToHost, st));
  }
  sp_blob_destroy(stage_h);   // stream-ordered re-use (see sp_blob_destroy)
  return rc;
}

extern "C" int sp_blob_h2d(uint64_t h, const void* host, const int64_t* ul, const int64_t* lr, void* stream) {
  return blob_transfer(h, const_cast<void*>(host), ul, lr, true, stream);
}

// Small driver-side operands (the weight vector of a gradient step, the centres of a k-means iteration) go through a
// ring of pinned staging slots: the caller's buffer is consumed by a host memcpy before the call returns, the copy
// to the device is a true asynchronous DMA from pinned memory, so the host neither waits for the stream (a pageable
// hipMemcpyAsync makes it) nor has to keep its buffer alive.
namespace {
constexpr int kStageSlots = 8;
constexpr size_t kStageBytes = 4u << 20;
struct StageSlot {
  void* host = nullptr;
  hipEvent_t done = nullptr;
  bool used = false;
};
StageSlot g_stage[kStageSlots];
int g_stage_next = 0;
std::mutex g_stage_mu;
}  // namespace

extern "C" int sp_blob_h2d_staged(uint64_t h, const void* host, const int64_t* ul, const int64_t* lr, void* stream,
                                  int32_t* host_consumed) {
  if (host_consumed) *host_consumed = 0;
  Blob b;
  if (blob_lookup(h, &b)) return 1;
  int64_t ext[SP_BLOB_MAX_DIMS], count;
  if (check_box(b, ul, lr, ext, &count)) return 1;
  const size_t bytes = (size_t)count * sp_dtype_size(b.dtype);
  if (!host || bytes == 0 || bytes > kStageBytes || !(box_is_whole(b, ext) || box_is_contiguous(b, ext)))
    return blob_transfer(h, const_cast<void*>(host), ul, lr, true, stream);
  std::lock_guard<std::mutex> lock(g_stage_mu);
  StageSlot& s = g_stage[g_stage_next];
  g_stage_next = (g_stage_next + 1) % kStageSlots;
  if (!s.host) {
    SP_HIP(hipHostMalloc(&s.host, kStageBytes, hipHostMallocDefault));
    SP_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  if (s.used) SP_HIP(hipEventSynchronize(s.done));      // the copy that last read this slot has finished
  memcpy(s.host, host, bytes);
  int64_t bst[SP_BLOB_MAX_DIMS], off = 0;
  dense_strides(b.shape, b.ndim, bst);
  for (int i = 0; i < b.ndim; ++i) off += (ul ? ul[i] : 0) * bst[i];
  char* dst = (char*)b.ptr + (size_t)off * sp_dtype_size(b.dtype);
  SP_HIP(hipMemcpyAsync(dst, s.host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  SP_HIP(hipEventRecord(s.done, (hipStream_t)stream));
  s.used = true;
  if (host_consumed) *host_consumed = 1;
  return 0;
}

// The way back for small results (a glommed gradient, the counts and sums of a k-means iteration): DMA into a pinned
// slot, wait for the stream, copy out -- COMPLETE when the call returns.  A pageable destination would make the
// runtime stage the transfer itself, at a fraction of the rate.
extern "C" int sp_blob_d2h_staged(uint64_t h, void* host, const int64_t* ul, const int64_t* lr, void* stream) {
  Blob b;
  if (blob_lookup(h, &b)) return 1;
  int64_t ext[SP_BLOB_MAX_DIMS], count;
  if (check_box(b, ul, lr, ext, &count)) return 1;
  const size_t bytes = (size_t)count * sp_dtype_size(b.dtype);
  hipStream_t st = (hipStream_t)stream;
  if (!host || bytes == 0 || bytes > kStageBytes || !(box_is_whole(b, ext) || box_is_contiguous(b, ext))) {
    if (blob_transfer(h, host, ul, lr, false, stream)) return 1;
    SP_HIP(hipStreamSynchronize(st));
    return 0;
  }
  std::lock_guard<std::mutex> lock(g_stage_mu);
  StageSlot& s = g_stage[g_stage_next];
  g_stage_next = (g_stage_next + 1) % kStageSlots;
  if (!s.host) {
    SP_HIP(hipHostMalloc(&s.host, kStageBytes, hipHostMallocDefault));
    SP_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  if (s.used) SP_HIP(hipEventSynchronize(s.done));
  int64_t bst[SP_BLOB_MAX_DIMS], off = 0;
  dense_strides(b.shape, b.ndim, bst);
  for (int i = 0; i < b.ndim; ++i) off += (ul ? ul[i] : 0) * bst[i];
  const char* src = (const char*)b.ptr + (size_t)off * sp_dtype_size(b.dtype);
  SP_HIP(hipMemcpyAsync(s.host, src, bytes, hipMemcpyDeviceToHost, st));
  SP_HIP(hipStreamSynchronize(st));
  memcpy(host, s.host, bytes);
  s.used = false;
  return 0;
}

extern "C" int sp_blob_d2h(uint64_t h, void* host, const int64_t* ul, const int64_t* lr, void* stream) {
  return blob_transfer(h, host, ul, lr, false, stream);
}

extern "C" int sp_blob_slice_copy(uint64_t dst, const int64_t* dst_ul, uint64_t src, const int64_t* src_ul,
                                  const int64_t* extent, void* stream) {
  Blob d, s;
  if (blob_lookup(dst, &d) || blob_lookup(src, &s)) return 1;
  if (d.ndim != s.ndim || d.dtype != s.dtype) SP_FAIL("sp_blob_slice_copy: rank / dtype of the two blobs differ");
  if (d.ndim > 4) SP_FAIL("sp_blob_slice_copy: boxes are supported up to 4-d");
  int64_t dst_st[SP_BLOB_MAX_DIMS], src_st[SP_BLOB_MAX_DIMS], doff = 0, soff = 0;
  dense_strides(d.shape, d.ndim, dst_st);
  dense_strides(s.shape, s.ndim, src_st);
  for (int i = 0; i < d.ndim; ++i) {
    const int64_t du = dst_ul ? dst_ul[i] : 0, su = src_ul ? src_ul[i] : 0;
    if (extent[i] < 0 || du < 0 || su < 0 || du + extent[i] > d.shape[i] || su + extent[i] > s.shape[i])
      SP_FAIL("sp_blob_slice_copy: box outside a blob on axis %d", i);
    doff += du * dst_st[i];
    soff += su * src_st[i];
  }
  const size_t es = sp_dtype_size(d.dtype);
  return sp_slice_copy((char*)d.ptr + (size_t)doff * es, dst_st, (const char*)s.ptr + (size_t)soff * es, src_st, extent,
                       d.ndim, (int32_t)es, stream);
}

extern "C" int sp_pinned_alloc(size_t bytes, void** host) {
  if (!host) SP_FAIL("sp_pinned_alloc: NULL argument");
  *host = nullptr;
  SP_HIP(hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault));
  return 0;
}

extern "C" int sp_pinned_free(void* host) {
  if (host) SP_HIP(hipHostFree(host));
  return 0;
}

extern "C" int sp_copy_d2h_async(void* pinned_host, const void* d_src, size_t bytes, void* stream) {
  if (!bytes) return 0;
  if (!pinned_host || !d_src) SP_FAIL("sp_copy_d2h_async: NULL pointer");
  SP_HIP(hipMemcpyAsync(pinned_host, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

// --------------------------------------------------------------------------------------------------------------
// collectives (RCCL)
// --------------------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId;
  decltype(&ncclCommInitRank) CommInitRank;
  decltype(&ncclCommDestroy) CommDestroy;
  decltype(&ncclCommAbort) CommAbort;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError;
  decltype(&ncclGetErrorString) GetErrorString;
  decltype(&ncclGetVersion) GetVersion;
  decltype(&ncclAllReduce) AllReduce;
  decltype(&ncclReduceScatter) ReduceScatter;
  decltype(&ncclReduce) Reduce;
  decltype(&ncclAllGather) AllGather;
  decltype(&ncclBroadcast) Broadcast;
  decltype(&ncclSend) Send;
  decltype(&ncclRecv) Recv;
  decltype(&ncclGroupStart) GroupStart;
  decltype(&ncclGroupEnd) GroupEnd;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

template <typename F>
bool bind(void* h, const char* name, F* out) {
  *out = (F)dlsym(h, name);
  return *out != nullptr;
}

// file that holds the code at `addr` (resolved: no symlinks), "" when unknown
std::string file_of(const void* addr) {
  Dl_info info;
  if (!addr || !dladdr(addr, &info) || !info.dli_fname) return "";
  char real[4096];
  return realpath(info.dli_fname, real) ? std::string(real) : std::string(info.dli_fname);
}

// the HIP runtime this library's own calls go to
std::string own_runtime() { return file_of((const void*)&hipGetDeviceCount); }

std::string g_rccl_path, g_rccl_runtime;

// 0 on success
int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.handle) return 0;
  // search order: $SPARTAN_RCCL_LIB; librccl.so.1 in the directory of the HIP runtime this library is bound to (by
  // full path: a bare soname would resolve to whatever copy the process has mapped already -- e.g. the one PyTorch
  // ships, linked against PyTorch's second HIP runtime); then the loader's search; then the default ROCm prefix
  const std::string mine = own_runtime();
  std::string beside;
  if (!mine.empty() && mine.rfind('/') != std::string::npos) beside = mine.substr(0, mine.rfind('/')) + "/librccl.so.1";
  const char* names[] = {getenv("SPARTAN_RCCL_LIB"), beside.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  std::string refused, last_err;
  for (const char* n : names) {
    if (!n || !*n) continue;
    void* cand = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!cand) {
      const char* e = dlerror();
      last_err = e ? e : "";
      continue;
    }
    // the HIP runtime THAT copy calls into: the symbol as its own dependency tree resolves it
    const std::string theirs = file_of(dlsym(cand, "hipGetDeviceCount"));
    if (!mine.empty() && !theirs.empty() && theirs != mine) {
      refused += std::string(refused.empty() ? "" : "; ") + n + " (" + file_of(dlsym(cand, "ncclGetVersion")) + ") runs on " + theirs;
      dlclose(cand);
      if (n == names[0]) break;          // the copy the user named: do not quietly take another one
      continue;
    }
    h = cand;
    g_rccl_runtime = theirs;
    break;
  }
  if (!h && !refused.empty())
    SP_FAIL("RCCL: every copy found is linked against another HIP runtime than this library's (%s): %s.  Device "
            "pointers and streams of one runtime mean nothing to the other; load the RCCL of this ROCm "
            "(SPARTAN_RCCL_LIB) or keep the second runtime out of the process", mine.c_str(), refused.c_str());
  if (!h) SP_FAIL("RCCL is not available: dlopen(librccl.so.1) failed: %s", last_err.c_str());
  Rccl r;
  r.handle = h;
  const bool ok = bind(h, "ncclGetUniqueId", &r.GetUniqueId) && bind(h, "ncclCommInitRank", &r.CommInitRank) &&
                  bind(h, "ncclCommDestroy", &r.CommDestroy) && bind(h, "ncclCommAbort", &r.CommAbort) &&
                  bind(h, "ncclCommGetAsyncError", &r.CommGetAsyncError) &&
                  bind(h, "ncclGetErrorString", &r.GetErrorString) && bind(h, "ncclGetVersion", &r.GetVersion) &&
                  bind(h, "ncclAllReduce", &r.AllReduce) && bind(h, "ncclReduceScatter", &r.ReduceScatter) &&
                  bind(h, "ncclReduce", &r.Reduce) && bind(h, "ncclAllGather", &r.AllGather) &&
                  bind(h, "ncclBroadcast", &r.Broadcast) && bind(h, "ncclSend", &r.Send) && bind(h, "ncclRecv", &r.Recv) &&
                  bind(h, "ncclGroupStart", &r.GroupStart) && bind(h, "ncclGroupEnd", &r.GroupEnd);
  if (!ok) {
    dlclose(h);
    SP_FAIL("the RCCL library found lacks an expected entry point");
  }
  g_rccl_path = file_of((const void*)r.GetVersion);
  g_rccl = r;
  return 0;
}

struct Comm {
  ncclComm_t comm;
  int world, rank;
};

#define SP_NCCL(expr)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      sp_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_)); \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

int nccl_dtype(int32_t dt, ncclDataType_t* out) {
  switch (dt) {
    case SP_F32: *out = ncclFloat32; return 0;
    case SP_F64: *out = ncclFloat64; return 0;
    case SP_I32: *out = ncclInt32; return 0;
    case SP_I64: *out = ncclInt64; return 0;
    case SP_BOOL:
    case SP_U8: *out = ncclUint8; return 0;
    default: SP_FAIL("collective: unknown dtype %d", dt);
  }
}

// np.add / multiply / maximum / minimum; logical and / or of 0/1 bytes are min / max
int nccl_op(int32_t reducer, int32_t dtype, ncclRedOp_t* out) {
  switch (reducer) {
    case SP_REDUCER_ADD: *out = ncclSum; return 0;
    case SP_REDUCER_MUL: *out = ncclProd; return 0;
    case SP_REDUCER_MAX: *out = ncclMax; return 0;
    case SP_REDUCER_MIN: *out = ncclMin; return 0;
    case SP_REDUCER_AND:
      if (dtype != SP_BOOL) SP_FAIL("collective: logical_and needs bool tiles");
      *out = ncclMin;
      return 0;
    case SP_REDUCER_OR:
      if (dtype != SP_BOOL) SP_FAIL("collective: logical_or needs bool tiles");
      *out = ncclMax;
      return 0;
    default: SP_FAIL("collective: reducer %d has no collective form", reducer);
  }
}

Comm* as_comm(void* c) { return (Comm*)c; }

}  // namespace

extern "C" int sp_comm_available(void) { return rccl_load() == 0 ? 1 : 0; }

// Which files the data plane runs on: the RCCL bound, the HIP runtime that copy calls into, and the HIP runtime
// of this library (the last two are the same file, or sp_comm_* would have refused to load it).
extern "C" int sp_comm_paths(char* rccl_path, char* rccl_runtime_path, char* own_runtime_path, size_t each_bytes) {
  if (!rccl_path || !rccl_runtime_path || !own_runtime_path || each_bytes < 2) SP_FAIL("sp_comm_paths: NULL / empty buffer");
  if (rccl_load()) return 1;
  snprintf(rccl_path, each_bytes, "%s", g_rccl_path.c_str());
  snprintf(rccl_runtime_path, each_bytes, "%s", g_rccl_runtime.c_str());
  snprintf(own_runtime_path, each_bytes, "%s", own_runtime().c_str());
  return 0;
}

extern "C" int sp_comm_version(int* version) {
  if (rccl_load()) return 1;
  SP_NCCL(g_rccl.GetVersion(version));
  return 0;
}

extern "C" int sp_comm_unique_id(void* uid, size_t uid_bytes) {
  if (!uid || uid_bytes < SP_COMM_UID_BYTES) SP_FAIL("sp_comm_unique_id: buffer of %d bytes needed", SP_COMM_UID_BYTES);
  if (rccl_load()) return 1;
  static_assert(sizeof(ncclUniqueId) == SP_COMM_UID_BYTES, "unique id size");
  ncclUniqueId id;
  SP_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(uid, &id, sizeof id);
  return 0;
}

extern "C" int sp_comm_init(int32_t world, int32_t rank, const void* uid, void** comm) {
  if (!uid || !comm) SP_FAIL("sp_comm_init: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) SP_FAIL("sp_comm_init: rank %d of %d", rank, world);
  if (rccl_load()) return 1;
  ncclUniqueId id;
  memcpy(&id, uid, sizeof id);
  Comm* c = new Comm;
  c->world = world;
  c->rank = rank;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    sp_set_error("sp_comm_init: ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    delete c;
    return 1;
  }
  *comm = c;
  return 0;
}

extern "C" int sp_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = as_comm(comm);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  delete c;
  if (r != ncclSuccess) SP_FAIL("sp_comm_destroy: %s", g_rccl.GetErrorString(r));
  return 0;
}

extern "C" int sp_comm_abort(void* comm) {
  if (!comm) return 0;
  Comm* c = as_comm(comm);
  ncclResult_t r = g_rccl.CommAbort(c->comm);
  delete c;
  if (r != ncclSuccess) SP_FAIL("sp_comm_abort: %s", g_rccl.GetErrorString(r));
  return 0;
}

extern "C" int sp_comm_async_error(void* comm) {
  if (!comm) SP_FAIL("sp_comm_async_error: NULL communicator");
  ncclResult_t err = ncclSuccess;
  SP_NCCL(g_rccl.CommGetAsyncError(as_comm(comm)->comm, &err));
  if (err != ncclSuccess) SP_FAIL("asynchronous RCCL error: %s", g_rccl.GetErrorString(err));
  return 0;
}

extern "C" int sp_comm_all_reduce(void* comm, const void* d_src, void* d_dst, int64_t count, int32_t dtype,
                                  int32_t reducer, void* stream) {
  if (!comm) SP_FAIL("sp_comm_all_reduce: NULL communicator");
  ncclDataType_t dt;
  ncclRedOp_t op;
  if (nccl_dtype(dtype, &dt) || nccl_op(reducer, dtype, &op)) return 1;
  if (count == 0) return 0;
  SP_NCCL(g_rccl.AllReduce(d_src, d_dst, (size_t)count, dt, op, as_comm(comm)->comm, (hipStream_t)stream));
  return 0;
}

extern "C" int sp_comm_reduce_scatter(void* comm, const void* d_src, void* d_dst, int64_t recv_count, int32_t dtype,
                                      int32_t reducer, void* stream) {
  if (!comm) SP_FAIL("sp_comm_reduce_scatter: NULL communicator");
  ncclDataType_t dt;
  ncclRedOp_t op;
  if (nccl_dtype(dtype, &dt) || nccl_op(reducer, dtype, &op)) return 1;
  if (recv_count == 0) return 0;
  SP_NCCL(g_rccl.ReduceScatter(d_src, d_dst, (size_t)recv_count, dt, op, as_comm(comm)->comm, (hipStream_t)stream));
  return 0;
}

extern "C" int sp_comm_reduce(void* comm, const void* d_src, void* d_dst, int64_t count, int32_t dtype, int32_t reducer,
                              int32_t root, void* stream) {
  if (!comm) SP_FAIL("sp_comm_reduce: NULL communicator");
  ncclDataType_t dt;
  ncclRedOp_t op;
  if (nccl_dtype(dtype, &dt) || nccl_op(reducer, dtype, &op)) return 1;
  if (root < 0 || root >= as_comm(comm)->world) SP_FAIL("sp_comm_reduce: root %d", root);
  if (count == 0) return 0;
  SP_NCCL(g_rccl.Reduce(d_src, d_dst, (size_t)count, dt, op, root, as_comm(comm)->comm, (hipStream_t)stream));
  return 0;
}

extern "C" int sp_comm_all_gather(void* comm, const void* d_src, void* d_dst, int64_t send_count, int32_t dtype,
                                  void* stream) {
  if (!comm) SP_FAIL("sp_comm_all_gather: NULL communicator");
  ncclDataType_t dt;
  if (nccl_dtype(dtype, &dt)) return 1;
  if (send_count == 0) return 0;
  SP_NCCL(g_rccl.AllGather(d_src, d_dst, (size_t)send_count, dt, as_comm(comm)->comm, (hipStream_t)stream));
  return 0;
}

extern "C" int sp_comm_bcast(void* comm, void* d_buf, int64_t count, int32_t dtype, int32_t root, void* stream) {
  if (!comm) SP_FAIL("sp_comm_bcast: NULL communicator");
  ncclDataType_t dt;
  if (nccl_dtype(dtype, &dt)) return 1;
  if (root < 0 || root >= as_comm(comm)->world) SP_FAIL("sp_comm_bcast: root %d", root);
  if (count == 0) return 0;
  SP_NCCL(g_rccl.Broadcast(d_buf, d_buf, (size_t)count, dt, root, as_comm(comm)->comm, (hipStream_t)stream));
  return 0;
}

extern "C" int sp_comm_all_to_all_blocks(void* comm, int32_t n_sends, const int32_t* send_peers,
                                         const void* const* d_send, const int64_t* send_bytes, int32_t n_recvs,
                                         const int32_t* recv_peers, void* const* d_recv, const int64_t* recv_bytes,
                                         void* stream) {
  if (!comm) SP_FAIL("sp_comm_all_to_all_blocks: NULL communicator");
  if (n_sends < 0 || n_recvs < 0) SP_FAIL("sp_comm_all_to_all_blocks: negative count");
  if ((n_sends && (!send_peers || !d_send || !send_bytes)) || (n_recvs && (!recv_peers || !d_recv || !recv_bytes)))
    SP_FAIL("sp_comm_all_to_all_blocks: NULL list");
  Comm* c = as_comm(comm);
  for (int i = 0; i < n_sends; ++i)
    if (send_peers[i] < 0 || send_peers[i] >= c->world || send_bytes[i] < 0) SP_FAIL("sp_comm_all_to_all_blocks: bad send %d", i);
  for (int i = 0; i < n_recvs; ++i)
    if (recv_peers[i] < 0 || recv_peers[i] >= c->world || recv_bytes[i] < 0) SP_FAIL("sp_comm_all_to_all_blocks: bad receive %d", i);
  if (n_sends == 0 && n_recvs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // one group: RCCL matches the sends and receives of a pair of ranks in the order they were issued, so blocks
  // between the same two ranks must be listed in the same order on both sides (the host's schedules are)
  SP_NCCL(g_rccl.GroupStart());
  ncclResult_t bad = ncclSuccess;
  for (int i = 0; i < n_sends && bad == ncclSuccess; ++i)
    if (send_bytes[i]) bad = g_rccl.Send(d_send[i], (size_t)send_bytes[i], ncclUint8, send_peers[i], c->comm, st);
  for (int i = 0; i < n_recvs && bad == ncclSuccess; ++i)
    if (recv_bytes[i]) bad = g_rccl.Recv(d_recv[i], (size_t)recv_bytes[i], ncclUint8, recv_peers[i], c->comm, st);
  ncclResult_t end = g_rccl.GroupEnd();
  if (bad != ncclSuccess) SP_FAIL("sp_comm_all_to_all_blocks: %s", g_rccl.GetErrorString(bad));
  if (end != ncclSuccess) SP_FAIL("sp_comm_all_to_all_blocks: ncclGroupEnd: %s", g_rccl.GetErrorString(end));
  return 0;
}

// --------------------------------------------------------------------------------------------------------------
// streams and ordering between them (a host without torch needs a compute and a communication stream)
// --------------------------------------------------------------------------------------------------------------
extern "C" int sp_stream_create(void** stream) { return sp_stream_create_priority(stream, 0); }

extern "C" int sp_stream_create_priority(void** stream, int32_t high_priority) {
  if (!stream) SP_FAIL("sp_stream_create: NULL");
  hipStream_t s;
  if (high_priority) {
    int least = 0, greatest = 0;          // numerically lower = higher priority
    SP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    SP_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
  } else {
    SP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  }
  *stream = s;
  return 0;
}

extern "C" int sp_stream_destroy(void* stream) {
  if (stream) SP_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

extern "C" int sp_stream_synchronize(void* stream) {
  SP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" int sp_stream_wait_event(void* stream, void* ev) {
  if (!ev) SP_FAIL("sp_stream_wait_event: NULL event");
  SP_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return 0;
}

extern "C" int sp_stream_query(void* stream, int32_t* done) {
  if (!done) SP_FAIL("sp_stream_query: NULL");
  hipError_t e = hipStreamQuery((hipStream_t)stream);
  if (e == hipSuccess) *done = 1;
  else if (e == hipErrorNotReady) {
    *done = 0;
    (void)hipGetLastError();
  } else SP_FAIL("sp_stream_query: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int sp_set_device(int32_t device) {
  SP_HIP(hipSetDevice(device));
  return 0;
}

extern "C" int sp_get_device(int32_t* device) {
  if (!device) SP_FAIL("sp_get_device: NULL pointer");
  int d = 0;
  SP_HIP(hipGetDevice(&d));
  *device = d;
  return 0;
}
