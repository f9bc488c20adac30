// Combine (Tile.merge), strided box copy, STREAM copy and the small runtime
// helpers of the C-ABI (errors, device info, HIP-event timing).
//
// sp_update   <- spartan/array/tile.pyx:200-297  merge(), dense->dense branch
// sp_slice_copy <- spartan/array/distarray.py:355-365 (fetch stitch),
//                  tile.pyx:64-113 (Tile.get(subslice)),
//                  sparse.pyx:297-301 (multiple_slice dense branch)
// All are HBM-bound streaming kernels: 16 B per lane when the innermost
// extent allows it, grid-stride over at most 8 workgroups per CU.
#include <stdarg.h>

#include "sp_interp.hpp"

// ------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";

void sp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sp_last_error(void) { return g_err; }
extern "C" int sp_abi_version(void) { return SP_ABI_VERSION; }

extern "C" int sp_device_count(int* count) {
  if (!count) SP_FAIL("sp_device_count: NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    SP_FAIL("hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return 0;
}

extern "C" int sp_device_info(int device, int* cu_count, int64_t* hbm_bytes, char* name, size_t name_len) {
  hipDeviceProp_t prop;
  SP_HIP(hipGetDeviceProperties(&prop, device));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (name && name_len) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return 0;
}

// ------------------------------------------------------------------- events
extern "C" int sp_event_create(void** ev) {
  if (!ev) SP_FAIL("sp_event_create: NULL");
  hipEvent_t e;
  SP_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return 0;
}
extern "C" int sp_event_destroy(void* ev) {
  SP_HIP(hipEventDestroy((hipEvent_t)ev));
  return 0;
}
extern "C" int sp_event_record(void* ev, void* stream) {
  SP_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
extern "C" int sp_event_synchronize(void* ev) {
  SP_HIP(hipEventSynchronize((hipEvent_t)ev));
  return 0;
}
extern "C" int sp_event_query(void* ev, int32_t* done) {
  if (!done) SP_FAIL("sp_event_query: NULL");
  hipError_t e = hipEventQuery((hipEvent_t)ev);
  if (e == hipSuccess) *done = 1;
  else if (e == hipErrorNotReady) {
    *done = 0;
    (void)hipGetLastError();
  } else SP_FAIL("sp_event_query: %s", hipGetErrorString(e));
  return 0;
}
extern "C" int sp_device_synchronize(void) {
  SP_HIP(hipDeviceSynchronize());
  return 0;
}
extern "C" int sp_memset(void* d_dst, int32_t value, size_t bytes, void* stream) {
  if (!bytes) return 0;
  if (!d_dst) SP_FAIL("sp_memset: NULL pointer");
  SP_HIP(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
  return 0;
}
extern "C" int sp_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) SP_FAIL("sp_event_elapsed_ms: NULL");
  SP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}

static inline int grid_for(int64_t n) {
  int64_t b = (n + SP_BLOCK - 1) / SP_BLOCK;
  const int64_t cap = (int64_t)SP_CUS * SP_BLOCKS_PER_CU;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------- STREAM copy
// Full grid, one 16-B vector per lane: the structure that reaches the box's
// achievable copy bandwidth (tools/hbm_probe.hip: 6.2 TB/s vs 4.7 TB/s for a
// capped grid that strides).  This is the "measured HBM bandwidth" yardstick.
// NT (copies bigger than the L2): non-temporal loads and stores, 6.23 -> 6.65 TB/s on 2 GiB.
template <bool NT>
__global__ __launch_bounds__(SP_BLOCK) void sp_stream_copy_kernel(float4* __restrict__ dst,
                                                                  const float4* __restrict__ src,
                                                                  int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * SP_BLOCK;
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4* s4 = (const v4*)src;
  v4* d4 = (v4*)dst;
  for (int64_t i = (int64_t)blockIdx.x * SP_BLOCK + threadIdx.x; i < n16; i += stride) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + i), d4 + i);
    else d4[i] = s4[i];
  }
}
__global__ void sp_byte_copy_kernel(uint8_t* dst, const uint8_t* src, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

extern "C" int sp_stream_copy(void* d_dst, const void* d_src, size_t bytes, void* stream) {
  return sp_stream_copy_wg(d_dst, d_src, bytes, 0, stream);
}

extern "C" int sp_stream_copy_wg(void* d_dst, const void* d_src, size_t bytes, int32_t max_workgroups, void* stream) {
  if (!bytes) return 0;
  if (!d_dst || !d_src) SP_FAIL("sp_stream_copy: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const bool al = ((((uintptr_t)d_dst) | ((uintptr_t)d_src)) & 15) == 0;
  size_t main_bytes = al ? (bytes / 16) * 16 : 0;
  if (main_bytes) {
    int64_t blocks = ((int64_t)(main_bytes / 16) + SP_BLOCK - 1) / SP_BLOCK;
    if (blocks > (1LL << 30)) blocks = 1LL << 30;
    if (max_workgroups > 0 && blocks > max_workgroups) blocks = max_workgroups;   // (the kernel strides over the rest)
    if (main_bytes >= (size_t)SP_STREAM_ELEMS * 4)
      hipLaunchKernelGGL(sp_stream_copy_kernel<true>, dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st,
                         (float4*)d_dst, (const float4*)d_src, (int64_t)(main_bytes / 16));
    else
      hipLaunchKernelGGL(sp_stream_copy_kernel<false>, dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st,
                         (float4*)d_dst, (const float4*)d_src, (int64_t)(main_bytes / 16));
    SP_CHECK_LAUNCH();
  }
  if (bytes - main_bytes) {
    hipLaunchKernelGGL(sp_byte_copy_kernel, dim3(grid_for(bytes - main_bytes)), dim3(SP_BLOCK), 0, st,
                       (uint8_t*)d_dst + main_bytes, (const uint8_t*)d_src + main_bytes,
                       (int64_t)(bytes - main_bytes));
    SP_CHECK_LAUNCH();
  }
  return 0;
}

// ----------------------------------------------------------- strided copy
struct BoxDesc {
  int32_t ndim;
  int64_t shape[SP_MAX_DIMS];
  int64_t dstride[SP_MAX_DIMS];
  int64_t sstride[SP_MAX_DIMS];
};

// W = bytes moved per thread step (one element of W bytes; the caller folds
// contiguous inner runs into 16-B elements when alignment allows).
template <typename E>
__global__ __launch_bounds__(SP_BLOCK) void sp_box_copy_kernel(E* __restrict__ dst,
                                                               const E* __restrict__ src, BoxDesc b,
                                                               int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * SP_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * SP_BLOCK + threadIdx.x; i < n; i += stride) {
    int64_t rem = i, doff = 0, soff = 0;
#pragma unroll
    for (int d = SP_MAX_DIMS - 1; d >= 0; --d) {
      if (d < b.ndim) {
        int64_t c;
        if (d == 0) {
          c = rem;
        } else {
          int64_t q = rem / b.shape[d];
          c = rem - q * b.shape[d];
          rem = q;
        }
        doff += c * b.dstride[d];
        soff += c * b.sstride[d];
      }
    }
    dst[doff] = src[soff];
  }
}

// 2-D transposing copy through LDS: dst[c][r] = src[r][c].  Both the global read
// and the global write are coalesced (64 consecutive 4/8-byte elements per wave
// row); the 64x65 LDS tile (one pad column) makes the column-wise read conflict
// free.  Used when a Transpose view (transpose.py:64-67 `base_tile.transpose()`)
// has to be handed to a kernel as a dense operand.
template <typename E>
__global__ __launch_bounds__(SP_BLOCK) void sp_transpose_kernel(E* __restrict__ dst, const E* __restrict__ src,
                                                                int64_t R, int64_t C, int64_t src_ld,
                                                                int64_t dst_ld) {
  __shared__ E tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  const int64_t tiles_c = (C + 63) / 64;
  const int64_t ntiles = ((R + 63) / 64) * tiles_c;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t r = r0 + ty + 4 * k, c = c0 + tx;
      if (r < R && c < C) tile[ty + 4 * k][tx] = src[r * src_ld + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t c = c0 + ty + 4 * k, r = r0 + tx;
      if (r < R && c < C) dst[c * dst_ld + r] = tile[tx][ty + 4 * k];
    }
    __syncthreads();
  }
}

extern "C" int sp_slice_copy(void* d_dst, const int64_t* dst_stride, const void* d_src,
                             const int64_t* src_stride, const int64_t* shape, int32_t ndim,
                             int32_t elem_size, void* stream) {
  if (ndim < 0 || ndim > SP_MAX_DIMS) SP_FAIL("sp_slice_copy: ndim=%d unsupported (max %d)", ndim, SP_MAX_DIMS);
  if (elem_size != 1 && elem_size != 4 && elem_size != 8) SP_FAIL("sp_slice_copy: elem_size=%d", elem_size);
  if (!d_dst || !d_src) SP_FAIL("sp_slice_copy: NULL pointer");
  BoxDesc b;
  memset(&b, 0, sizeof(b));
  int64_t n = 1;
  if (ndim == 0) {
    b.ndim = 1;
    b.shape[0] = 1;
    b.dstride[0] = b.sstride[0] = 1;
  } else {
    b.ndim = ndim;
    for (int d = 0; d < ndim; ++d) {
      if (shape[d] < 0) SP_FAIL("sp_slice_copy: negative extent");
      b.shape[d] = shape[d];
      b.dstride[d] = dst_stride[d];
      b.sstride[d] = src_stride[d];
      n *= shape[d];
    }
  }
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // 2-D transpose pattern (dst dense rows, src walks a column): LDS-tiled kernel
  if (b.ndim == 2 && (elem_size == 4 || elem_size == 8) && b.dstride[1] == 1 && b.sstride[0] == 1 &&
      b.sstride[1] >= b.shape[0] && b.dstride[0] >= b.shape[1] && b.shape[0] >= 16 && b.shape[1] >= 16) {
    // dst[i][j] = src[j][i] with src row-major of leading dimension sstride[1]
    const int64_t R = b.shape[1], Cc = b.shape[0];   // src is R x Cc
    int64_t blocks = ((R + 63) / 64) * ((Cc + 63) / 64);
    const int64_t cap = (int64_t)SP_CUS * SP_BLOCKS_PER_CU * 4;
    if (blocks > cap) blocks = cap;
    if (elem_size == 4)
      hipLaunchKernelGGL((sp_transpose_kernel<uint32_t>), dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st,
                         (uint32_t*)d_dst, (const uint32_t*)d_src, R, Cc, b.sstride[1], b.dstride[0]);
    else
      hipLaunchKernelGGL((sp_transpose_kernel<uint64_t>), dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st,
                         (uint64_t*)d_dst, (const uint64_t*)d_src, R, Cc, b.sstride[1], b.dstride[0]);
    SP_CHECK_LAUNCH();
    return 0;
  }
  // widen: if the innermost run is contiguous on both sides, move 16-B words
  const int last = b.ndim - 1;
  int64_t es = elem_size;
  if (b.dstride[last] == 1 && b.sstride[last] == 1) {
    const int64_t per16 = 16 / es;
    bool ok = (b.shape[last] % per16 == 0) && ((((uintptr_t)d_dst) | ((uintptr_t)d_src)) & 15) == 0;
    for (int d = 0; d < last && ok; ++d)
      if (b.dstride[d] % per16 != 0 || b.sstride[d] % per16 != 0) ok = false;
    if (ok) {
      b.shape[last] /= per16;
      for (int d = 0; d < last; ++d) {
        b.dstride[d] /= per16;
        b.sstride[d] /= per16;
      }
      n /= per16;
      es = 16;
    }
  }
  const dim3 g(grid_for(n)), blk(SP_BLOCK);
  switch (es) {
    case 16:
      hipLaunchKernelGGL((sp_box_copy_kernel<float4>), g, blk, 0, st, (float4*)d_dst, (const float4*)d_src, b, n);
      break;
    case 8:
      hipLaunchKernelGGL((sp_box_copy_kernel<uint64_t>), g, blk, 0, st, (uint64_t*)d_dst, (const uint64_t*)d_src, b, n);
      break;
    case 4:
      hipLaunchKernelGGL((sp_box_copy_kernel<uint32_t>), g, blk, 0, st, (uint32_t*)d_dst, (const uint32_t*)d_src, b, n);
      break;
    default:
      hipLaunchKernelGGL((sp_box_copy_kernel<uint8_t>), g, blk, 0, st, (uint8_t*)d_dst, (const uint8_t*)d_src, b, n);
      break;
  }
  SP_CHECK_LAUNCH();
  return 0;
}

// -------------------------------------------------------------------- merge
struct UpdDesc {
  int32_t ndim;
  int64_t box[SP_MAX_DIMS];      // extents of the updated box
  int64_t dstride[SP_MAX_DIMS];  // element strides of the tile
  int64_t doff;                  // element offset of the box origin in the tile
  int32_t dst_dtype, src_dtype, reducer, mask_mode;
};

template <typename T>
__device__ __forceinline__ T sp_apply_reducer(int r, T old, T upd) {
  switch (r) {
    case SP_REDUCER_ADD: return old + upd;
    case SP_REDUCER_MUL: return old * upd;
    case SP_REDUCER_MAX: return sp_nanmax<T>(old, upd);
    case SP_REDUCER_MIN: return sp_nanmin<T>(old, upd);
    case SP_REDUCER_AND: return (T)((old != (T)0) && (upd != (T)0));
    case SP_REDUCER_OR: return (T)((old != (T)0) || (upd != (T)0));
    default: return upd;
  }
}

// T = arithmetic type of the destination tile (float / double / int64 for all
// integer + bool tiles).  V consecutive elements of the innermost box
// dimension per thread (contiguous in both tile and update).
template <typename T, int V>
__global__ __launch_bounds__(SP_BLOCK) void sp_update_kernel(void* __restrict__ dst,
                                                             const void* __restrict__ src,
                                                             uint8_t* __restrict__ mask, UpdDesc u,
                                                             int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * SP_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * SP_BLOCK + threadIdx.x; i < nvec; i += stride) {
    const int64_t L = i * V;  // linear index inside the box (== offset in src)
    int64_t rem = L, off = u.doff;
#pragma unroll
    for (int d = SP_MAX_DIMS - 1; d >= 0; --d) {
      if (d < u.ndim) {
        int64_t c;
        if (d == 0) {
          c = rem;
        } else {
          int64_t q = rem / u.box[d];
          c = rem - q * u.box[d];
          rem = q;
        }
        off += c * u.dstride[d];
      }
    }
    T upd[V], old[V], res[V];
    sp_load_vec<T, V>(src, u.src_dtype, L, upd);
    bool m[V];
    if (u.mask_mode == SP_MASK_ARRAY) {
#pragma unroll
      for (int v = 0; v < V; ++v) m[v] = mask[off + v] != 0;
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) m[v] = u.mask_mode == SP_MASK_ALL_SET;
    }
    const bool need_old = u.reducer != SP_REDUCER_NONE && u.mask_mode != SP_MASK_ALL_CLEAR;
    if (need_old) {
      sp_load_vec<T, V>(dst, u.dst_dtype, off, old);
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) old[v] = (T)0;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) res[v] = m[v] ? sp_apply_reducer<T>(u.reducer, old[v], upd[v]) : upd[v];
    sp_store_vec<T, V>(dst, u.dst_dtype, off, res);
    if (mask) {
#pragma unroll
      for (int v = 0; v < V; ++v) mask[off + v] = 1;
    }
  }
}

template <typename T>
static int sp_update_launch(void* d_dst, const void* d_src, uint8_t* d_mask, const UpdDesc& u, int64_t n,
                            hipStream_t st) {
  constexpr int VV = sp_cls<T>::V;
  const int last = u.ndim - 1;
  bool vec = (u.box[last] % VV == 0) && (u.doff % VV == 0) &&
             ((((uintptr_t)d_dst) | ((uintptr_t)d_src)) & 15) == 0;
  for (int d = 0; d < last && vec; ++d)
    if (u.dstride[d] % VV != 0) vec = false;
  if (vec) {
    hipLaunchKernelGGL((sp_update_kernel<T, VV>), dim3(grid_for(n / VV)), dim3(SP_BLOCK), 0, st, d_dst, d_src,
                       d_mask, u, n / VV);
  } else {
    hipLaunchKernelGGL((sp_update_kernel<T, 1>), dim3(grid_for(n)), dim3(SP_BLOCK), 0, st, d_dst, d_src,
                       d_mask, u, n);
  }
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sp_update(void* d_dst, int32_t dst_dtype, const int64_t* dst_shape, int32_t ndim,
                         const int64_t* ul, const int64_t* lr, const void* d_src, int32_t src_dtype,
                         int32_t reducer, int32_t mask_mode, uint8_t* d_mask, void* stream) {
  if (ndim < 0 || ndim > SP_MAX_DIMS) SP_FAIL("sp_update: ndim=%d unsupported (max %d)", ndim, SP_MAX_DIMS);
  if (!d_dst || !d_src) SP_FAIL("sp_update: NULL pointer");
  if (dst_dtype < 0 || dst_dtype >= SP_DTYPE_COUNT || src_dtype < 0 || src_dtype >= SP_DTYPE_COUNT)
    SP_FAIL("sp_update: bad dtype");
  if (reducer < SP_REDUCER_NONE || reducer > SP_REDUCER_OR) SP_FAIL("sp_update: bad reducer %d", reducer);
  if (mask_mode < SP_MASK_ALL_CLEAR || mask_mode > SP_MASK_ARRAY) SP_FAIL("sp_update: bad mask_mode");
  if (mask_mode == SP_MASK_ARRAY && !d_mask) SP_FAIL("sp_update: SP_MASK_ARRAY needs d_mask");
  UpdDesc u;
  memset(&u, 0, sizeof(u));
  u.dst_dtype = dst_dtype;
  u.src_dtype = src_dtype;
  u.reducer = reducer;
  u.mask_mode = mask_mode;
  int64_t n = 1;
  if (ndim == 0) {
    // zero-dimensional tile (tile.pyx:212-217): a single cell
    u.ndim = 1;
    u.box[0] = 1;
    u.dstride[0] = 1;
    u.doff = 0;
  } else {
    u.ndim = ndim;
    int64_t stride = 1;
    for (int d = ndim - 1; d >= 0; --d) {
      if (ul[d] < 0 || lr[d] > dst_shape[d] || ul[d] > lr[d])
        SP_FAIL("sp_update: box [%lld,%lld) outside tile extent %lld on dim %d", (long long)ul[d],
                (long long)lr[d], (long long)dst_shape[d], d);
      u.box[d] = lr[d] - ul[d];
      u.dstride[d] = stride;
      u.doff += ul[d] * stride;
      stride *= dst_shape[d];
      n *= u.box[d];
    }
  }
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (dst_dtype) {
    case SP_F32: return sp_update_launch<float>(d_dst, d_src, d_mask, u, n, st);
    case SP_F64: return sp_update_launch<double>(d_dst, d_src, d_mask, u, n, st);
    default: return sp_update_launch<int64_t>(d_dst, d_src, d_mask, u, n, st);
  }
}

// ---- cumulative sum / product along one axis of a dense tile viewed as [outer, axis_len, inner] ----
// (the per-tile `scan_fn(tile, axis)` of the reference's scan operator, spartan/expr/operator/scan.py:42-63;
// np.cumsum / np.cumprod).  inner > 1: one thread per (outer, inner) line walks the axis in order --
// coalesced across `inner`, NumPy's own sequential order.  inner == 1: one wavefront per line, 64
// elements at a time (shuffle scan inside the chunk, running carry between chunks).
template <typename T, bool PROD>
__device__ __forceinline__ T sp_scan_op(T a, T b) { return PROD ? a * b : a + b; }

template <typename T, bool PROD>
__global__ __launch_bounds__(256) void sp_cumscan_cols_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t O,
                                                              int64_t A, int64_t I) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= O * I) return;
  const int64_t o = e / I, i = e - o * I;
  const T* p = in + o * A * I + i;
  T* q = out + o * A * I + i;
  T acc = PROD ? (T)1 : (T)0;
  for (int64_t a = 0; a < A; ++a) {
    acc = sp_scan_op<T, PROD>(acc, p[a * I]);
    q[a * I] = acc;
  }
}

template <typename T, bool PROD>
__global__ __launch_bounds__(256) void sp_cumscan_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t O,
                                                              int64_t A) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t o = wave; o < O; o += nwaves) {
    T carry = PROD ? (T)1 : (T)0;
    for (int64_t a0 = 0; a0 < A; a0 += 64) {
      const int64_t a = a0 + lane;
      T v = a < A ? in[o * A + a] : (PROD ? (T)1 : (T)0);
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const T u = __shfl_up(v, off);
        if (lane >= off) v = sp_scan_op<T, PROD>(u, v);
      }
      v = sp_scan_op<T, PROD>(carry, v);
      if (a < A) out[o * A + a] = v;
      carry = __shfl(v, 63);
    }
  }
}

extern "C" int sp_cumscan(const void* d_in, void* d_out, int32_t dtype, int64_t outer, int64_t axis_len, int64_t inner,
                          int32_t product, void* stream) {
  if (outer < 0 || axis_len < 0 || inner < 0) SP_FAIL("sp_cumscan: negative size");
  if (outer == 0 || axis_len == 0 || inner == 0) return 0;
  if (!d_in || !d_out) SP_FAIL("sp_cumscan: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
#define SP_SCAN_GO(T)                                                                                             \
  do {                                                                                                            \
    if (inner > 1) {                                                                                              \
      const int64_t n = outer * inner;                                                                            \
      const unsigned blocks = (unsigned)((n + 255) / 256);                                                        \
      if (product) hipLaunchKernelGGL((sp_cumscan_cols_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)d_in, (T*)d_out, outer, axis_len, inner); \
      else hipLaunchKernelGGL((sp_cumscan_cols_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)d_in, (T*)d_out, outer, axis_len, inner); \
    } else {                                                                                                      \
      int64_t blocks = (outer + 3) / 4;                                                                           \
      if (blocks > SP_CUS * 16) blocks = SP_CUS * 16;                                                             \
      if (product) hipLaunchKernelGGL((sp_cumscan_rows_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)d_in, (T*)d_out, outer, axis_len); \
      else hipLaunchKernelGGL((sp_cumscan_rows_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)d_in, (T*)d_out, outer, axis_len); \
    }                                                                                                             \
  } while (0)
  switch (dtype) {
    case SP_F32: SP_SCAN_GO(float); break;
    case SP_F64: SP_SCAN_GO(double); break;
    case SP_I64: SP_SCAN_GO(int64_t); break;
    case SP_I32: SP_SCAN_GO(int32_t); break;
    default: SP_FAIL("sp_cumscan: unsupported dtype %d", dtype);
  }
#undef SP_SCAN_GO
  SP_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------ row gather (integer-array indexing)
// dst[i, :] = src[idx[i], :] for rows of `row_bytes` bytes: the tile body of the reference's _int_index_mapper
// (spartan/expr/operator/filter.py:50-75, one src.select(row) per index there).  One thread per 4-byte word
// (16-byte words when the row length and both bases allow); negative indices count from the end like NumPy's.
template <typename W>
__global__ __launch_bounds__(256) void sp_gather_rows_kernel(const W* __restrict__ src, int64_t src_row_words,
                                                             const int64_t* __restrict__ idx, int64_t n_idx,
                                                             int64_t n_src_rows, int64_t row_words,
                                                             W* __restrict__ dst) {
  const int64_t total = n_idx * row_words;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / row_words, w = e - i * row_words;
    int64_t r = idx[i];
    if (r < 0) r += n_src_rows;
    dst[e] = src[r * src_row_words + w];
  }
}

extern "C" int sp_gather_rows(const void* d_src, int64_t src_row_stride_bytes, int64_t n_src_rows, const int64_t* d_idx,
                              int64_t n_idx, int64_t row_bytes, void* d_dst, void* stream) {
  if (n_idx < 0 || row_bytes < 0 || n_src_rows < 0 || src_row_stride_bytes < row_bytes) SP_FAIL("sp_gather_rows: bad sizes");
  if (n_idx == 0 || row_bytes == 0) return 0;
  if (!d_src || !d_idx || !d_dst) SP_FAIL("sp_gather_rows: NULL pointer");
  if (row_bytes % 4 || src_row_stride_bytes % 4) SP_FAIL("sp_gather_rows: rows must be a multiple of 4 bytes");
  hipStream_t st = (hipStream_t)stream;
  const bool wide = row_bytes % 16 == 0 && src_row_stride_bytes % 16 == 0 && ((uintptr_t)d_src % 16) == 0 &&
                    ((uintptr_t)d_dst % 16) == 0;
  const int64_t words = wide ? row_bytes / 16 : row_bytes / 4;
  int64_t blocks = (n_idx * words + 255) / 256;
  const int64_t cap = (int64_t)SP_CUS * SP_BLOCKS_PER_CU * 4;
  if (blocks > cap) blocks = cap;
  if (wide)
    hipLaunchKernelGGL((sp_gather_rows_kernel<float4>), dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)d_src,
                       src_row_stride_bytes / 16, d_idx, n_idx, n_src_rows, words, (float4*)d_dst);
  else
    hipLaunchKernelGGL((sp_gather_rows_kernel<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t*)d_src,
                       src_row_stride_bytes / 4, d_idx, n_idx, n_src_rows, words, (uint32_t*)d_dst);
  SP_CHECK_LAUNCH();
  return 0;
}
