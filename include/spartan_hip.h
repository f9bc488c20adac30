/*
 * spartan_hip.h -- C-ABI of libspartan_hip.so, the MI355X (gfx950) tile-kernel
 * backend for Spartan-style lazy arrays.
 *
 * The reference has no FFI for this path: its per-tile "kernels" are NumPy calls
 * made inside the worker (spartan/expr/operator/local.py:115-127) and its
 * alternate-backend precedent is ParakeetExpr (local.py:187-209).  Each entry
 * point below names the reference call site whose *body* it replaces; the
 * Python host (spartan_amd/backend_hip.py) binds them with ctypes, and
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the failure
 *     text is available from sp_last_error() (thread-local), mirroring the
 *     reference's RemoteException-carries-traceback convention
 *     (spartan/rpc/common.py:43-47,165-167).
 *   - all pointers named d_* are DEVICE pointers into HBM (tile blobs are
 *     allocated by the host; the library never owns tile memory).
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on
 *     that stream; no call synchronises the device.
 *   - no torch / C++ types cross this boundary.
 */
#ifndef SPARTAN_HIP_H_
#define SPARTAN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_ABI_VERSION 1
#define SP_BLOB_MAX_DIMS 8     /* rank of a library-owned blob (boxes of one: up to 4-d) */
#define SP_COMM_UID_BYTES 128  /* size of the rendezvous token of sp_comm_unique_id / sp_comm_init */

/* ---- element types of tile blobs (spartan/array/tile.pyx:34-46: a Tile is
 *      shape + dtype + dense data) ---- */
enum sp_dtype {
  SP_F32 = 0,
  SP_F64 = 1,
  SP_I32 = 2,
  SP_I64 = 3,
  SP_BOOL = 4, /* numpy bool_: one byte, 0/1 */
  SP_U8 = 5,
  SP_DTYPE_COUNT = 6
};

/* ---- fused LocalExpr program --------------------------------------------
 * A serialised FnCallExpr tree (spartan/expr/operator/local.py:76-127) after
 * MapMapFusion / ReduceMapFusion (spartan/expr/operator/optimize.py:133-227).
 * It is a straight-line register program over SP_NREG virtual registers,
 * evaluated per element in ONE arithmetic class (`cls`): float (SP_F32),
 * double (SP_F64) or int64 (SP_I64).  Inputs are converted to the class on
 * load; SP_OP_TO_* ops re-normalise a value to a narrower NumPy dtype where
 * the reference's intermediate would have been narrower.  Register j <
 * n_inputs is pre-loaded with input j; every other register must be written
 * before it is read and result_reg must have been written (the evaluators do
 * not clear their register file: a program that breaks the rule is refused).
 */
#define SP_MAX_INPUTS 8
#define SP_MAX_INSTR 64
#define SP_MAX_CONSTS 16
#define SP_MAX_DIMS 4
#define SP_NREG 8

enum sp_opcode {
  SP_OP_NOP = 0,
  /* leaves */
  SP_OP_CONST = 1, /* dst = consts[a] */
  SP_OP_IOTA = 2,  /* dst = linear element index inside this tile (row-major) */
  SP_OP_MOV = 3,
  /* binary arithmetic (numpy ufunc named in comment) */
  SP_OP_ADD = 10,      /* np.add */
  SP_OP_SUB = 11,      /* np.subtract */
  SP_OP_MUL = 12,      /* np.multiply */
  SP_OP_DIV = 13,      /* np.divide / np.true_divide (float classes) */
  SP_OP_FLOORDIV = 14, /* np.floor_divide */
  SP_OP_MOD = 15,      /* np.mod / np.remainder (sign of divisor) */
  SP_OP_FMOD = 16,     /* np.fmod (sign of dividend) */
  SP_OP_POW = 17,      /* np.power */
  SP_OP_MAX = 18,      /* np.maximum (NaN-propagating) */
  SP_OP_MIN = 19,      /* np.minimum (NaN-propagating) */
  /* comparisons -> 0/1 */
  SP_OP_EQ = 20,
  SP_OP_NE = 21,
  SP_OP_LT = 22,
  SP_OP_LE = 23,
  SP_OP_GT = 24,
  SP_OP_GE = 25,
  /* logical on truthiness -> 0/1 */
  SP_OP_LAND = 26,
  SP_OP_LOR = 27,
  SP_OP_LXOR = 28,
  SP_OP_LNOT = 29,
  /* unary */
  SP_OP_NEG = 30,
  SP_OP_ABS = 31,
  SP_OP_SQRT = 32,
  SP_OP_SQUARE = 33,
  SP_OP_EXP = 34,
  SP_OP_LOG = 35,
  SP_OP_RECIP = 36,
  SP_OP_SIGN = 37,
  SP_OP_FLOOR = 38,
  SP_OP_CEIL = 39,
  SP_OP_TANH = 40,
  SP_OP_NORM_CDF = 41, /* standard normal CDF, 0.5 * erfc(-x / sqrt(2)) (statistics.py:224-225) */
  /* ternary: dst = a ? b : c */
  SP_OP_WHERE = 45,
  /* dtype normalisation inside a wider class */
  SP_OP_TO_F32 = 50,  /* round to float32 */
  SP_OP_TO_I32 = 51,  /* C cast to int32 (wraps / truncates toward zero) */
  SP_OP_TO_I64 = 52,  /* truncate toward zero */
  SP_OP_TO_BOOL = 53, /* x != 0 */
  SP_OP_TO_U8 = 54,
  /* one operand a constant: dst = reg[b] (op) consts[a] -- the emitter folds `x + 1`, `x * 0.5`, `2 - x` ... into ONE
   * instruction instead of a CONST into a register followed by the operator (a third fewer dispatches for the
   * interpreter tier on typical fused trees; the specialised tiers compile to the same code either way). */
  SP_OP_ADDC = 60,
  SP_OP_SUBC = 61,  /* reg - const */
  SP_OP_RSUBC = 62, /* const - reg */
  SP_OP_MULC = 63,
  SP_OP_DIVC = 64,  /* reg / const */
  SP_OP_RDIVC = 65, /* const / reg */
  SP_OP_MAXC = 66,
  SP_OP_MINC = 67
};

typedef struct sp_instr {
  uint8_t op, dst, a, b, c, pad0, pad1, pad2;
} sp_instr;

typedef struct sp_program {
  int32_t cls; /* SP_F32 | SP_F64 | SP_I64 : arithmetic class */
  int32_t n_inputs;
  int32_t n_instr;
  int32_t result_reg; /* register holding the value to store / reduce */
  int32_t ndim;       /* 1..SP_MAX_DIMS, collapsed row-major OUTPUT index space */
  int32_t out_dtype;  /* sp_dtype of the output blob (map) */
  int32_t linear;     /* 1: every input is dense with the output's layout (or a
                         scalar): offset == linear index, vectorised path */
  int32_t pad;
  int64_t shape[SP_MAX_DIMS];                       /* output index space */
  int64_t in_stride[SP_MAX_INPUTS][SP_MAX_DIMS];    /* in ELEMENTS; 0 = broadcast */
  int32_t in_dtype[SP_MAX_INPUTS];
  double consts[SP_MAX_CONSTS];
  int64_t iconsts[SP_MAX_CONSTS]; /* same constants for the int64 class */
  sp_instr instr[SP_MAX_INSTR];
} sp_program;

/* ---- reductions (spartan/expr/operator/reduce.py:21-70: the local reduce
 *      `op.evaluate(ctx)` e.g. data.sum(axis), mathematics.py:126-127) ---- */
enum sp_reduce_op {
  SP_RED_SUM = 0,  /* np.add      */
  SP_RED_PROD = 1, /* np.multiply */
  SP_RED_MAX = 2,  /* np.maximum  */
  SP_RED_MIN = 3,  /* np.minimum  */
  SP_RED_AND = 4,  /* np.logical_and over truthiness */
  SP_RED_OR = 5    /* np.logical_or  */
};

/* ---- combine (spartan/array/tile.pyx:200-297 `merge`) ---- */
enum sp_reducer {
  SP_REDUCER_NONE = 0, /* replace */
  SP_REDUCER_ADD = 1,
  SP_REDUCER_MUL = 2,
  SP_REDUCER_MAX = 3,
  SP_REDUCER_MIN = 4,
  SP_REDUCER_AND = 5,
  SP_REDUCER_OR = 6
};

/* tile mask states: tile.pyx:15-16 MASK_ALL_CLEAR / MASK_ALL_SET, or an
 * explicit per-element byte mask in HBM */
enum sp_mask_mode { SP_MASK_ALL_CLEAR = 0, SP_MASK_ALL_SET = 1, SP_MASK_ARRAY = 2 };

/* ------------------------------------------------------------------------ */
int sp_abi_version(void);
const char* sp_last_error(void);

/* Number of HIP devices / properties of one (CU count, bytes of HBM). */
int sp_device_count(int* count);
int sp_device_info(int device, int* cu_count, int64_t* hbm_bytes, char* name, size_t name_len);

/* sp_map_fused: one coalesced launch evaluating `prog` for every element of
 * the output tile.  Replaces `op.evaluate(op_ctx)` + `tile.from_data(result)`
 * in tile_mapper (spartan/expr/operator/map.py:48-88).
 *   d_inputs[j] : device pointer of input j (element type prog->in_dtype[j])
 *   d_out       : dense row-major output, prod(shape) elements of out_dtype
 */
int sp_map_fused(const sp_program* prog, const void* const* d_inputs, void* d_out, void* stream);

/* sp_program_static_id: hot fp32 shapes (x+c, a*b, x*x+x, x*(yp-y), plain x ...)
 * run on kernels specialised at build time on their instruction stream
 * (spartan_amd/csrc/sp_interp.hpp StaticProg); everything else runs on the
 * generic interpreter kernels.  Returns the library id the program would use,
 * or -1.  out_dtype < 0 = "any" (reductions).  SP_NO_STATIC=1 in the
 * environment disables the specialised kernels (A/B measurements). */
int sp_program_static_id(const sp_program* prog, int32_t out_dtype);

/* Run-time specialisation: a program OUTSIDE that library, applied to a large
 * tile (>= min_elems elements, default 2^22), is compiled once per instruction
 * stream with hipRTC from the same evaluator source (spartan_amd/csrc/sp_jit.hip;
 * the reference's precedent is its JIT local op, local.py:187-209) and cached for
 * the life of the process.  The compile (~0.4 s) runs on a background thread: the
 * launch that requests it, and every launch until it is ready, uses the interpreter
 * kernels, as do smaller tiles and programs whose compile fails.  Results are
 * bit-identical across the three tiers, so the switch is not observable.
 * SP_JIT_SYNC=1 in the environment compiles in the calling thread instead.
 *   sp_jit_configure      enabled: 0/1, or -1 to leave; min_elems: threshold, or -1 to leave
 *                         (defaults: SP_NO_JIT / SP_JIT_MIN_ELEMS in the environment).
 *                         Returns 1 if the tier is usable (libhiprtc present and enabled).
 *   sp_jit_wait           block until every queued specialisation is ready.
 *   sp_jit_compiled_count kernels specialised so far.
 *   sp_jit_compile_check  does `template_expr` (a kernel template-id naming the program
 *                         type StaticProg<1000>) compile for `prog`?  Needs no device. */
/* The code objects hipRTC produced persist across processes (one file per library build x header x kernel x
 * program; later processes load them instead of compiling): in $SPARTAN_JIT_CACHE, else
 * $XDG_CACHE_HOME/spartan_amd/jit, else ~/.cache/spartan_amd/jit.  SPARTAN_JIT_CACHE=off keeps them in memory only. */
int sp_jit_configure(int enabled, long long min_elems);
void sp_jit_wait(void);
/* Loads every code object of csrc/jit_seed and of SPARTAN_JIT_CACHE onto `device` (about a millisecond each) so that
 * the first launch of a seeded program does not wait for the file; the backend calls it on a background thread
 * when it comes up.  Returns the number of functions loaded. */
int sp_jit_preload(int device);
void sp_jit_shutdown(void);         /* stop the compile thread (an in-flight compile finishes first) */
int sp_jit_compiled_count(void);
int sp_jit_compile_check(const char* header, const char* template_expr, const sp_program* prog);
/* Seed mode: between sp_jit_seed_begin(dir) (NULL / "": <library directory>/jit_seed) and sp_jit_seed_end() every
 * specialisation a launch asks for is compiled at once and WRITTEN to dir instead of being loaded -- no device is
 * needed, the launches themselves fail.  A process that later finds no code object for a program in its own cache
 * looks there before compiling.  sp_jit_seed_end returns the number of code objects written. */
int sp_jit_seed_begin(const char* dir);
int sp_jit_seed_end(void);

/* sp_reduce: fused map -> reduce over ONE axis of the program's index space
 * viewed as [outer, axis_len, inner] (prod == prod(prog->shape)); axis=None is
 * outer=1, inner=1.  Replaces `_reduce_mapper`'s local reduction
 * (reduce.py:54) incl. the ReduceMapFusion prologue (optimize.py:190-227).
 *   d_out : [outer, inner] elements of out_dtype.
 *   d_ws  : scratch of at least sp_reduce_workspace_bytes(...) bytes.
 */
size_t sp_reduce_workspace_bytes(int32_t cls, int64_t outer, int64_t axis_len, int64_t inner);
int sp_reduce(const sp_program* prog, const void* const* d_inputs, int32_t red_op, int64_t outer,
              int64_t axis_len, int64_t inner, void* d_out, int32_t out_dtype, void* d_ws,
              size_t ws_bytes, void* stream);

/* sp_argreduce: single-pass argmax/argmin with first-occurrence tie-break,
 * emitting int64 indices `index_offset + a` (a = position along the reduced
 * axis) and the extreme values.  Replaces the reference's three-pass
 * formulation (max-reduce, _arg_mapper, min-reduce:
 * spartan/expr/sorting.py:67-123); results are bit-identical.
 *   which: 0 = argmax, 1 = argmin.   nan_index: index reported when the
 *   extreme is NaN (the reference's `a == b` never matches NaN, so every
 *   candidate becomes the sentinel prod(array_shape), sorting.py:84).
 *   d_out_val: [outer, inner] of the program class's dtype (may be NULL).
 */
size_t sp_argreduce_workspace_bytes(int32_t cls, int64_t outer, int64_t axis_len, int64_t inner);
int sp_argreduce(const sp_program* prog, const void* const* d_inputs, int32_t which, int64_t outer,
                 int64_t axis_len, int64_t inner, int64_t index_offset, int64_t nan_index,
                 int64_t* d_out_idx, void* d_out_val, void* d_ws, size_t ws_bytes, void* stream);

/* sp_update: Tile.merge(old, subslice, update, reducer), dense->dense branch
 * (spartan/array/tile.pyx:250-283).  dst is a dense row-major tile of
 * dst_shape; the update covers the box [ul, lr) of it; src is dense row-major
 * of the box shape.  mask_mode says what the tile's mask was BEFORE the call;
 * with SP_MASK_ARRAY d_mask (bytes, tile shape) is read and set to 1 over the
 * box; otherwise d_mask, if non-NULL, is only written (set to 1 over the box).
 * Cells whose mask was clear are replaced (`update.astype`), cells whose mask
 * was set get reducer(old, update) (or replace when reducer == NONE).
 */
int sp_update(void* d_dst, int32_t dst_dtype, const int64_t* dst_shape, int32_t ndim,
              const int64_t* ul, const int64_t* lr, const void* d_src, int32_t src_dtype,
              int32_t reducer, int32_t mask_mode, uint8_t* d_mask, void* stream);

/* sp_slice_copy: strided <=4-d box copy (element size 1/4/8 bytes) used by
 * DistArrayImpl.fetch's stitch `tgt[dst_slice] = result`
 * (spartan/array/distarray.py:355-365), Tile.get(subslice) (tile.pyx:64-113)
 * and sparse.multiple_slice's dense branch (sparse.pyx:297-301).
 * Strides are in ELEMENTS.
 */
int sp_slice_copy(void* d_dst, const int64_t* dst_stride, const void* d_src,
                  const int64_t* src_stride, const int64_t* shape, int32_t ndim,
                  int32_t elem_size, void* stream);

/* sp_gemm_f32: C[M,N] (+)= A[M,K] . B[K,N], fp32 in / fp32 accumulate on the
 * f32 MFMA (v_mfma_f32_32x32x2_f32).  Row-major with leading dimensions in
 * elements.  Replaces `tiles[0].dot(tiles[1])` in dot_map2_mapper /
 * dot_outer_mapper / dot_map2_np_mapper (spartan/expr/dot.py:172-238).
 * accumulate != 0 computes C += A.B (the np.add reducer of the dot target,
 * dot.py:289-294, fused into the epilogue).
 */
int sp_gemm_f32(const float* d_A, int64_t lda, const float* d_B, int64_t ldb, float* d_C,
                int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t accumulate, void* stream);

/* sp_gemm_f64: the same contract in float64 on the f64 MFMA (v_mfma_f64_16x16x4_f64); the reference's
 * builders produce float64 arrays by default, so `spartan.dot` of two such matrices lands here. */
int sp_gemm_f64(const double* d_A, int64_t lda, const double* d_B, int64_t ldb, double* d_C,
                int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t accumulate, void* stream);

/* sp_gemm_ws: sp_gemm_f32 / sp_gemm_f64 (dtype SP_F32 | SP_F64) with scratch for SPLIT-K.  A product
 * with fewer than 256 output tiles and a long contraction (x^T.x of a tall matrix,
 * ridge_regression.py:18-19) would leave most CUs idle; with d_ws of at least
 * sp_gemm_workspace_bytes(...) bytes the contraction is cut into slices computed side by side and the
 * partial products are added in slice order (deterministic, no atomics).  Without scratch, or when no
 * split pays, it is exactly sp_gemm_f32 / sp_gemm_f64. */
size_t sp_gemm_workspace_bytes(int32_t dtype, int64_t M, int64_t N, int64_t K);
int sp_gemm_ws(int32_t dtype, const void* d_A, int64_t lda, const void* d_B, int64_t ldb, void* d_C,
               int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t accumulate, void* d_ws,
               size_t ws_bytes, void* stream);

/* Matrix . vector products (dot.py:180-183, dot_map2_np_mapper with a 1-D rhs;
 * the lreg step X.w and X^T.r) are HBM-bound and have no GEMM entry point:
 * the host lowers them to sp_reduce with the fused program MUL(in0, in1) and
 * a broadcast stride on the vector operand (row kernels for A.x, column
 * kernels for A^T.x). */

/* sp_rowdot_colsum_f32: out[c] (+)= sum_i X[i][c] * (X[i,:] . w - y[i])  (y == NULL: no subtraction) in ONE pass
 * over the row tile X [n][d] -- the fused form of `sum(x * (dot(x, w) - y), axis=0)`, the gradient of the
 * least-squares workload (spartan/examples/linear_regression.py:10-24: a matrix.vector launch and a fused map ->
 * column-reduce launch, X read twice).  A row stays in its wavefront's registers between the two uses: 4 <= d <= 4096,
 * d % 4 == 0, rows of X, w and out 16-byte aligned, y with stride ldy.  Deterministic (per-wave partial sums in the
 * workspace, added in wave order); agrees with the two-launch form to rounding, not bit for bit. */
size_t sp_rowdot_colsum_workspace_bytes(int64_t n, int64_t d);          /* 0: shape not supported */
int sp_rowdot_colsum_f32(const float* d_x, int64_t ldx, int64_t n, int64_t d, const float* d_w, const float* d_y,
                         int64_t ldy, float* d_out, int32_t accumulate, void* d_ws, size_t ws_bytes, void* stream);

/* k-means tile kernels: the bodies of the reference's k-means mappers
 * (spartan/examples/sklearn/cluster/k_means_.py), BASELINE configs[3].
 *
 * sp_nearest_center: d_labels[i] = np.argmin(cdist(points, centers), axis=1)[i]
 * (kmeans_map2_dist_mapper :61-66, kmeans_outer_dist_mapper :52-58; also
 * _find_closest :11-28).  points [n, d] (SP_F32 | SP_F64, row stride ldx
 * elements), centers [k, d] (SP_F32 | SP_F64, row stride ldc).  tier:
 *   SP_NEAREST_EXACT (1): squared differences accumulated in fp64 in feature
 *     order, sqrt, first minimum -- cdist's arithmetic, bit for bit;
 *   SP_NEAREST_FUSED (2): fp32 MFMA GEMM with the argmin fused into the epilogue
 *     (the n x k distance matrix is never written) + the exact kernel for the
 *     points whose two best scores are within the fp32 error bound: SAME labels;
 *   SP_NEAREST_SPLIT (4): the same filter with every fp32 operand cut into two bf16 numbers and the
 *     contraction on the bf16 matrix pipe (three exact-product MFMAs per 16 features, fp32 accumulation,
 *     16 x the fp32 pipe's rate); its wider error window sends more points to the exact re-check: SAME labels
 *     (csrc/kmeans_split.hpp);
 *   SP_NEAREST_AUTO (0): split (>= 32 features, and >= 192 centers unless the points come prepared;
 *     SP_KM_SPLIT=0: fused) or fused for fp32 points when n*k*d >= 2^24, else exact.
 *   SP_NEAREST_FUSED_UNCHECKED (3) / SP_NEAREST_SPLIT_UNCHECKED (5): diagnostics only -- the filter alone; points it
 *     could not decide are left as -1 - (fp32 best) (used to report the re-check rate).
 * d_ws: sp_nearest_center_workspace_bytes(n, k, d) bytes of scratch.
 *
 * sp_bincount_i64: np.bincount(labels, minlength=k)[:k] (kmeans_count_mapper :69-72);
 * labels outside [0, k) are ignored.  k <= 16384.
 *
 * sp_segment_sum: d_out[c, :] = points[labels == c].sum(axis=0) (kmeans_center_mapper
 * :75-97, _find_cluster_mapper :35-42).  After a stable counting sort of the row
 * ids by label, the rows of a label are added per feature column in ascending row
 * order in chunks of 512 rows, and the chunk sums are added in order: deterministic,
 * no floating-point atomics, and for a label with <= 512 rows exactly NumPy's axis-0
 * order (bit-identical); larger labels differ from it by rounding only.
 * d_out [k, d] of the points' dtype; k <= 16384.
 * sp_segment_sum_counts: the same, and d_counts[c] (NULL: not wanted) = the number of rows with label c =
 * sp_bincount_i64 of the labels -- the counting sort has the number anyway, so a k-means iteration that needs
 * both (k_means_.py:69-72 and :75-97) reads the labels once less and launches two kernels fewer.
 */
#define SP_NEAREST_AUTO 0
#define SP_NEAREST_EXACT 1
#define SP_NEAREST_FUSED 2
#define SP_NEAREST_FUSED_UNCHECKED 3
#define SP_NEAREST_SPLIT 4
#define SP_NEAREST_SPLIT_UNCHECKED 5
size_t sp_nearest_center_workspace_bytes(int64_t n, int64_t k, int64_t d);
int sp_nearest_center(const void* d_points, int32_t dtype, int64_t ldx, const void* d_centers,
                      int32_t cdtype, int64_t ldc, int64_t n, int64_t k, int64_t d, int64_t* d_labels,
                      int32_t tier, void* d_ws, size_t ws_bytes, void* stream);
/* The points of a k-means fit do not change between its iterations: sp_kmeans_points_prepare cuts them ONCE into what
 * the split tier reads (two bf16 images and |x|^2 per point, sp_kmeans_points_prepared_bytes(n, d) bytes) and
 * sp_nearest_center_prepared is sp_nearest_center with that buffer handed in (its workspace is then
 * sp_nearest_center_prepared_workspace_bytes(n, k, d): without room for the images).  The caller answers for the
 * buffer matching the points it passes: same n, d, and contents not written since. */
size_t sp_kmeans_points_prepared_bytes(int64_t n, int64_t d);
int sp_kmeans_points_prepare(const float* d_points, int64_t ldx, int64_t n, int64_t d, void* d_prepared, size_t bytes,
                             void* stream);
size_t sp_nearest_center_prepared_workspace_bytes(int64_t n, int64_t k, int64_t d);
int sp_nearest_center_prepared(const void* d_points, int32_t dtype, int64_t ldx, const void* d_prepared,
                               const void* d_centers, int32_t cdtype, int64_t ldc, int64_t n, int64_t k, int64_t d,
                               int64_t* d_labels, int32_t tier, void* d_ws, size_t ws_bytes, void* stream);
int sp_bincount_i64(const int64_t* d_labels, int64_t n, int64_t k, int64_t* d_counts, void* stream);
size_t sp_segment_sum_workspace_bytes(int64_t n, int64_t k, int64_t d);
int sp_segment_sum(const void* d_points, int32_t dtype, int64_t ldx, const int64_t* d_labels, int64_t n,
                   int64_t k, int64_t d, void* d_out, void* d_ws, size_t ws_bytes, void* stream);
int sp_segment_sum_counts(const void* d_points, int32_t dtype, int64_t ldx, const int64_t* d_labels, int64_t n,
                          int64_t k, int64_t d, void* d_out, int64_t* d_counts, void* d_ws, size_t ws_bytes,
                          void* stream);

/* sp_random_fill: the per-tile bodies of the reference's random builders
 * (spartan/expr/srandom.py:38-55, np.random.rand / randn / randint per tile) as a
 * counter-based generator: element i = Philox4x32-10(seed, offset + i), so a tile's
 * content is independent of the launch geometry.  kind 0: uniform [0,1), 1: standard
 * normal, 2: integers in [lo, hi).  dtype: SP_F32 | SP_F64 | SP_I32 | SP_I64.
 * (Not NumPy's stream; the reference re-seeds every worker from the clock.) */
int sp_random_fill(void* d_out, int32_t dtype, int64_t n, int32_t kind, uint64_t seed, uint64_t offset,
                   int64_t lo, int64_t hi, void* stream);

/* sp_cumscan: np.cumsum (product == 0) / np.cumprod (product != 0) along one axis of a dense tile viewed
 * as [outer, axis_len, inner] -- the per-tile `scan_fn(tile, axis)` of the scan operator
 * (spartan/expr/operator/scan.py:42-63).  dtype: SP_F32 | SP_F64 | SP_I32 | SP_I64, in and out alike. */
int sp_cumscan(const void* d_in, void* d_out, int32_t dtype, int64_t outer, int64_t axis_len, int64_t inner,
               int32_t product, void* stream);

/* ---- sparse tiles ---------------------------------------------------------------
 * Device format of a sparse tile: canonical CSR -- int64 indptr[nrows + 1], int32 column indices ascending
 * inside a row, no duplicate coordinates, values SP_F32 | SP_F64.  Replaces the scipy.sparse objects the
 * reference keeps in Tile.data (spartan/array/tile.pyx:149-156) and the conversions done on every use
 * (sparse.pyx:232-242 convert_sparse_array, dot.py:212-216 tocsr()).
 *
 * sp_coo_to_csr: (row, col, value) list in any order -> canonical CSR.  Entries with row < 0 are dropped;
 * entries with equal coordinates are added in list order (stable LSD radix sort of the row*ncols+col keys,
 * no atomics).  One primitive carries: upload of a mapper's scipy matrix, transpose (rows <-> cols),
 * slicing (sp_coo_box then this; sparse.pyx:198-230 slice / slice_coo, :289-341 multiple_slice[_coo]),
 * region updates (sparse.pyx:246-286 compute_sparse_update), A + B (concatenated lists; scipy's `+` behind
 * np.add on two sparse tiles) and the reduction of the sparse x sparse expansion.
 * d_indices / d_vals_out need room for nnz entries; the number kept is indptr[nrows] (read it back). */
size_t sp_coo_to_csr_workspace_bytes(int64_t nnz);
int sp_coo_to_csr(int32_t dtype, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t* d_rows,
                  const int32_t* d_cols, const void* d_vals, int64_t* d_indptr, int32_t* d_indices,
                  void* d_vals_out, void* d_ws, size_t ws_bytes, void* stream);
/* row index of every stored entry (CSR -> COO). */
int sp_csr_rows(int64_t nrows, int64_t nnz, const int64_t* d_indptr, int32_t* d_rows, void* stream);
/* In-place edit of a COO list against the box [r0, r1) x [c0, c1): drop_inside == 0 keeps the entries inside
 * (the others get row = -1) and shifts them by (dr, dc); drop_inside != 0 drops the entries inside and shifts
 * the others. */
int sp_coo_box(int64_t nnz, int32_t* d_rows, int32_t* d_cols, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
               int64_t dr, int64_t dc, int32_t drop_inside, void* stream);
/* In-place reshape of a COO list (the sparse branch of Reshape.fetch, spartan/expr/operator/reshape.py:181-193):
 * entry (r, c) of a matrix with old_cols columns sits at linear position r * old_cols + c - offset of the target
 * [new_rows, new_cols]; entries outside the target get row = -1. */
int sp_coo_reshape(int64_t nnz, int32_t* d_rows, int32_t* d_cols, int64_t old_cols, int64_t offset, int64_t new_rows,
                   int64_t new_cols, void* stream);
/* sp_csr_spmm: C[m, n] (+)= A[m, k] (CSR) x B[k, n] (dense, row-major, ldb) -- the tile body of
 * dot_map2_mapper / dot_outer_mapper when tile_a is sparse (spartan/expr/dot.py:193-240, scipy's csr .dot)
 * and of dot_coo_dense_unordered_map (sparse.pyx:103-158, n == 1; the result is written dense).
 * d_b == NULL with n == 1 multiplies by a vector of ones (row sums).  n == 1: workgroups own
 * 2048 consecutive stored entries (coalesced loads, products in LDS, one thread per row, carries of rows that
 * span chunks added by a fix-up pass in chunk order -- no floating-point atomics; needs the workspace);
 * n == 1, rows of >= 1024 entries on average (or no workspace): 2..64 lanes per row, shuffle reduction; n > 1: entries of a row in storage order
 * (scipy's csr_matvecs order).  Without a workspace the lanes-per-row kernel is used for every n == 1.
 * d_plan (may be NULL): the output of sp_csr_spmv_plan for this matrix -- the first row that starts in each
 * 2048-entry chunk, the length of the longest row and an arrival counter, sp_csr_spmv_plan_entries(nnz) int64 values
 * -- computed once per matrix and reused by every multiply (an iteration like p <- W.p keeps W).  With a plan the
 * n == 1 product is ONE launch (rows no longer than 65 entries are summed whole by the chunk they start in; longer
 * rows: carries added by the last workgroup to arrive, which uses -- and resets -- the counter in the plan, so one
 * plan serves one stream at a time); without it every workgroup searches indptr itself and a fix-up launch follows. */
size_t sp_csr_spmm_workspace_bytes(int64_t nnz, int64_t n);
int64_t sp_csr_spmv_plan_entries(int64_t nnz);
int sp_csr_spmv_plan(int64_t m, int64_t nnz, const int64_t* d_indptr, int64_t* d_plan, void* stream);
int sp_csr_spmm(int32_t dtype, int64_t m, int64_t k, int64_t n, int64_t nnz, const int64_t* d_indptr,
                const int32_t* d_indices, const void* d_vals, const void* d_b, int64_t ldb, void* d_c, int64_t ldc,
                int32_t accumulate, const int64_t* d_plan, void* d_ws, size_t ws_bytes, void* stream);
/* Column-blocked y (+)= A . x for an fp32 CSR tile whose rows are sorted by column (csrc/spmv_blocked.hip): the same
 * product as sp_csr_spmm with n == 1, bit for bit, for matrices whose row blocks draw most of their columns from a
 * few 44 KB slices of x (PageRank's site-local link structure, spartan/examples/pagerank.py): those slices are
 * staged through LDS instead of being gathered line by line from the L2.
 *   sp_csr_spmv_blockplan_bytes: size of the plan buffer, or 0 when no blocked plan exists for the shape (not fp32,
 *     fewer than 262 144 entries or 4 096 rows, more than 64 entries per row on average): keep sp_csr_spmm.
 *   sp_csr_spmv_blockplan: builds the plan (a re-ordered copy of the entries, 10 bytes each, plus a segment table
 *     per row block) -- once per tile.  The first int64 of the plan is 1 afterwards, or 0 if some row is not sorted
 *     by column or holds a column outside [0, k): then the plan must not be used.
 *   sp_csr_spmv_blocked: the product; d_x contiguous and 16-byte aligned, y with stride ldy. */
size_t sp_csr_spmv_blockplan_bytes(int32_t dtype, int64_t m, int64_t k, int64_t nnz);
int sp_csr_spmv_blockplan(int32_t dtype, int64_t m, int64_t k, int64_t nnz, const int64_t* d_indptr,
                          const int32_t* d_indices, const void* d_vals, void* d_plan, size_t plan_bytes, void* stream);
int sp_csr_spmv_blocked(int32_t dtype, int64_t m, int64_t k, int64_t nnz, const int64_t* d_indptr, const void* d_plan,
                        const void* d_x, void* d_y, int64_t ldy, int32_t accumulate, void* stream);
/* sp_csr_scatter: write a CSR tile into the box of a dense tile whose upper-left corner is (row0, col0).
 * mode 0: assign, 1: add, 2: sparse_to_dense_update with REDUCE_ADD (sparse.pyx:21-38; tile.pyx:229-233):
 * where mask == 0 assign and set the mask, else add. */
int sp_csr_scatter(int32_t dtype, int64_t m, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                   const void* d_vals, void* d_out, int64_t ld, int64_t row0, int64_t col0, uint8_t* d_mask,
                   int64_t ldmask, int32_t mode, void* stream);
/* sparse x sparse (scipy's csr_matmat behind tile_a.dot(tile_b), dot.py:216,237), as expand -> sort ->
 * compress: sp_spgemm_count gives every stored entry of A the offset of its products (d_offs[nnz_a + 1], int32)
 * and the exact number of products (*d_total, device int64); sp_spgemm_expand writes the (row, col, a*b) list
 * in A's storage order; sp_coo_to_csr adds the products of a cell in that (k ascending) order. */
size_t sp_spgemm_count_workspace_bytes(int64_t nnz_a);
int sp_spgemm_count(int64_t nnz_a, const int32_t* d_indices_a, const int64_t* d_indptr_b, int32_t* d_offs,
                    int64_t* d_total, void* d_ws, size_t ws_bytes, void* stream);
int sp_spgemm_expand(int32_t dtype, int64_t m_a, int64_t nnz_a, const int64_t* d_indptr_a,
                     const int32_t* d_indices_a, const void* d_vals_a, const int64_t* d_indptr_b,
                     const int32_t* d_indices_b, const void* d_vals_b, const int32_t* d_offs, int32_t* d_rows,
                     int32_t* d_cols, void* d_vals, void* stream);

/* sp_tiling_solve: the solver behind the auto-tiling pass (reference: the CPython-2 extension
 * spartan/expr/operator/tiling.cc called from AutomaticTiling.calc_tiling, optimize.py:937-975).  HOST code, no GPU
 * work.  Nodes 0..n_nodes-1 are (expression, tiling) alternatives; group g owns the nodes
 * group_nodes[group_ptr[g] .. group_ptr[g+1]) and exactly one of them is chosen; nodes in no group are always chosen;
 * an edge's cost (>= 0; here bytes over xGMI links) is paid when both its ends are chosen.  choice[g] receives the
 * index (inside its group) of the chosen alternative, *total the cost.  Exact for small problems, greedy + local
 * moves beyond that. */
int sp_tiling_solve(int32_t n_nodes, int64_t n_edges, const int32_t* edge_u, const int32_t* edge_v,
                    const double* edge_cost, int32_t n_groups, const int32_t* group_ptr, const int32_t* group_nodes,
                    int32_t* choice, double* total);

/* sp_gather_rows: dst[i, :] = src[idx[i], :] -- integer-array indexing `x[idx]`, the tile body of _int_index_mapper
 * (spartan/expr/operator/filter.py:50-75).  Rows of row_bytes bytes (a multiple of 4), source rows
 * src_row_stride_bytes apart; idx int64 on the device, negative values count from the end. */
int sp_gather_rows(const void* d_src, int64_t src_row_stride_bytes, int64_t n_src_rows, const int64_t* d_idx,
                   int64_t n_idx, int64_t row_bytes, void* d_dst, void* stream);

/* sp_stream_copy: STREAM-style float4 copy used by bench.py to measure the
 * achievable HBM bandwidth of the box ("measured HBM bandwidth"). */
int sp_stream_copy(void* d_dst, const void* d_src, size_t bytes, void* stream);
/* The same copy on at most `max_workgroups` workgroups (0 = the full grid): a transfer that occupies a bounded
 * share of the CUs, the way a collective's channels do -- bench.py's one-GPU emulation of a rank of the K-split
 * dot uses it in place of the RCCL kernels it cannot run on one device. */
int sp_stream_copy_wg(void* d_dst, const void* d_src, size_t bytes, int32_t max_workgroups, void* stream);
/* hipMemsetAsync on `stream` (zero-initialised tiles: `Tile._initialize`, spartan/array/tile.pyx:115-127). */
int sp_memset(void* d_dst, int32_t value, size_t bytes, void* stream);

/* ---- the tile store: HBM blobs owned by the library ------------------------------------------------------
 * What a worker's `_blobs: TileId -> Tile` dictionary stores (spartan/worker.py:70) and BlobCtx creates, reads,
 * updates and destroys (spartan/blob_ctx.py:103-254; worker.py:126-185).  A handle is an opaque non-zero uint64;
 * the blob is dense, row-major, on the device that was current at creation, and lives until sp_blob_destroy
 * (the reference's refcnt / destroy_all, worker.py:152-170).  Destroyed allocations are kept for re-use (tile
 * sizes repeat every iteration and hipFree synchronises the device); sp_blob_trim returns them to the driver.
 * Re-use is stream-ordered: keep to one compute stream per device, or synchronise before destroying a blob that
 * another stream still uses.  The entry points above take plain device pointers: sp_blob_info gives a blob's.
 *   sp_blob_h2d / sp_blob_d2h   `create(Tile.from_data(ndarray))` / `get(tile_id, subslice)` towards a host
 *                               array: the box [ul, lr) of the blob (NULL, NULL = all of it) <-> a contiguous
 *                               host buffer of the box's shape; asynchronous on `stream`.
 *   sp_blob_slice_copy          `get` + `update(reducer=None)` between two tiles of one worker: the fetch stitch
 *                               of spartan/array/distarray.py:355-365 on handles. */
int sp_blob_create(const int64_t* shape, int32_t ndim, int32_t dtype, uint64_t* handle);
int sp_blob_destroy(uint64_t handle);
int sp_blob_trim(void);
int sp_blob_info(uint64_t handle, void** d_ptr, int64_t* shape, int32_t* ndim, int32_t* dtype);
int sp_blob_stats(int64_t* live_blobs, int64_t* pooled_bytes);
int sp_blob_h2d(uint64_t handle, const void* host, const int64_t* ul, const int64_t* lr, void* stream);
/* sp_blob_h2d for small driver-side operands (up to 4 MiB, a contiguous byte range of the blob): the host buffer is
 * copied into a pinned staging slot of the library before the call returns (*host_consumed = 1: the caller may
 * reuse or free it at once and need not synchronise) and reaches the device by an asynchronous copy on `stream`.
 * Anything else is handed to sp_blob_h2d as it is (*host_consumed = 0: the buffer must stay valid until the stream
 * has run the copy).  The reference pickles such operands into every kernel request (spartan/expr/operator/dot.py:
 * 172-187, worker.py:232-263). */
int sp_blob_h2d_staged(uint64_t handle, const void* host, const int64_t* ul, const int64_t* lr, void* stream,
                       int32_t* host_consumed);
/* sp_blob_d2h + wait: the box is in `host` when the call returns.  Up to 4 MiB (a contiguous byte range of the
 * blob) it travels through a pinned staging slot -- `glom` of a reduction's result, distarray.py:294-367. */
int sp_blob_d2h_staged(uint64_t handle, void* host, const int64_t* ul, const int64_t* lr, void* stream);
int sp_blob_d2h(uint64_t handle, void* host, const int64_t* ul, const int64_t* lr, void* stream);
int sp_blob_slice_copy(uint64_t dst, const int64_t* dst_ul, uint64_t src, const int64_t* src_ul,
                       const int64_t* extent, void* stream);

/* ---- the data plane between workers: collectives over RCCL / xGMI ------------------------------------------
 * One process per GPU.  These replace the reference's tile traffic between workers -- `BlobCtx.update(tile_id,
 * region, data, reducer)` pushed to the owner and merged there, `BlobCtx.get(tile_id, subslice)` pulled from it
 * (spartan/blob_ctx.py:143-179, worker.py:172-230, carried by ZeroMQ in spartan/rpc/zeromq.py) -- for the regular
 * patterns of the tile path:
 *   sp_comm_reduce_scatter   every worker holds a partial of the WHOLE target, the target is tiled one equal
 *                            contiguous piece per worker: `update(np.add)` of sum(axis=0) / dot(tile_hint=(M/p,N))
 *                            (distarray.py:372-422).  d_src: world * recv_count elements, d_dst: recv_count.
 *   sp_comm_reduce           the same into a one-tile target (dot's default tile_hint, dot.py:277-278; scalars)
 *   sp_comm_all_reduce       replicated results (k-means counts and sums)
 *   sp_comm_all_gather       `glom` / fetch of a whole one-tile-per-worker array (distarray.py:294-367)
 *   sp_comm_bcast            replicated fetch of one tile; small driver-side operands
 *   sp_comm_all_to_all_blocks  `fetch` of slabs that lie in other workers' tiles (the A column slabs of dot's
 *                            map2 join, map.py:243-286) and irregular updates: any set of point-to-point
 *                            blocks (bytes), issued as ONE group.  Blocks between the same two ranks are matched
 *                            in list order.
 * Rendezvous: rank 0 calls sp_comm_unique_id and hands the SP_COMM_UID_BYTES token to every rank by whatever
 * channel the host has (the reference's workers register with the master over TCP, worker.py:98-124); every
 * rank then calls sp_comm_init with the device it computes on current.  reducer: enum sp_reducer (ADD MUL MAX
 * MIN; AND / OR for SP_BOOL).  All calls are asynchronous on `stream`; RCCL is bound at run time
 * (sp_comm_available() == 0 when the host has none): $SPARTAN_RCCL_LIB, then the librccl.so.1 installed beside the
 * HIP runtime this library is bound to, then the loader's search path.  A copy linked against ANOTHER HIP runtime
 * than this library's (a process can hold two) is refused with the reason in sp_last_error(): the pointers,
 * streams and events a collective is handed must belong to the runtime it runs on.  sp_comm_paths reports the
 * files in use (each buffer each_bytes long): the RCCL bound, the HIP runtime it calls, this library's own. */
int sp_comm_available(void);
int sp_comm_version(int* version);
int sp_comm_paths(char* rccl_path, char* rccl_runtime_path, char* own_runtime_path, size_t each_bytes);
int sp_comm_unique_id(void* uid, size_t uid_bytes);
int sp_comm_init(int32_t world, int32_t rank, const void* uid, void** comm);
int sp_comm_destroy(void* comm);
int sp_comm_abort(void* comm);
int sp_comm_async_error(void* comm);
int sp_comm_all_reduce(void* comm, const void* d_src, void* d_dst, int64_t count, int32_t dtype, int32_t reducer,
                       void* stream);
int sp_comm_reduce_scatter(void* comm, const void* d_src, void* d_dst, int64_t recv_count, int32_t dtype,
                           int32_t reducer, void* stream);
int sp_comm_reduce(void* comm, const void* d_src, void* d_dst, int64_t count, int32_t dtype, int32_t reducer,
                   int32_t root, void* stream);
int sp_comm_all_gather(void* comm, const void* d_src, void* d_dst, int64_t send_count, int32_t dtype, void* stream);
int sp_comm_bcast(void* comm, void* d_buf, int64_t count, int32_t dtype, int32_t root, void* stream);
int sp_comm_all_to_all_blocks(void* comm, int32_t n_sends, const int32_t* send_peers, const void* const* d_send,
                              const int64_t* send_bytes, int32_t n_recvs, const int32_t* recv_peers,
                              void* const* d_recv, const int64_t* recv_bytes, void* stream);

/* Streams for a host that has none of its own (compute + communication), and ordering between streams through
 * the events below (the reference's worker serialises tile mutation with a lock, worker.py:134,160,181; here it
 * is stream order). */
int sp_set_device(int32_t device);
int sp_get_device(int32_t* device);                 /* the calling thread's current device */
int sp_device_synchronize(void);
int sp_stream_create(void** stream);
int sp_stream_create_priority(void** stream, int32_t high_priority);   /* high: collectives ahead of queued GEMMs */
int sp_stream_destroy(void* stream);
int sp_stream_synchronize(void* stream);
int sp_stream_query(void* stream, int32_t* done);
int sp_stream_wait_event(void* stream, void* ev);

/* HIP-event timing of work already enqueued on `stream`; used by bench.py so
 * kernel durations are measured on the stream the kernels were launched on. */
int sp_event_create(void** ev);
int sp_event_destroy(void* ev);
int sp_event_record(void* ev, void* stream);
int sp_event_synchronize(void* ev);
int sp_event_query(void* ev, int32_t* done);
int sp_event_elapsed_ms(void* start, void* stop, float* ms);

/* Pinned host buffers and a device -> host copy that does NOT wait: a driver loop that needs a few bytes of a result
 * on the host one iteration LATER (the cluster counts of a k-means iteration: is any cluster empty?) enqueues the copy
 * on a side stream behind an event and reads the buffer after sp_event_synchronize, while the compute stream is
 * already working on the next iteration.  (The reference's driver waits for every glom: blob_ctx.get, worker.py
 * 172-179; this is what replaces the wait when the value is only a check.) */
int sp_pinned_alloc(size_t bytes, void** host);
int sp_pinned_free(void* host);
int sp_copy_d2h_async(void* pinned_host, const void* d_src, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARTAN_HIP_H_ */
