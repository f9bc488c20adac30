// Fused elementwise map: ONE coalesced launch per output tile.
// Replaces op.evaluate(LocalCtx) + tile.from_data in tile_mapper
// (reference spartan/expr/operator/map.py:48-88, local.py:115-127).
//
// HBM roofline kernel: algorithmic bytes = sum(sizeof(in_j) for dense inputs)
// + sizeof(out) per element; 16 B per lane per operand.  Three tiers run the same
// evaluator (sp_interp.hpp): prebuilt specialised kernels and run-time specialised
// ones (sp_jit.hip) cover the tile with a full grid, one vector per lane; the
// interpreter kernels use a grid capped at 8 workgroups per CU that strides.
#include <type_traits>

#include "map_kernel.hpp"
#include "sp_jit.hpp"


// Can the program be evaluated V elements at a time?  Groups of V consecutive output elements must
// not cross the end of the innermost dimension (broadcast / strided operands change address mode
// there); alignment is not required (element-aligned vector accesses, sp_interp.hpp).
template <int V>
static bool sp_can_vectorize(const sp_program* p, const void* const* in, const void* out) {
  (void)in;
  (void)out;
  if (V == 1) return true;
  return p->shape[p->ndim - 1] % V == 0;
}

int sp_validate_program(const sp_program* p) {
  if (!p) SP_FAIL("sp_program is NULL");
  if (p->cls != SP_F32 && p->cls != SP_F64 && p->cls != SP_I64)
    SP_FAIL("sp_program.cls=%d is not SP_F32/SP_F64/SP_I64", p->cls);
  if (p->n_inputs < 0 || p->n_inputs > SP_MAX_INPUTS) SP_FAIL("n_inputs=%d out of range", p->n_inputs);
  if (p->n_instr < 0 || p->n_instr > SP_MAX_INSTR) SP_FAIL("n_instr=%d out of range", p->n_instr);
  if (p->ndim < 1 || p->ndim > SP_MAX_DIMS) SP_FAIL("ndim=%d out of range", p->ndim);
  if (p->result_reg < 0 || p->result_reg >= SP_NREG) SP_FAIL("result_reg=%d out of range", p->result_reg);
  if (p->n_inputs > SP_NREG) SP_FAIL("n_inputs=%d exceeds register file", p->n_inputs);
  // registers 0 .. n_inputs-1 hold the operands; every other register must be written before it is read (the
  // interpreter does not clear its register file: 32 moves per evaluation that no lowered program needs)
  unsigned defined = (1u << p->n_inputs) - 1u;
  for (int i = 0; i < p->n_instr; ++i) {
    const sp_instr& I = p->instr[i];
    if (I.dst >= SP_NREG || I.c >= SP_NREG) SP_FAIL("instr %d: register out of range", i);
    if (I.op == SP_OP_CONST) {
      if (I.a >= SP_MAX_CONSTS) SP_FAIL("instr %d: const index out of range", i);
    } else if (I.op >= SP_OP_ADDC && I.op <= SP_OP_MINC) {      // reg[b] (op) consts[a]
      if (I.a >= SP_MAX_CONSTS) SP_FAIL("instr %d: const index out of range", i);
      if (I.b >= SP_NREG) SP_FAIL("instr %d: register out of range", i);
      if ((1u << I.b) & ~defined) SP_FAIL("instr %d: reads a register nothing has written", i);
    } else if (I.a >= SP_NREG || I.b >= SP_NREG) {
      SP_FAIL("instr %d: register out of range", i);
    } else if (I.op != SP_OP_NOP && I.op != SP_OP_IOTA) {
      const bool binary = I.op >= SP_OP_ADD && I.op <= SP_OP_LXOR;
      unsigned reads = 1u << I.a;
      if (binary || I.op == SP_OP_WHERE) reads |= 1u << I.b;
      if (I.op == SP_OP_WHERE) reads |= 1u << I.c;
      if (reads & ~defined) SP_FAIL("instr %d: reads a register nothing has written", i);
    }
    if (I.op != SP_OP_NOP) defined |= 1u << I.dst;
  }
  if (!(defined & (1u << p->result_reg))) SP_FAIL("result_reg=%d is never written", p->result_reg);
  for (int j = 0; j < p->n_inputs; ++j)
    if (p->in_dtype[j] < 0 || p->in_dtype[j] >= SP_DTYPE_COUNT) SP_FAIL("input %d: bad dtype", j);
  for (int d = 0; d < p->ndim; ++d)
    if (p->shape[d] < 0) SP_FAIL("negative shape");
  return 0;
}

static inline unsigned sp_grid_for(int64_t nvec, int U) {
  int64_t blocks = (nvec + (int64_t)SP_BLOCK * U - 1) / ((int64_t)SP_BLOCK * U);
  // interpreter kernels pay a per-workgroup prologue (program + strides from the kernel
  // argument segment): a capped grid that strides measured faster than a full grid
  // (round 4, 2 GiB tile, 5-op chain: 8 / 16 / 32 workgroups per CU 1.23 / 1.18 / 1.16 ms, the whole tile 1.34;
  // round 5, after the trip lost its hoisted index conversions: 16 / 32 / 64 / 128 / 256 per CU and the whole tile
  // 1.00 / 0.97 / 0.93 / 0.98 / 1.01 / 1.20 ms, `x + 1` 0.85 / 0.84 / 0.77 / 0.77 / 0.74 / 0.90)
  static int per_cu = -1;
  if (per_cu < 0) {
    const char* e = getenv("SP_INTERP_WG_PER_CU");
    per_cu = e ? atoi(e) : 8 * SP_BLOCKS_PER_CU;
  }
  const int64_t cap = (int64_t)SP_CUS * per_cu;
  if (per_cu > 0 && blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// SP_NO_STATIC=1 forces the generic interpreter (A/B measurements, tests).
int sp_static_enabled() {
  static int v = -1;
  if (v < 0) v = getenv("SP_NO_STATIC") ? 0 : 1;
  return v;
}

// Reports which specialised kernel (if any) a program would run on; -1 = interpreter.
extern "C" int sp_program_static_id(const sp_program* prog, int32_t out_dtype) {
  return prog ? sp_find_static(prog, out_dtype) : -1;
}

// Tuning knob: groups per lane for the vectorised fp32 paths (SP_MAP_UNROLL=1|2|4).
static int sp_map_unroll() {
  static int u = -1;
  if (u < 0) {
    const char* e = getenv("SP_MAP_UNROLL");
    // two groups per lane: the dispatch (fetch, decode, branch) is shared by both; four need > 256 VGPRs
    // (profiles/r04_notes.md: 5-op chain on the 2 GiB tile 1.21 / 1.11 / 2.2 ms at 1 / 2 / 4)
    u = e ? atoi(e) : 2;
    if (u != 1 && u != 2 && u != 4) u = 1;
  }
  return u;
}

template <typename T, int V, int U, bool LINEAR>
static int sp_map_go(const sp_program* p, const sp_inputs& in, void* out, int64_t start, int64_t nvec,
                     hipStream_t st) {
  hipLaunchKernelGGL((sp_map_kernel<T, V, U, LINEAR>), dim3(sp_grid_for(nvec, U)), dim3(SP_BLOCK), 0, st, *p, in,
                     out, start, nvec);
  SP_CHECK_LAUNCH();
  return 0;
}

// Specialised (compile-time instruction stream) fp32 kernels: full grid, one
// 16-B vector per lane -- the structure that reaches the copy bandwidth
// (tools/hbm_probe.hip: 6.2 TB/s vs 4.7 TB/s for a capped grid-stride loop).
template <bool LINEAR>
static int sp_map_go_static(int sid, const sp_program* p, const sp_inputs& in, void* out, int64_t nvec,
                            hipStream_t st) {
  int64_t blocks = (nvec + SP_BLOCK - 1) / SP_BLOCK;
  if (blocks > (1LL << 30)) blocks = 1LL << 30;
  switch (sid) {
#define SP_CASE(ID)                                                                                   \
  case ID:                                                                                            \
    if (p->pad & SP_PAD_STREAM)                                                                       \
      hipLaunchKernelGGL((sp_map_kernel<float, 4, 1, LINEAR, StaticProg<ID>, -1, false, 1>),          \
                         dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st, *p, in, out, (int64_t)0, nvec); \
    else                                                                                              \
      hipLaunchKernelGGL((sp_map_kernel<float, 4, 1, LINEAR, StaticProg<ID>, -1, false, 0>),          \
                         dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st, *p, in, out, (int64_t)0, nvec); \
    break;
    SP_FOR_EACH_STATIC(SP_CASE)
#undef SP_CASE
    default: SP_FAIL("internal: unknown static program %d", sid);
  }
  SP_CHECK_LAUNCH();
  return 0;
}

// Specialised programs with NumPy-broadcast operands (x - row_means, x * col_scale,
// x * (yp - y) ...): addressing mode per operand fixed at compile time (MASK).
static int sp_map_go_static_2d(int sid, int mask, const sp_program* p, const sp_inputs& in, void* out,
                               int64_t nvec, hipStream_t st, bool* handled) {
  int64_t blocks = (nvec + SP_BLOCK - 1) / SP_BLOCK;
  if (blocks > (1LL << 30)) blocks = 1LL << 30;
  *handled = true;
#define SP_CASE2(ID, MSK)                                                                                  \
  if (sid == ID && mask == MSK) {                                                                          \
    if (p->pad & SP_PAD_STREAM)                                                                            \
      hipLaunchKernelGGL((sp_map_kernel<float, 4, 1, false, StaticProg<ID>, MSK, false, 1>),               \
                         dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st, *p, in, out, (int64_t)0, nvec);    \
    else                                                                                                   \
      hipLaunchKernelGGL((sp_map_kernel<float, 4, 1, false, StaticProg<ID>, MSK, false, 0>),               \
                         dim3((unsigned)blocks), dim3(SP_BLOCK), 0, st, *p, in, out, (int64_t)0, nvec);    \
    SP_CHECK_LAUNCH();                                                                                     \
    return 0;                                                                                              \
  }
  // binary ops: second operand broadcast along columns (mask 2) or along rows (mask 0, strides (0,1));
  // first operand broadcast (mask 1)
  SP_CASE2(5, 0) SP_CASE2(5, 1) SP_CASE2(5, 2) SP_CASE2(6, 0) SP_CASE2(6, 1) SP_CASE2(6, 2)
  SP_CASE2(7, 0) SP_CASE2(7, 1) SP_CASE2(7, 2) SP_CASE2(8, 0) SP_CASE2(8, 1) SP_CASE2(8, 2)
  SP_CASE2(10, 6) SP_CASE2(10, 0)
#undef SP_CASE2
  *handled = false;
  return 0;
}

// Run-time specialised kernel for a program outside the prebuilt library (large tiles only).
template <typename T, int V, bool LINEAR>
static int sp_map_go_jit(const sp_program* p, const sp_inputs& in, void* out, int64_t nvec, int mask,
                         hipStream_t st, bool* handled, bool ragged = false) {
  *handled = false;
  if (!sp_jit_enabled() || p->n_instr == 0 || nvec * V < sp_jit_min_elems()) return 0;
  char expr[176];
  snprintf(expr, sizeof(expr), "sp_map_kernel<%s, %d, 1, %s, StaticProg<1000>, %d, %s, %d>", sp_cls<T>::name(), V,
           LINEAR ? "true" : "false", mask, ragged ? "true" : "false", (p->pad & SP_PAD_STREAM) ? 1 : 0);
  void* fn = sp_jit_get("map_kernel.hpp", expr, p);
  if (!fn) return 0;
  int64_t blocks = (nvec + SP_BLOCK - 1) / SP_BLOCK;
  if (blocks > (1LL << 30)) blocks = 1LL << 30;
  sp_program pc = *p;
  sp_inputs ic = in;
  int64_t start = 0;
  void* args[] = {&pc, &ic, &out, &start, &nvec};
  if (sp_jit_launch(fn, dim3((unsigned)blocks), dim3(SP_BLOCK), args, st)) return 1;
  *handled = true;
  return 0;
}

template <typename T, int V, bool LINEAR>
static int sp_map_go_u(const sp_program* p, const sp_inputs& in, void* out, int64_t start, int64_t nvec,
                       hipStream_t st) {
  // multi-group variants are instantiated for the fp32 class only (the hot dtype)
  if constexpr (std::is_same<T, float>::value && V > 1) {
    const int u = nvec >= 4 * SP_BLOCK * 64 ? sp_map_unroll() : 1;
    if (u == 4) return sp_map_go<T, V, 4, LINEAR>(p, in, out, start, nvec, st);
    if (u == 2) return sp_map_go<T, V, 2, LINEAR>(p, in, out, start, nvec, st);
  }
  return sp_map_go<T, V, 1, LINEAR>(p, in, out, start, nvec, st);
}

template <typename T>
static int sp_map_launch(const sp_program* p, const sp_inputs& in, const void* const* inp, void* out,
                         hipStream_t st) {
  constexpr int V = sp_cls<T>::V;
  int64_t n = 1;
  for (int d = 0; d < p->ndim; ++d) n *= p->shape[d];
  if (n == 0) return 0;
  const bool lin = p->linear != 0;
  if (lin) {
    // dense: vector body + scalar tail (element-aligned vector accesses: any base address will do)
    (void)inp;
    int64_t nmain = (n / V) * V;
    if constexpr (std::is_same<T, float>::value) {
      const int sid = sp_static_enabled() ? sp_find_static(p, p->out_dtype) : -1;
      if (nmain && sid >= 0) {
        if (sp_map_go_static<true>(sid, p, in, out, nmain / V, st)) return 1;
        if (n - nmain && sp_map_go<T, 1, 1, true>(p, in, out, nmain, n - nmain, st)) return 1;
        return 0;
      }
    }
    if (nmain) {
      bool handled = false;
      if (sp_map_go_jit<T, V, true>(p, in, out, nmain / V, -1, st, &handled)) return 1;
      if (handled) {
        if (n - nmain && sp_map_go<T, 1, 1, true>(p, in, out, nmain, n - nmain, st)) return 1;
        return 0;
      }
    }
    if (nmain && sp_map_go_u<T, V, true>(p, in, out, 0, nmain / V, st)) return 1;
    if (n - nmain && sp_map_go<T, 1, 1, true>(p, in, out, nmain, n - nmain, st)) return 1;
    return 0;
  }
  if (sp_can_vectorize<V>(p, inp, out)) {
    if constexpr (std::is_same<T, float>::value) {
      const int sid = sp_static_enabled() ? sp_find_static(p, p->out_dtype) : -1;
      if (sid >= 0) {
        const int mask = sp_mask_2d(p, p->n_inputs);
        if (mask >= 0) {
          bool handled = false;
          if (sp_map_go_static_2d(sid, mask, p, in, out, n / V, st, &handled)) return 1;
          if (handled) return 0;
        }
        return sp_map_go_static<false>(sid, p, in, out, n / V, st);
      }
    }
    {
      // outside the library: run-time specialisation, with compile-time 2-D addressing when it applies
      bool handled = false;
      const int mask = sp_mask_2d(p, p->n_inputs);
      if (sp_map_go_jit<T, V, false>(p, in, out, n / V, mask, st, &handled)) return 1;
      if (handled) return 0;
    }
    return sp_map_go_u<T, V, false>(p, in, out, 0, n / V, st);
  }
  if constexpr (V > 1) {
    // innermost dimension not a multiple of V: rows x ceil(inner / V) groups, scalar last group
    const int64_t inner = p->shape[p->ndim - 1];
    const int64_t nvec = (n / inner) * ((inner + V - 1) / V);
    bool handled = false;
    const int mask = sp_mask_2d(p, p->n_inputs);
    if (sp_map_go_jit<T, V, false>(p, in, out, nvec, mask, st, &handled, true)) return 1;
    if (handled) return 0;
    hipLaunchKernelGGL((sp_map_kernel<T, V, 1, false, DynProg, -1, true>), dim3(sp_grid_for(nvec, 1)), dim3(SP_BLOCK), 0,
                       st, *p, in, out, (int64_t)0, nvec);
    SP_CHECK_LAUNCH();
    return 0;
  }
  return sp_map_go<T, 1, 1, false>(p, in, out, 0, n, st);
}

extern "C" int sp_map_fused(const sp_program* prog, const void* const* d_inputs, void* d_out,
                            void* stream) {
  if (sp_validate_program(prog)) return 1;
  if (!d_out) SP_FAIL("sp_map_fused: d_out is NULL");
  if (prog->out_dtype < 0 || prog->out_dtype >= SP_DTYPE_COUNT) SP_FAIL("bad out_dtype");
  sp_inputs in;
  memset(&in, 0, sizeof(in));
  for (int j = 0; j < prog->n_inputs; ++j) {
    if (!d_inputs || !d_inputs[j]) SP_FAIL("sp_map_fused: input %d is NULL", j);
    in.p[j] = d_inputs[j];
  }
  hipStream_t st = (hipStream_t)stream;
  const sp_program prepared = sp_prepare_program(prog);
  prog = &prepared;
  switch (prog->cls) {
    case SP_F32: return sp_map_launch<float>(prog, in, d_inputs, d_out, st);
    case SP_F64: return sp_map_launch<double>(prog, in, d_inputs, d_out, st);
    default: return sp_map_launch<int64_t>(prog, in, d_inputs, d_out, st);
  }
}
