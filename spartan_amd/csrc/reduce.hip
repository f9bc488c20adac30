// sp_reduce: C-ABI entry for the fused map->reduce kernels (see reduce_impl.hpp).
#include "reduce_impl.hpp"

static_assert(sizeof(RedOut) == 56, "sp_jit_preload (sp_jit.hip) warms the reduce kernels with a zeroed RedOut of this size");

extern "C" size_t sp_reduce_workspace_bytes(int32_t cls, int64_t outer, int64_t axis_len, int64_t inner) {
  return sp_ws_bytes(cls, outer, axis_len, inner, false);
}
extern "C" int sp_reduce(const sp_program* prog, const void* const* d_inputs, int32_t red_op,
                         int64_t outer, int64_t axis_len, int64_t inner, void* d_out, int32_t out_dtype,
                         void* d_ws, size_t ws_bytes, void* stream) {
  if (sp_validate_program(prog)) return 1;
  if (red_op < SP_RED_SUM || red_op > SP_RED_OR) SP_FAIL("sp_reduce: bad red_op %d", red_op);
  if (!d_out) SP_FAIL("sp_reduce: d_out is NULL");
  if (out_dtype < 0 || out_dtype >= SP_DTYPE_COUNT) SP_FAIL("sp_reduce: bad out_dtype");
  if (sp_check_space(prog, outer, axis_len, inner)) return 1;
  sp_inputs in;
  memset(&in, 0, sizeof(in));
  for (int j = 0; j < prog->n_inputs; ++j) {
    if (!d_inputs || !d_inputs[j]) SP_FAIL("sp_reduce: input %d is NULL", j);
    in.p[j] = d_inputs[j];
  }
  RedOut ro;
  memset(&ro, 0, sizeof(ro));
  ro.out = d_out;
  ro.out_dtype = out_dtype;
  hipStream_t st = (hipStream_t)stream;
  const sp_program prepared = sp_prepare_program(prog);
  prog = &prepared;
  switch (prog->cls) {
    case SP_F32:
      return sp_reduce_launch<float, PlainAcc>(prog, in, d_inputs, red_op, outer, axis_len, inner, ro, d_ws,
                                               ws_bytes, st);
    case SP_F64:
      return sp_reduce_launch<double, PlainAcc>(prog, in, d_inputs, red_op, outer, axis_len, inner, ro, d_ws,
                                                ws_bytes, st);
    default:
      return sp_reduce_launch<int64_t, PlainAcc>(prog, in, d_inputs, red_op, outer, axis_len, inner, ro,
                                                 d_ws, ws_bytes, st);
  }
}

