// sp_argreduce: C-ABI entry for the single-pass arg-reduce kernels (see reduce_impl.hpp).
#include "reduce_impl.hpp"

extern "C" size_t sp_argreduce_workspace_bytes(int32_t cls, int64_t outer, int64_t axis_len, int64_t inner) {
  return sp_ws_bytes(cls, outer, axis_len, inner, true);
}

extern "C" int sp_argreduce(const sp_program* prog, const void* const* d_inputs, int32_t which,
                            int64_t outer, int64_t axis_len, int64_t inner, int64_t index_offset,
                            int64_t nan_index, int64_t* d_out_idx, void* d_out_val, void* d_ws,
                            size_t ws_bytes, void* stream) {
  if (sp_validate_program(prog)) return 1;
  if (which != 0 && which != 1) SP_FAIL("sp_argreduce: which must be 0 (max) or 1 (min)");
  if (!d_out_idx) SP_FAIL("sp_argreduce: d_out_idx is NULL");
  if (sp_check_space(prog, outer, axis_len, inner)) return 1;
  sp_inputs in;
  memset(&in, 0, sizeof(in));
  for (int j = 0; j < prog->n_inputs; ++j) {
    if (!d_inputs || !d_inputs[j]) SP_FAIL("sp_argreduce: input %d is NULL", j);
    in.p[j] = d_inputs[j];
  }
  RedOut ro;
  memset(&ro, 0, sizeof(ro));
  ro.out = d_out_val;
  ro.out_idx = d_out_idx;
  ro.index_offset = index_offset;
  ro.nan_index = nan_index;
  hipStream_t st = (hipStream_t)stream;
  const sp_program prepared = sp_prepare_program(prog);
  prog = &prepared;
  switch (prog->cls) {
    case SP_F32:
      return sp_reduce_launch<float, ArgAcc>(prog, in, d_inputs, which, outer, axis_len, inner, ro, d_ws,
                                             ws_bytes, st);
    case SP_F64:
      return sp_reduce_launch<double, ArgAcc>(prog, in, d_inputs, which, outer, axis_len, inner, ro, d_ws,
                                              ws_bytes, st);
    default:
      return sp_reduce_launch<int64_t, ArgAcc>(prog, in, d_inputs, which, outer, axis_len, inner, ro, d_ws,
                                               ws_bytes, st);
  }
}
