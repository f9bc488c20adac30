/* C-ABI of libspartan_hip_extras.so: kernels behind operators that are NOT on the tile path this repository is
 * about (SURVEY.md section 2 marks them out of scope) but that the host framework still offers.  Built by
 * `make extras` in spartan_amd/csrc (and by __graft_entry__.build()), not by the default `make`; the library links
 * against libspartan_hip.so (error reporting: sp_last_error()). */
#ifndef SPARTAN_HIP_EXTRAS_H_
#define SPARTAN_HIP_EXTRAS_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* sp_sort_rows: np.sort / np.argsort along the LAST axis of a contiguous [rows, cols] tile -- the tile bodies of the
 * sort operators (spartan/expr/operator/sort.py:68-69 _sort_mapper, :137-138 _argsort_mapper, :24 / :65 the
 * flat np.sort of the sample sort; rows == 1 sorts a flattened tile).  Stable (np.argsort(kind='stable')), NaN
 * last, -0.0 == +0.0; d_out_vals (sorted values, may be NULL) and d_out_idx (int64 column of each sorted value,
 * may be NULL) are out of place.  dtype: SP_F32 | SP_F64 | SP_I32 | SP_I64.  Rows of <= 4096 32-bit elements are
 * sorted in LDS in one pass over HBM; anything else by an LSD radix sort of the whole tile (key bytes, then the
 * row of each position). */
size_t sp_sort_rows_workspace_bytes(int32_t dtype, int64_t rows, int64_t cols);
int sp_sort_rows(const void* d_in, int32_t dtype, int64_t rows, int64_t cols, void* d_out_vals, int64_t* d_out_idx,
                 void* d_ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARTAN_HIP_EXTRAS_H_ */
