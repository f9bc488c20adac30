// Run-time specialisation of the evaluator kernels (host interface; see sp_jit.hip).
#pragma once
#include "sp_common.hpp"

// Is run-time specialisation available and enabled (SP_NO_JIT unset, libhiprtc loadable)?
int sp_jit_enabled();
// Minimum number of elements of the index space for which specialising pays (SP_JIT_MIN_ELEMS).
int64_t sp_jit_min_elems();
// Returns the hipFunction_t of `template_expr` (a kernel template-id in which the program
// type is spelled StaticProg<1000>) specialised for the instruction stream of `p`, compiling
// it on first use; NULL if unavailable (the caller then uses the interpreter kernels).
// `header`: "map_kernel.hpp" or "reduce_impl.hpp".
void* sp_jit_get(const char* header, const char* template_expr, const sp_program* p);
// hipModuleLaunchKernel wrapper; returns 0 on success.
int sp_jit_launch(void* fn, dim3 grid, dim3 block, void** args, hipStream_t st);
