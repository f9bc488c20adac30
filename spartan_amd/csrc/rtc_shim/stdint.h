// hipRTC has no C library headers; these shims (used ONLY by sp_jit.hip's run-time
// compiles, via -I) map the few names the evaluator headers need onto hipRTC's built-ins.
#pragma once
typedef __hip_internal::int8_t int8_t;
typedef __hip_internal::uint8_t uint8_t;
typedef __hip_internal::int16_t int16_t;
typedef __hip_internal::uint16_t uint16_t;
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::int64_t int64_t;
typedef __hip_internal::uint64_t uint64_t;
#ifndef INT64_MAX
#define INT64_MAX 9223372036854775807LL
#define INT64_MIN (-INT64_MAX - 1)
#endif
#ifndef INT32_MAX
#define INT32_MAX 2147483647
#define INT32_MIN (-INT32_MAX - 1)
#endif
