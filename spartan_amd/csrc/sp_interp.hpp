// Device-side evaluator for a fused LocalExpr program (see sp_program in
// include/spartan_hip.h).  The program is wave-uniform (it lives in the kernel
// argument segment and is read with scalar loads), so the per-instruction
// dispatch is scalar control flow and the virtual register file r[SP_NREG][V]
// stays in VGPRs, addressed with s_set_gpr_idx (checked in the ISA: no scratch).
//
// One thread evaluates V consecutive elements of the row-major output index
// space per call; V*sizeof(T) == 16 B so the dense operands are read with one
// global_load_dwordx4 per lane (1 KiB per wave-instruction).
#pragma once

#include "sp_common.hpp"

// the two type traits the evaluator needs, spelled here so that these headers also
// compile under hipRTC (no C++ standard library there; see sp_jit.hip)
template <typename A, typename B> struct sp_is_same { static constexpr bool value = false; };
template <typename A> struct sp_is_same<A, A> { static constexpr bool value = true; };
template <typename T> struct sp_is_integral { static constexpr bool value = false; };
template <> struct sp_is_integral<int64_t> { static constexpr bool value = true; };
template <> struct sp_is_integral<int32_t> { static constexpr bool value = true; };
template <> struct sp_is_integral<uint8_t> { static constexpr bool value = true; };

template <typename T>
struct sp_cls;
template <>
struct sp_cls<float> {
  static constexpr const char* name() { return "float"; }
  static constexpr int V = 4;
  static constexpr int id = SP_F32;
};
template <>
struct sp_cls<double> {
  static constexpr const char* name() { return "double"; }
  static constexpr int V = 2;
  static constexpr int id = SP_F64;
};
template <>
struct sp_cls<int64_t> {
  static constexpr const char* name() { return "int64_t"; }
  static constexpr int V = 2;
  static constexpr int id = SP_I64;
};

// Vector memory types with ELEMENT alignment: global_load/store_dwordx2/x4 do not need 16-B aligned
// addresses on gfx950 (the compiler still emits one wide instruction), so rows of odd length and
// views that start anywhere keep the 16-B-per-lane access pattern.
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef double sp_f64x2 __attribute__((ext_vector_type(2)));
typedef int32_t sp_i32x4 __attribute__((ext_vector_type(4)));
typedef int32_t sp_i32x2 __attribute__((ext_vector_type(2)));
typedef int64_t sp_i64x2 __attribute__((ext_vector_type(2)));
typedef uint8_t sp_u8x4 __attribute__((ext_vector_type(4)));
typedef uint8_t sp_u8x2 __attribute__((ext_vector_type(2)));
typedef sp_f32x4 sp_f32x4u __attribute__((aligned(4)));
typedef sp_f32x2 sp_f32x2u __attribute__((aligned(4)));
typedef sp_f64x2 sp_f64x2u __attribute__((aligned(8)));
typedef sp_i32x4 sp_i32x4u __attribute__((aligned(4)));
typedef sp_i32x2 sp_i32x2u __attribute__((aligned(4)));
typedef sp_i64x2 sp_i64x2u __attribute__((aligned(8)));
typedef sp_u8x4 sp_u8x4u __attribute__((aligned(1)));
typedef sp_u8x2 sp_u8x2u __attribute__((aligned(1)));

constexpr int32_t SP_PAD_IDX32 = 1, SP_PAD_STREAM = 2;
// How a kernel learns that its launch streams (template parameter NTM of the evaluators): decided when the kernel
// is chosen (0: no, 1: yes -- the specialised map kernels, whose two forms are separate instantiations) or read
// from the program's flags at run time (2: reductions, interpreter).
#define SP_STREAMS(NTM, p) ((NTM) == 1 || ((NTM) == 2 && ((p).pad & SP_PAD_STREAM) != 0))
constexpr int64_t SP_STREAM_ELEMS = 8LL << 20;   // 32 MiB of fp32 = the aggregate L2

// ---- typed loads: `n` consecutive elements (n == V, or 1) converted to T ----
// NT: the operand is read exactly once by the whole launch (no broadcast dimension), so its lines need not stay in
// L2: non-temporal vector loads (`global_load_dwordx4 ... nt`).  On the 2 GiB tile this is worth 6.0 -> 6.6-6.8 TB/s
// for the reductions and 6.25 -> 6.45 TB/s for the maps; a re-read operand (a row vector broadcast down the rows)
// loses 3 % with it, hence the flag.
// (a macro, not a function template: deducing the vector type would drop the reduced alignment of the `...u` types)
// (the empty asm statements on both sides keep LLVM from merging the non-temporal load with its plain twin in the
// other arm of a run-time `if (stream)` -- hoisted or sunk, a merged load keeps only the metadata both have, i.e.
// loses the hint)
__device__ __forceinline__ void sp_nt_mark() { asm volatile(""); }
#define SP_VLD(TYPE, ptr)                                                  \
  (NT ? ({                                                                 \
    sp_nt_mark();                                                          \
    const TYPE nt_ = __builtin_nontemporal_load((const TYPE*)(ptr));       \
    sp_nt_mark();                                                          \
    nt_;                                                                   \
  })                                                                       \
      : *(const TYPE*)(ptr))

template <typename T, int N, bool NT = false>
__device__ __forceinline__ void sp_load_vec(const void* base, int32_t dt, int64_t off, T* dst) {
  switch (dt) {
    case SP_F32: {
      const float* p = (const float*)base + off;
      if constexpr (N == 4) {
        sp_f32x4u v = SP_VLD(sp_f32x4u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y; dst[2] = (T)v.z; dst[3] = (T)v.w;
      } else if constexpr (N == 2) {
        sp_f32x2u v = SP_VLD(sp_f32x2u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y;
      } else {
        dst[0] = (T)p[0];
      }
    } break;
    case SP_F64: {
      const double* p = (const double*)base + off;
      if constexpr (N == 4) {
        sp_f64x2u v0 = SP_VLD(sp_f64x2u, p), v1 = SP_VLD(sp_f64x2u, (p + 2));
        dst[0] = (T)v0.x; dst[1] = (T)v0.y; dst[2] = (T)v1.x; dst[3] = (T)v1.y;
      } else if constexpr (N == 2) {
        sp_f64x2u v = SP_VLD(sp_f64x2u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y;
      } else {
        dst[0] = (T)p[0];
      }
    } break;
    case SP_I32: {
      const int32_t* p = (const int32_t*)base + off;
      if constexpr (N == 4) {
        sp_i32x4u v = SP_VLD(sp_i32x4u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y; dst[2] = (T)v.z; dst[3] = (T)v.w;
      } else if constexpr (N == 2) {
        sp_i32x2u v = SP_VLD(sp_i32x2u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y;
      } else {
        dst[0] = (T)p[0];
      }
    } break;
    case SP_I64: {
      const int64_t* p = (const int64_t*)base + off;
      if constexpr (N == 4) {
        sp_i64x2u v0 = SP_VLD(sp_i64x2u, p), v1 = SP_VLD(sp_i64x2u, (p + 2));
        dst[0] = (T)v0.x; dst[1] = (T)v0.y; dst[2] = (T)v1.x; dst[3] = (T)v1.y;
      } else if constexpr (N == 2) {
        sp_i64x2u v = SP_VLD(sp_i64x2u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y;
      } else {
        dst[0] = (T)p[0];
      }
    } break;
    default: {  // SP_BOOL / SP_U8
      const uint8_t* p = (const uint8_t*)base + off;
      if constexpr (N == 4) {
        sp_u8x4u v = SP_VLD(sp_u8x4u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y; dst[2] = (T)v.z; dst[3] = (T)v.w;
      } else if constexpr (N == 2) {
        sp_u8x2u v = SP_VLD(sp_u8x2u, p);
        dst[0] = (T)v.x; dst[1] = (T)v.y;
      } else {
        dst[0] = (T)p[0];
      }
    } break;
  }
}

template <typename T>
__device__ __forceinline__ int64_t sp_to_i64(T x) {
  return (int64_t)x;
}

// ---- typed stores with the NumPy cast semantics of ndarray.astype ----
// NT: the output of a launch bigger than the L2 (non-temporal 16-B stores: 6.42 -> 6.51 TB/s on the 2 GiB map)
#define SP_VST(TYPE, ptr, ...)                                        \
  do {                                                                \
    const TYPE v_ = __VA_ARGS__;                                      \
    if constexpr (NT) { sp_nt_mark(); __builtin_nontemporal_store(v_, (TYPE*)(ptr)); sp_nt_mark(); } \
    else *(TYPE*)(ptr) = v_;                                          \
  } while (0)
template <typename T, int N, bool NT = false>
__device__ __forceinline__ void sp_store_vec(void* base, int32_t dt, int64_t off, const T* src) {
  switch (dt) {
    case SP_F32: {
      float* p = (float*)base + off;
      if constexpr (N == 4) {
        SP_VST(sp_f32x4u, p, {(float)src[0], (float)src[1], (float)src[2], (float)src[3]});
      } else if constexpr (N == 2) {
        *(sp_f32x2u*)p = sp_f32x2{(float)src[0], (float)src[1]};
      } else {
        p[0] = (float)src[0];
      }
    } break;
    case SP_F64: {
      double* p = (double*)base + off;
#pragma unroll
      for (int j = 0; j < N; j += 2) {
        if constexpr (N >= 2) {
          SP_VST(sp_f64x2u, p + j, {(double)src[j], (double)src[j + 1]});
        } else {
          p[0] = (double)src[0];
        }
      }
    } break;
    case SP_I32: {
      int32_t* p = (int32_t*)base + off;
      if constexpr (N == 4) {
        SP_VST(sp_i32x4u, p, {(int32_t)src[0], (int32_t)src[1], (int32_t)src[2], (int32_t)src[3]});
      } else if constexpr (N == 2) {
        *(sp_i32x2u*)p = sp_i32x2{(int32_t)src[0], (int32_t)src[1]};
      } else {
        p[0] = (int32_t)src[0];
      }
    } break;
    case SP_I64: {
      int64_t* p = (int64_t*)base + off;
#pragma unroll
      for (int j = 0; j < N; j += 2) {
        if constexpr (N >= 2) {
          SP_VST(sp_i64x2u, p + j, {(int64_t)src[j], (int64_t)src[j + 1]});
        } else {
          p[0] = (int64_t)src[0];
        }
      }
    } break;
    case SP_BOOL: {
      uint8_t* p = (uint8_t*)base + off;
      if constexpr (N == 4) {
        *(sp_u8x4u*)p = sp_u8x4{(uint8_t)(src[0] != (T)0), (uint8_t)(src[1] != (T)0), (uint8_t)(src[2] != (T)0), (uint8_t)(src[3] != (T)0)};
      } else if constexpr (N == 2) {
        *(sp_u8x2u*)p = sp_u8x2{(uint8_t)(src[0] != (T)0), (uint8_t)(src[1] != (T)0)};
      } else {
        p[0] = src[0] != (T)0;
      }
    } break;
    default: {  // SP_U8
      uint8_t* p = (uint8_t*)base + off;
#pragma unroll
      for (int j = 0; j < N; ++j) p[j] = (uint8_t)(int64_t)src[j];
    } break;
  }
}

// ---- scalar op semantics (NumPy ufunc semantics; see header enum) ----
template <typename T>
struct sp_math;

template <>
struct sp_math<float> {
  using T = float;
  static __device__ __forceinline__ T div(T a, T b) { return a / b; }
  static __device__ __forceinline__ T fmod_(T a, T b) { return fmodf(a, b); }
  static __device__ __forceinline__ T mod(T a, T b) {
    if (b == 0.0f) return fmodf(a, b);
    T r = fmodf(a, b);
    if (r != 0.0f) {
      if ((b < 0.0f) != (r < 0.0f)) r += b;
    } else {
      r = copysignf(0.0f, b);
    }
    return r;
  }
  static __device__ __forceinline__ T floordiv(T a, T b) {
    // npy_divmod (numpy/core/src/npymath): floor division consistent with mod
    if (b == 0.0f) return a / b;
    T m = fmodf(a, b);
    T d = (a - m) / b;
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) d -= 1.0f;
    if (d != 0.0f) {
      T f = floorf(d);
      if (d - f > 0.5f) f += 1.0f;
      return f;
    }
    return copysignf(0.0f, a / b);
  }
  static __device__ __forceinline__ T pow_(T a, T b) { return powf(a, b); }
  static __device__ __forceinline__ T sqrt_(T a) { return sqrtf(a); }
  static __device__ __forceinline__ T exp_(T a) { return expf(a); }
  static __device__ __forceinline__ T log_(T a) { return logf(a); }
  static __device__ __forceinline__ T tanh_(T a) { return tanhf(a); }
  static __device__ __forceinline__ T normcdf_(T a) { return 0.5f * erfcf(-a * 0.70710678118654752440f); }
  static __device__ __forceinline__ T floor_(T a) { return floorf(a); }
  static __device__ __forceinline__ T ceil_(T a) { return ceilf(a); }
  static __device__ __forceinline__ T abs_(T a) { return fabsf(a); }
  static __device__ __forceinline__ bool isnan_(T a) { return a != a; }
  static __device__ __forceinline__ T to_f32(T a) { return a; }
  static __device__ __forceinline__ T to_i32(T a) { return (T)(int32_t)a; }
  static __device__ __forceinline__ T to_i64(T a) { return (T)(int64_t)a; }
};

template <>
struct sp_math<double> {
  using T = double;
  static __device__ __forceinline__ T div(T a, T b) { return a / b; }
  static __device__ __forceinline__ T fmod_(T a, T b) { return fmod(a, b); }
  static __device__ __forceinline__ T mod(T a, T b) {
    if (b == 0.0) return fmod(a, b);
    T r = fmod(a, b);
    if (r != 0.0) {
      if ((b < 0.0) != (r < 0.0)) r += b;
    } else {
      r = copysign(0.0, b);
    }
    return r;
  }
  static __device__ __forceinline__ T floordiv(T a, T b) {
    if (b == 0.0) return a / b;
    T m = fmod(a, b);
    T d = (a - m) / b;
    if (m != 0.0 && ((b < 0.0) != (m < 0.0))) d -= 1.0;
    if (d != 0.0) {
      T f = floor(d);
      if (d - f > 0.5) f += 1.0;
      return f;
    }
    return copysign(0.0, a / b);
  }
  static __device__ __forceinline__ T pow_(T a, T b) { return pow(a, b); }
  static __device__ __forceinline__ T sqrt_(T a) { return sqrt(a); }
  static __device__ __forceinline__ T exp_(T a) { return exp(a); }
  static __device__ __forceinline__ T log_(T a) { return log(a); }
  static __device__ __forceinline__ T tanh_(T a) { return tanh(a); }
  static __device__ __forceinline__ T normcdf_(T a) { return 0.5 * erfc(-a * 0.70710678118654752440); }
  static __device__ __forceinline__ T floor_(T a) { return floor(a); }
  static __device__ __forceinline__ T ceil_(T a) { return ceil(a); }
  static __device__ __forceinline__ T abs_(T a) { return fabs(a); }
  static __device__ __forceinline__ bool isnan_(T a) { return a != a; }
  static __device__ __forceinline__ T to_f32(T a) { return (T)(float)a; }
  static __device__ __forceinline__ T to_i32(T a) { return (T)(int32_t)a; }
  static __device__ __forceinline__ T to_i64(T a) { return (T)(int64_t)a; }
};

template <>
struct sp_math<int64_t> {
  using T = int64_t;
  static __device__ __forceinline__ T floordiv(T a, T b) {
    if (b == 0) return 0;  // numpy: 0 with a RuntimeWarning
    T q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
    return q;
  }
  static __device__ __forceinline__ T div(T a, T b) { return floordiv(a, b); }
  static __device__ __forceinline__ T fmod_(T a, T b) { return b == 0 ? 0 : a % b; }
  static __device__ __forceinline__ T mod(T a, T b) {
    if (b == 0) return 0;
    T r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) r += b;
    return r;
  }
  static __device__ __forceinline__ T pow_(T a, T b) {
    if (b < 0) return 0;
    T r = 1;
    while (b) {
      if (b & 1) r *= a;
      a *= a;
      b >>= 1;
    }
    return r;
  }
  static __device__ __forceinline__ T sqrt_(T a) { return (T)sqrt((double)a); }
  static __device__ __forceinline__ T exp_(T a) { return (T)exp((double)a); }
  static __device__ __forceinline__ T log_(T a) { return (T)log((double)a); }
  static __device__ __forceinline__ T tanh_(T a) { return (T)tanh((double)a); }
  static __device__ __forceinline__ T normcdf_(T a) { return (T)(0.5 * erfc(-(double)a * 0.70710678118654752440)); }
  static __device__ __forceinline__ T floor_(T a) { return a; }
  static __device__ __forceinline__ T ceil_(T a) { return a; }
  static __device__ __forceinline__ T abs_(T a) { return a < 0 ? -a : a; }
  static __device__ __forceinline__ bool isnan_(T) { return false; }
  static __device__ __forceinline__ T to_f32(T a) { return (T)(float)a; }
  static __device__ __forceinline__ T to_i32(T a) { return (T)(int32_t)a; }
  static __device__ __forceinline__ T to_i64(T a) { return a; }
};

template <typename T>
__device__ __forceinline__ T sp_nanmax(T a, T b) {
  if (sp_math<T>::isnan_(a)) return a;
  if (sp_math<T>::isnan_(b)) return b;
  return a > b ? a : b;
}
template <typename T>
__device__ __forceinline__ T sp_nanmin(T a, T b) {
  if (sp_math<T>::isnan_(a)) return a;
  if (sp_math<T>::isnan_(b)) return b;
  return a < b ? a : b;
}

template <typename T>
__device__ __forceinline__ T sp_const(const sp_program& p, int i) {
  if constexpr (sp_is_integral<T>::value) {
    return (T)p.iconsts[i];
  } else {
    return (T)p.consts[i];
  }
}

// ---- program sources ----------------------------------------------------------
// DynProg: the generic path -- instructions are fetched from the kernel-argument
// copy of sp_program and dispatched at run time (any tree the host can lower).
// StaticProg<ID>: hot shapes whose instruction stream is a compile-time constant:
// the very same evaluator source is then fully unrolled by the compiler (static
// register indices, no dispatch), which is what lets the HBM-bound kernels stream
// at the chip's copy bandwidth.  The host-emitted stream is matched against this
// library (sp_find_static); constants, strides and shapes stay run-time values.
struct DynProg {
  static constexpr bool kStatic = false;
  static constexpr int N = 0;
  static constexpr int NIN = SP_MAX_INPUTS;
  static constexpr int RESULT = 0;
  static __host__ __device__ constexpr sp_instr at(int) { return sp_instr{0, 0, 0, 0, 0, 0, 0, 0}; }
  static __host__ __device__ constexpr int in_dtype(int) { return SP_F32; }
};

template <int ID>
struct StaticProg;

#define SP_I(op, dst, a, b) \
  sp_instr { (uint8_t)(op), (uint8_t)(dst), (uint8_t)(a), (uint8_t)(b), 0, 0, 0, 0 }
// NB: the stream is exposed through a constexpr FUNCTION, not a static array: a
// constexpr array becomes a device global whose loads hipcc does not fold, and
// the "static" kernel would still dispatch at run time (checked in the ISA).
#define SP_DEF_STATIC(ID, NIN_, RESULT_, N_, I0, I1, I2, I3)                         \
  template <>                                                                        \
  struct StaticProg<ID> {                                                            \
    static constexpr bool kStatic = true;                                            \
    static constexpr int N = N_;                                                     \
    static constexpr int NIN = NIN_;                                                 \
    static constexpr int RESULT = RESULT_;                                           \
    static __host__ __device__ constexpr sp_instr at(int pc) {                       \
      return pc == 0 ? I0 : (pc == 1 ? I1 : (pc == 2 ? I2 : I3));                    \
    }                                                                                \
    /* the prebuilt library is all-fp32; run-time specialised programs (sp_jit.hip) \
       carry their own operand dtypes */                                             \
    static __host__ __device__ constexpr int in_dtype(int) { return SP_F32; }        \
  };
#define SP_NOPI SP_I(SP_OP_NOP, 0, 0, 0)

// The streams below are exactly what spartan_amd/lower.py's Emitter produces for
// the named expression (tests/test_hip_kernels.py::test_static_program_library
// asserts that each of them is recognised).
SP_DEF_STATIC(0, 1, 0, 0, SP_NOPI, SP_NOPI, SP_NOPI, SP_NOPI)                                        // x (reduce / argreduce of a tile)
SP_DEF_STATIC(1, 1, 1, 1, SP_I(SP_OP_ADDC, 1, 0, 0), SP_NOPI, SP_NOPI, SP_NOPI)                      // x + c  (c + x)
SP_DEF_STATIC(2, 1, 1, 1, SP_I(SP_OP_SUBC, 1, 0, 0), SP_NOPI, SP_NOPI, SP_NOPI)                      // x - c
SP_DEF_STATIC(3, 1, 1, 1, SP_I(SP_OP_MULC, 1, 0, 0), SP_NOPI, SP_NOPI, SP_NOPI)                      // x * c  (c * x)
SP_DEF_STATIC(4, 1, 1, 1, SP_I(SP_OP_DIVC, 1, 0, 0), SP_NOPI, SP_NOPI, SP_NOPI)                      // x / c
SP_DEF_STATIC(5, 2, 2, 1, SP_I(SP_OP_ADD, 2, 0, 1), SP_NOPI, SP_NOPI, SP_NOPI)                       // a + b
SP_DEF_STATIC(6, 2, 2, 1, SP_I(SP_OP_SUB, 2, 0, 1), SP_NOPI, SP_NOPI, SP_NOPI)                       // a - b
SP_DEF_STATIC(7, 2, 2, 1, SP_I(SP_OP_MUL, 2, 0, 1), SP_NOPI, SP_NOPI, SP_NOPI)                       // a * b (also matrix.vector)
SP_DEF_STATIC(8, 2, 2, 1, SP_I(SP_OP_DIV, 2, 0, 1), SP_NOPI, SP_NOPI, SP_NOPI)                       // a / b
SP_DEF_STATIC(9, 1, 1, 2, SP_I(SP_OP_MUL, 1, 0, 0), SP_I(SP_OP_ADD, 1, 1, 0), SP_NOPI, SP_NOPI)      // x*x + x
SP_DEF_STATIC(10, 3, 3, 2, SP_I(SP_OP_SUB, 3, 1, 2), SP_I(SP_OP_MUL, 3, 0, 3), SP_NOPI, SP_NOPI)     // x * (yp - y) (lreg gradient)
SP_DEF_STATIC(11, 1, 1, 1, SP_I(SP_OP_MUL, 1, 0, 0), SP_NOPI, SP_NOPI, SP_NOPI)                      // x * x
SP_DEF_STATIC(12, 1, 1, 1, SP_I(SP_OP_RSUBC, 1, 0, 0), SP_NOPI, SP_NOPI, SP_NOPI)                    // c - x
#define SP_NUM_STATIC 13
#define SP_FOR_EACH_STATIC(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)

// an instruction in a dword: op 8 | dst 3 | a 4 | b 3 | c 3 (register numbers < SP_NREG = 8, constant indices < 16)
static_assert(SP_NREG <= 8 && SP_MAX_CONSTS <= 16 && SP_MAX_INSTR <= 64, "sp_pack_instr");
__device__ __forceinline__ uint32_t sp_pack_instr(const sp_instr I) {
  return (uint32_t)I.op | ((uint32_t)(I.dst & 7) << 8) | ((uint32_t)(I.a & 15) << 11) | ((uint32_t)(I.b & 7) << 15) |
         ((uint32_t)(I.c & 7) << 18);
}
__device__ __forceinline__ sp_instr sp_unpack_instr(uint32_t w) {
  return sp_instr{(uint8_t)(w & 255), (uint8_t)((w >> 8) & 7), (uint8_t)((w >> 11) & 15), (uint8_t)((w >> 15) & 7),
                  (uint8_t)((w >> 18) & 7), 0, 0, 0};
}

// The interpreted program as the kernel holds it: lane k of `word` is instruction k.  Made ONCE, at kernel entry
// (every lane of the wave active), by sp_dyn_program; the evaluators take it as their last argument.
struct sp_dyn {
  uint32_t word;
  bool lanes_hold_program;
  // constant k of the program, already of the kernel's arithmetic type, in lane k (low / high dword): an operator with
  // a constant operand gets it with v_readlane instead of s_load_dwordx2 + s_waitcnt lgkmcnt(0) + v_cvt on every trip
  uint32_t c_lo = 0u, c_hi = 0u;
};
template <typename P, typename T = float>
__device__ __forceinline__ sp_dyn sp_dyn_program(const sp_program& p) {
  if constexpr (P::kStatic) {
    return sp_dyn{0u, false};
  } else {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t word = sp_pack_instr(p.instr[lane & (SP_MAX_INSTR - 1)]);
    const T c = sp_const<T>(p, (int)(lane & (SP_MAX_CONSTS - 1)));
    uint32_t lo, hi = 0u;
    if constexpr (sizeof(T) == 4) {
      lo = __builtin_bit_cast(uint32_t, c);
    } else {
      const uint64_t w = __builtin_bit_cast(uint64_t, c);
      lo = (uint32_t)w;
      hi = (uint32_t)(w >> 32);
    }
    // (pinned HERE: sunk into a loop that only some lanes enter, the load would leave the other lanes' instructions
    // unread -- v_readlane reads a lane whether or not it is active)
    asm volatile("" : "+v"(word), "+v"(lo), "+v"(hi));
    return sp_dyn{word, true, lo, hi};
  }
}
// constant i as the interpreter reads it: from the lanes when the kernel holds the program there
template <typename T>
__device__ __forceinline__ T sp_const_of(const sp_program& p, const sp_dyn& dyn, int i) {
  if (!dyn.lanes_hold_program) return sp_const<T>(p, i);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)dyn.c_lo, i);
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, lo);
  } else {
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)dyn.c_hi, i);
    return __builtin_bit_cast(T, (uint64_t)lo | ((uint64_t)hi << 32));
  }
}

// one interpreted (DYN) / unrolled instruction on U register files
template <typename T, int V, int U, bool DYN = false>
__device__ __forceinline__ void sp_step(const sp_program& p, const sp_instr I, const int64_t (&L)[U],
                                        T (&r0)[SP_NREG * V], T (&r1)[SP_NREG * V], T (&r2)[SP_NREG * V],
                                        T (&r3)[SP_NREG * V], const sp_dyn dyn = sp_dyn{0u, false}) {
  using M = sp_math<T>;
#define SP_U_LIST(X)                 \
  X(0)                               \
  if constexpr (U > 1) { X(1) }      \
  if constexpr (U > 2) { X(2) X(3) }
  T a[U][V], b[U][V], d[U][V];
  const int ra = (I.a & (SP_NREG - 1)) * V, rb = (I.b & (SP_NREG - 1)) * V;
  const int rd = (I.dst & (SP_NREG - 1)) * V;
#define SP_WR(u) _Pragma("unroll") for (int v = 0; v < V; ++v) r##u[rd + v] = d[u][v];
#define SP_EACH(expr)                               \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {   \
    _Pragma("unroll") for (int v = 0; v < V; ++v) { \
      const T av = a[u][v], bv = b[u][v];           \
      (void)av; (void)bv;                           \
      d[u][v] = (expr);                             \
    }                                               \
  }
  bool done = false;
  if constexpr (DYN) {
    // operator with a constant operand (reg[b] op consts[a]): one register read, tested for before the plain operators
    const unsigned kc = (unsigned)I.op - (unsigned)SP_OP_ADDC;
    if (kc < 8u) {
      done = true;
#define SP_RD(u) _Pragma("unroll") for (int v = 0; v < V; ++v) b[u][v] = r##u[rb + v];
      SP_U_LIST(SP_RD)
#undef SP_RD
      __builtin_amdgcn_sched_barrier(0);
      const T cv = sp_const_of<T>(p, dyn, I.a);
      if (kc < 4u) {
        if (kc < 2u) {
          if (kc == 0u) { SP_EACH(bv + cv); } else { SP_EACH(bv - cv); }
        } else {
          if (kc == 2u) { SP_EACH(cv - bv); } else { SP_EACH(bv * cv); }
        }
      } else {
        if (kc < 6u) {
          if (kc == 4u) { SP_EACH(M::div(bv, cv)); } else { SP_EACH(M::div(cv, bv)); }
        } else {
          if (kc == 6u) { SP_EACH(sp_nanmax<T>(bv, cv)); } else { SP_EACH(sp_nanmin<T>(bv, cv)); }
        }
      }
    }
  }
  if (!done) {
  // (all of `a`, then all of `b`: reads that share an index share one s_set_gpr_idx_on / off bracket)
#define SP_RD(u) _Pragma("unroll") for (int v = 0; v < V; ++v) a[u][v] = r##u[ra + v];
  SP_U_LIST(SP_RD)
#undef SP_RD
  if constexpr (DYN) __builtin_amdgcn_sched_barrier(0);
#define SP_RD(u) _Pragma("unroll") for (int v = 0; v < V; ++v) b[u][v] = r##u[rb + v];
  SP_U_LIST(SP_RD)
#undef SP_RD
  if constexpr (DYN) __builtin_amdgcn_sched_barrier(0);
  if constexpr (DYN) {
    // the four arithmetic operators first: two scalar compares instead of the six levels of the full switch
    const unsigned k = (unsigned)I.op - (unsigned)SP_OP_ADD;
    if (k < 4u) {
      done = true;
      if (k < 2u) {
        if (k == 0u) { SP_EACH(av + bv); } else { SP_EACH(av - bv); }
      } else {
        if (k == 2u) { SP_EACH(av * bv); } else { SP_EACH(M::div(av, bv)); }
      }
    }
  }
  if (!done) switch (I.op) {
    case SP_OP_CONST: { const T c = sp_const_of<T>(p, dyn, I.a); SP_EACH(c); } break;
    case SP_OP_IOTA:
      if constexpr (DYN) {
        // The index -> T conversions depend on nothing the dispatch loop changes: left alone, the compiler computes
        // them ahead of the loop, i.e. in EVERY trip of EVERY interpreted program (8 conversions of a 64-bit integer,
        // 125 of the 223 VALU instructions a trip of `x + 1` issued).  An asm statement is not moved.
        int64_t at[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          at[u] = L[u];
          asm volatile("" : "+v"(at[u]));
        }
        SP_EACH((T)(at[u] + v));
      } else {
        SP_EACH((T)(L[u] + v));
      }
      break;
    case SP_OP_MOV: SP_EACH(av); break;
    case SP_OP_ADD: SP_EACH(av + bv); break;
    case SP_OP_SUB: SP_EACH(av - bv); break;
    case SP_OP_MUL: SP_EACH(av * bv); break;
    case SP_OP_DIV: SP_EACH(M::div(av, bv)); break;
    case SP_OP_FLOORDIV: SP_EACH(M::floordiv(av, bv)); break;
    case SP_OP_MOD: SP_EACH(M::mod(av, bv)); break;
    case SP_OP_FMOD: SP_EACH(M::fmod_(av, bv)); break;
    case SP_OP_POW: SP_EACH(M::pow_(av, bv)); break;
    case SP_OP_MAX: SP_EACH(sp_nanmax<T>(av, bv)); break;
    case SP_OP_MIN: SP_EACH(sp_nanmin<T>(av, bv)); break;
    case SP_OP_EQ: SP_EACH((T)(av == bv)); break;
    case SP_OP_NE: SP_EACH((T)(av != bv)); break;
    case SP_OP_LT: SP_EACH((T)(av < bv)); break;
    case SP_OP_LE: SP_EACH((T)(av <= bv)); break;
    case SP_OP_GT: SP_EACH((T)(av > bv)); break;
    case SP_OP_GE: SP_EACH((T)(av >= bv)); break;
    case SP_OP_LAND: SP_EACH((T)((av != (T)0) && (bv != (T)0))); break;
    case SP_OP_LOR: SP_EACH((T)((av != (T)0) || (bv != (T)0))); break;
    case SP_OP_LXOR: SP_EACH((T)((av != (T)0) != (bv != (T)0))); break;
    case SP_OP_LNOT: SP_EACH((T)(av == (T)0)); break;
    case SP_OP_NEG: SP_EACH(-av); break;
    case SP_OP_ABS: SP_EACH(M::abs_(av)); break;
    case SP_OP_SQRT: SP_EACH(M::sqrt_(av)); break;
    case SP_OP_SQUARE: SP_EACH(av * av); break;
    case SP_OP_EXP: SP_EACH(M::exp_(av)); break;
    case SP_OP_LOG: SP_EACH(M::log_(av)); break;
    case SP_OP_RECIP: SP_EACH(M::div((T)1, av)); break;
    case SP_OP_SIGN: SP_EACH((T)((av > (T)0) - (av < (T)0))); break;
    case SP_OP_FLOOR: SP_EACH(M::floor_(av)); break;
    case SP_OP_CEIL: SP_EACH(M::ceil_(av)); break;
    case SP_OP_TANH: SP_EACH(M::tanh_(av)); break;
    case SP_OP_NORM_CDF: SP_EACH(M::normcdf_(av)); break;
    case SP_OP_WHERE: {
      const int rc = (I.c & (SP_NREG - 1)) * V;
      T c[U][V];
#define SP_RC(u) _Pragma("unroll") for (int v = 0; v < V; ++v) c[u][v] = r##u[rc + v];
      SP_U_LIST(SP_RC)
#undef SP_RC
      SP_EACH(av != (T)0 ? bv : c[u][v]);
    } break;
    case SP_OP_TO_F32: SP_EACH(M::to_f32(av)); break;
    case SP_OP_TO_I32: SP_EACH(M::to_i32(av)); break;
    case SP_OP_TO_I64: SP_EACH(M::to_i64(av)); break;
    case SP_OP_TO_BOOL: SP_EACH((T)(av != (T)0)); break;
    case SP_OP_TO_U8: SP_EACH((T)(uint8_t)(int64_t)av); break;
    case SP_OP_ADDC: { const T cv = sp_const<T>(p, I.a); SP_EACH(bv + cv); } break;
    case SP_OP_SUBC: { const T cv = sp_const<T>(p, I.a); SP_EACH(bv - cv); } break;
    case SP_OP_RSUBC: { const T cv = sp_const<T>(p, I.a); SP_EACH(cv - bv); } break;
    case SP_OP_MULC: { const T cv = sp_const<T>(p, I.a); SP_EACH(bv * cv); } break;
    case SP_OP_DIVC: { const T cv = sp_const<T>(p, I.a); SP_EACH(M::div(bv, cv)); } break;
    case SP_OP_RDIVC: { const T cv = sp_const<T>(p, I.a); SP_EACH(M::div(cv, bv)); } break;
    case SP_OP_MAXC: { const T cv = sp_const<T>(p, I.a); SP_EACH(sp_nanmax<T>(bv, cv)); } break;
    case SP_OP_MINC: { const T cv = sp_const<T>(p, I.a); SP_EACH(sp_nanmin<T>(bv, cv)); } break;
    default: SP_EACH(av); break;
  }
#undef SP_EACH
  }
  SP_U_LIST(SP_WR)
#undef SP_WR
#undef SP_U_LIST
}

// Operands of the NEXT evaluation, fetched while this one computes (interpreted dense programs: with 3 waves per
// SIMD resident, load -> wait -> interpret -> store one after the other leaves HBM idle while the waves interpret).
// The first NPRE operands of each of the U groups are held; sp_eval_u moves them into its register file, calls the
// caller's `mid` -- which requests the next ones into the same slots and issues the stores of the PREVIOUS
// evaluation, so that the wait at the top of an evaluation is for loads a whole evaluation old and never for a
// store just issued -- and only then runs the program.
struct sp_no_ahead {
  __device__ __forceinline__ void operator()() const {}
};
template <typename T, int V, int U, int NPRE>
struct sp_ahead {
  static constexpr int N = NPRE;
  T v[U][NPRE][V];
};

// operand j of a dense (LINEAR) program at flat index L: V consecutive elements, or its one element V times
template <typename T, int V, int NTM>
__device__ __forceinline__ void sp_load_linear(const sp_program& p, const sp_inputs& in, int j, int32_t dt, int64_t L,
                                               T* dst) {
  if (p.in_stride[j][p.ndim - 1] != 0) {
    if (SP_STREAMS(NTM, p)) sp_load_vec<T, V, true>(in.p[j], dt, L, dst);   // dense: read exactly once
    else sp_load_vec<T, V>(in.p[j], dt, L, dst);
  } else {
    T s;
    sp_load_vec<T, 1>(in.p[j], dt, 0, &s);
#pragma unroll
    for (int v = 0; v < V; ++v) dst[v] = s;
  }
}

// (taken for operands that are all dense fp32 -- sp_ahead_applies: one vector load each, no dtype dispatch)
template <typename T, int V, int U, int NTM, typename AH>
__device__ __forceinline__ void sp_fetch_ahead(const sp_program& p, const sp_inputs& in, const int64_t (&L)[U], AH& ah) {
  static_assert(sp_is_same<T, float>::value && V == 4, "operand prefetch: the fp32 class");
  const bool streams = SP_STREAMS(NTM, p);
#pragma unroll
  for (int j = 0; j < AH::N; ++j)
    if (j < p.n_inputs) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* src = (const float*)in.p[j] + L[u];
        sp_f32x4u x;
        if (streams) x = __builtin_nontemporal_load((const sp_f32x4u*)src);
        else x = *(const sp_f32x4u*)src;
        ah.v[u][j][0] = x.x; ah.v[u][j][1] = x.y; ah.v[u][j][2] = x.z; ah.v[u][j][3] = x.w;
      }
    }
}
template <typename T, int N>
__device__ __forceinline__ bool sp_ahead_applies(const sp_program& p) {
  if (!sp_is_same<T, float>::value || p.n_inputs > N || p.n_inputs < 1) return false;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (j < p.n_inputs) ok = ok && p.in_dtype[j] == SP_F32 && p.in_stride[j][p.ndim - 1] != 0;
  return ok;
}

// ---- the evaluator ----------------------------------------------------------
// Evaluates the program for U groups of V consecutive elements; group u starts
// at row-major linear index L[u] (callers guarantee the V elements of a group
// share every coordinate but the last when !LINEAR).  Results in out[u][0..V).
// Each group has its own 8 x V register file so that every dynamically indexed
// array stays within the 32 dwords the s_set_gpr_idx path handles (a [U][32]
// array would be one 32*U-dword alloca and go to scratch).
// `ahead` / `mid`: see sp_ahead -- the operands come from `ahead`; `mid()` runs once they sit in the register file and
// before the program does (the caller's place for the next fetch and for the stores it held back).
template <typename T, int V, int U, bool LINEAR, typename P = DynProg, int NTM = 2, typename AH = sp_no_ahead,
          typename MID = sp_no_ahead>
__device__ __forceinline__ void sp_eval_u(const sp_program& p, const sp_inputs& in, const int64_t (&L)[U],
                                          T (&out)[U][V], const int64_t (*pre)[2] = nullptr,
                                          const sp_dyn dyn = sp_dyn{0u, false}, AH* ahead = nullptr,
                                          const MID& mid = MID()) {
  constexpr bool AHEAD = !sp_is_same<AH, sp_no_ahead>::value;
  static_assert(!AHEAD || (LINEAR && !P::kStatic), "operand prefetch: interpreted dense programs");
  T r0[SP_NREG * V], r1[SP_NREG * V], r2[SP_NREG * V], r3[SP_NREG * V];
  static_assert(U == 1 || U == 2 || U == 4, "U must be 1, 2 or 4");
#define SP_U_LIST(X)                 \
  X(0)                               \
  if constexpr (U > 1) { X(1) }      \
  if constexpr (U > 2) { X(2) X(3) }
  // Specialised programs: the zeros fold away.  Interpreted programs never read a register nothing has written
  // (sp_validate_program), so their register files start as whatever the VGPRs hold: 32 moves per evaluation less.
#define SP_ZERO(u)                                                     \
  _Pragma("unroll") for (int k = 0; k < SP_NREG * V; ++k) {            \
    if constexpr (P::kStatic) r##u[k] = (T)0;                          \
    else asm volatile("" : "=v"(r##u[k]));                             \
  }
  SP_U_LIST(SP_ZERO)
#undef SP_ZERO
  (void)r1; (void)r2; (void)r3;

  // coordinates of L[u] for the strided path
  // Coordinates of L[u] for the strided path.  p.pad != 0 (set by the launcher):
  // the whole index space -- hence every operand offset -- fits 32 bits, so the
  // row/column split and the offset arithmetic are 32-bit (the 64-bit versions
  // made these HBM kernels VALU-bound).  `pre`: the caller already knows the
  // (row, column) of group 0 of a 2-D program (reduce kernels): no division.
  int64_t idx[U][SP_MAX_DIMS];
  uint32_t idx32[U][SP_MAX_DIMS];
  if constexpr (!LINEAR) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int d = 0; d < SP_MAX_DIMS; ++d) {
        idx[u][d] = 0;
        idx32[u][d] = 0;
      }
      if (U == 1 && pre != nullptr) {
        idx[u][0] = pre[0][0];
        idx[u][1] = pre[0][1];
        idx32[u][0] = (uint32_t)pre[0][0];
        idx32[u][1] = (uint32_t)pre[0][1];
      } else if (p.pad & SP_PAD_IDX32) {
        uint32_t rem = (uint32_t)L[u];
#pragma unroll
        for (int d = SP_MAX_DIMS - 1; d >= 0; --d) {
          if (d < p.ndim) {
            const uint32_t s = (uint32_t)p.shape[d];
            if (d == 0) {
              idx32[u][d] = rem;
            } else {
              const uint32_t q = rem / s;
              idx32[u][d] = rem - q * s;
              rem = q;
            }
          }
        }
      } else {
        int64_t rem = L[u];
#pragma unroll
        for (int d = SP_MAX_DIMS - 1; d >= 0; --d) {
          if (d < p.ndim) {
            int64_t s = p.shape[d];
            if (d == 0) {
              idx[u][d] = rem;
            } else {
              int64_t q = rem / s;
              idx[u][d] = rem - q * s;
              rem = q;
            }
          }
        }
      }
    }
  }

  // operand loads: static register slots, so all loads are in flight together.
  // Static programs are only selected for all-fp32 operands: the dtype is a constant.
#pragma unroll
  for (int j = 0; j < SP_MAX_INPUTS; ++j) {
    if (j < (P::kStatic ? P::NIN : p.n_inputs)) {
      const int32_t dt = P::kStatic ? (int32_t)P::in_dtype(j) : p.in_dtype[j];
      if constexpr (LINEAR) {
        // dense operand (stride pattern == output) or scalar (all strides 0)
        if constexpr (AHEAD) {
          // fetched during the previous evaluation.  (The caller takes this path for programs of at most AH::N
          // operands only: NO load may target the register file itself here -- with one pending, every indexed
          // access of the dispatch loop would wait for vmcnt(0), i.e. for the operands just requested as well.)
          if (j < AH::N) {
#define SP_TAKE(u) _Pragma("unroll") for (int v = 0; v < V; ++v) r##u[j * V + v] = ahead->v[u][j < AH::N ? j : 0][v];
            SP_U_LIST(SP_TAKE)
#undef SP_TAKE
          }
        } else {
#define SP_LD(u) sp_load_linear<T, V, NTM>(p, in, j, dt, L[u], &r##u[j * V]);
          SP_U_LIST(SP_LD)
#undef SP_LD
        }
      } else {
        const int64_t inner = p.in_stride[j][p.ndim - 1];
        bool once = SP_STREAMS(NTM, p);   // big launch, no broadcast dimension: every element is read once
#pragma unroll
        for (int d = 0; d < SP_MAX_DIMS; ++d)
          if (d < p.ndim && p.shape[d] > 1 && p.in_stride[j][d] == 0) once = false;
#define SP_LDS(u)                                                                              \
  {                                                                                            \
    int64_t off = 0;                                                                           \
    if (p.pad & SP_PAD_IDX32) {                                                                \
      uint32_t o32 = 0;                                                                        \
      _Pragma("unroll") for (int d = 0; d < SP_MAX_DIMS; ++d)                                  \
          if (d < p.ndim) o32 += idx32[u][d] * (uint32_t)p.in_stride[j][d];                    \
      off = (int64_t)o32;                                                                      \
    } else {                                                                                   \
      _Pragma("unroll") for (int d = 0; d < SP_MAX_DIMS; ++d)                                  \
          if (d < p.ndim) off += idx[u][d] * p.in_stride[j][d];                                \
    }                                                                                          \
    if ((inner == 1 || V == 1) && once) {                                                      \
      sp_load_vec<T, V, true>(in.p[j], dt, off, &r##u[j * V]);                                 \
    } else if (inner == 1 || V == 1) {                                                         \
      sp_load_vec<T, V>(in.p[j], dt, off, &r##u[j * V]);                                       \
    } else if (inner == 0) {                                                                   \
      T s;                                                                                     \
      sp_load_vec<T, 1>(in.p[j], dt, off, &s);                                                 \
      _Pragma("unroll") for (int v = 0; v < V; ++v) r##u[j * V + v] = s;                       \
    } else {                                                                                   \
      _Pragma("unroll") for (int v = 0; v < V; ++v)                                            \
          sp_load_vec<T, 1>(in.p[j], dt, off + v * inner, &r##u[j * V + v]);                   \
    }                                                                                          \
  }
        SP_U_LIST(SP_LDS)
#undef SP_LDS
      }
    }
  }

  if constexpr (AHEAD) mid();
  if constexpr (P::kStatic) {
#pragma unroll
    for (int pc = 0; pc < P::N; ++pc) sp_step<T, V, U>(p, P::at(pc), L, r0, r1, r2, r3);
  } else {
    if (dyn.lanes_hold_program) {
      // fetched with v_readlane from the kernel's lane-held copy of the stream (sp_dyn_program): no trip to the scalar
      // cache, no s_waitcnt lgkmcnt(0) between fetch and execution
      for (int pc = 0; pc < p.n_instr; ++pc) {
        const sp_instr cur = sp_unpack_instr((uint32_t)__builtin_amdgcn_readlane((int)dyn.word, pc));
        if (cur.op != SP_OP_NOP) sp_step<T, V, U, true>(p, cur, L, r0, r1, r2, r3, dyn);
      }
    } else {
      for (int pc = 0; pc < p.n_instr; ++pc)
        if (p.instr[pc].op != SP_OP_NOP) sp_step<T, V, U, true>(p, p.instr[pc], L, r0, r1, r2, r3);
    }
  }
  const int rr = ((P::kStatic ? P::RESULT : p.result_reg) & (SP_NREG - 1)) * V;
#define SP_OUT(u) _Pragma("unroll") for (int v = 0; v < V; ++v) out[u][v] = r##u[rr + v];
  SP_U_LIST(SP_OUT)
#undef SP_OUT
#undef SP_U_LIST
}

template <typename T, int V, bool LINEAR, typename P = DynProg>
__device__ __forceinline__ void sp_eval(const sp_program& p, const sp_inputs& in, int64_t L, T (&out)[V]) {
  const int64_t Ls[1] = {L};
  T o[1][V];
  sp_eval_u<T, V, 1, LINEAR, P>(p, in, Ls, o);
#pragma unroll
  for (int v = 0; v < V; ++v) out[v] = o[0][v];
}

// ---- 2-D strided evaluator for specialised programs -----------------------------
// The general strided path decides per operand, per evaluation, how to address
// it (dense inner run / broadcast inner / arbitrary stride, 32- or 64-bit): a few
// dozen wave-uniform branches per trip, which is what bounded the fused
// broadcast kernels (lreg gradient, matrix.vector) at ~45% of the copy bandwidth.
// Here every such decision is a compile-time constant: the program space is 2-D
// and fits 32 bits, operand j is read with ONE 16-B load when bit j of MASK is
// clear (inner stride 1) and with one broadcast dword when it is set (inner
// stride 0); offsets are row*s0 + col*s1 in 32-bit arithmetic.
template <typename T, int V, typename P, int MASK, int NTM = 2>
__device__ __forceinline__ void sp_eval_2d(const sp_program& p, const sp_inputs& in, uint32_t row, uint32_t col,
                                           int64_t L, T (&out)[V]) {
  static_assert(P::kStatic, "sp_eval_2d is for specialised programs");
  T r0[SP_NREG * V], r1[SP_NREG * V], r2[SP_NREG * V], r3[SP_NREG * V];
#pragma unroll
  for (int k = 0; k < SP_NREG * V; ++k) r0[k] = (T)0;
  (void)r1; (void)r2; (void)r3;
#pragma unroll
  for (int j = 0; j < P::NIN; ++j) {
    const uint32_t off = row * (uint32_t)p.in_stride[j][0] + col * (uint32_t)p.in_stride[j][1];
    if (((MASK >> j) & 1) != 0) {  // folds once the loop is unrolled
      T s;
      sp_load_vec<T, 1>(in.p[j], P::in_dtype(j), (int64_t)off, &s);
#pragma unroll
      for (int v = 0; v < V; ++v) r0[j * V + v] = s;
    } else if (SP_STREAMS(NTM, p) && (p.in_stride[j][0] != 0 || p.shape[0] == 1)) {   // not a re-read row vector
      sp_load_vec<T, V, true>(in.p[j], P::in_dtype(j), (int64_t)off, &r0[j * V]);
    } else {
      sp_load_vec<T, V>(in.p[j], P::in_dtype(j), (int64_t)off, &r0[j * V]);
    }
  }
  const int64_t Ls[1] = {L};
#pragma unroll
  for (int pc = 0; pc < P::N; ++pc) sp_step<T, V, 1>(p, P::at(pc), Ls, r0, r1, r2, r3);
#pragma unroll
  for (int v = 0; v < V; ++v) out[v] = r0[(P::RESULT & (SP_NREG - 1)) * V + v];
}

// Host side: is the 2-D specialised path applicable, and with which MASK?
// Returns -1 if not (then the general strided evaluator is used).
static inline int sp_mask_2d(const sp_program* p, int nin) {
  if (p->ndim != 2 || !(p->pad & SP_PAD_IDX32) || p->linear) return -1;
  int mask = 0;
  for (int j = 0; j < nin; ++j) {
    const int64_t s1 = p->in_stride[j][1];
    if (s1 == 0) mask |= 1 << j;
    else if (s1 != 1) return -1;
    if (p->in_stride[j][0] < 0 || p->in_stride[j][0] >= (1LL << 32)) return -1;
  }
  return mask;
}

// Host side: the copy of the program handed to a kernel; `pad` carries launch flags: SP_PAD_IDX32 "index space
// fits 32 bits" (strided evaluator), SP_PAD_STREAM "the launch walks more elements than the L2 holds": operands
// read exactly once and the output are then accessed non-temporally (see sp_load_vec).
static inline sp_program sp_prepare_program(const sp_program* p) {
  sp_program q = *p;
  int64_t n = 1;
  for (int d = 0; d < p->ndim; ++d) n *= p->shape[d];
  q.pad = (n > 0 && n < (1LL << 32)) ? SP_PAD_IDX32 : 0;
  if (n >= SP_STREAM_ELEMS) q.pad |= SP_PAD_STREAM;
  return q;
}

// Host side: which StaticProg (if any) has exactly this instruction stream?
// Only all-fp32 programs (class, operands, result) qualify.
static inline int sp_find_static(const sp_program* p, int32_t out_dtype) {
  if (p->cls != SP_F32 || (out_dtype >= 0 && out_dtype != SP_F32)) return -1;
  for (int j = 0; j < p->n_inputs; ++j)
    if (p->in_dtype[j] != SP_F32) return -1;
#define SP_MATCH(ID)                                                                          \
  {                                                                                           \
    using S = StaticProg<ID>;                                                                 \
    bool ok = p->n_inputs == S::NIN && p->n_instr == S::N && p->result_reg == S::RESULT;      \
    for (int i = 0; ok && i < S::N; ++i) {                                                    \
      const sp_instr& x = p->instr[i];                                                        \
      const sp_instr y = S::at(i);                                                          \
      ok = x.op == y.op && x.dst == y.dst && x.a == y.a && (x.op == SP_OP_CONST || x.b == y.b); \
    }                                                                                         \
    if (ok) return ID;                                                                        \
  }
  SP_FOR_EACH_STATIC(SP_MATCH)
#undef SP_MATCH
  return -1;
}
