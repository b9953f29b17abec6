// Shared device/host helpers for the gfx950 VAE-GAN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vg_api.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define VG_LRELU_SLOPE 0.2f
#define VG_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing (no exceptions cross the ABI; message via vg_last_error())
// ---------------------------------------------------------------------------------------------
void vg_set_error(const char* fmt, ...);

#define VG_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      vg_set_error(__VA_ARGS__);         \
      return VG_ERR_BAD_ARG;             \
    }                                    \
  } while (0)

#define VG_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      vg_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VG_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

#define VG_LAUNCH_CHECK()                                                         \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      vg_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return VG_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

static inline int vg_elt_size(int dtype) { return dtype == VG_BF16 ? 2 : 4; }
static inline int64_t vg_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// element traits: a "piece" is 16 bytes = 8 bf16 or 4 f32; activations are NHWC with the
// channel stride a multiple of one 64-byte K-chunk (32 bf16 / 16 f32).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int PIECE = 4;    // elements per 16 B
  static constexpr int KC = 16;      // elements per 64-byte K chunk
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <> struct Elt<bf16_t> {
  static constexpr int PIECE = 8;
  static constexpr int KC = 32;
  __device__ static inline float to_f(bf16_t v) { return (float)v; }
  __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }
};

// 16-byte vector of T with float access
template <typename T> struct Piece;
template <> struct Piece<float> {
  f32x4 v;
  __device__ inline float get(int i) const { return v[i]; }
  __device__ inline void set(int i, float f) { v[i] = f; }
};
template <> struct Piece<bf16_t> {
  bf16x8 v;
  __device__ inline float get(int i) const { return (float)v[i]; }
  __device__ inline void set(int i, float f) { v[i] = (bf16_t)f; }
};

__device__ inline float vg_lrelu(float x) { return x > 0.f ? x : x * VG_LRELU_SLOPE; }
__device__ inline float vg_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
