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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): thread-safe, and a second GPU in the
// same process gets its own call (a function-local `static bool` is neither).  Defined in pack.hip.
hipError_t vg_lds_attr(const void* kernel, int bytes);

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

// ---------------------------------------------------------------------------------------------
// Per-channel statistics accumulators: stats[VG_STATS_REPLICAS][2][cpad], 8 bytes per slot, zeroed by the caller.
//   det == 0: the slots are doubles; a block adds its partial sum to replica (hint % VG_STATS_REPLICAS) with an f64
//             atomic.  The order of the adds, hence the last bits of the sum, changes from run to run.
//   det != 0: the slots are 64-bit integers, a fixed-point superaccumulator: a partial sum is cut into f32 pieces
//             (exactly), a piece M * 2^(E-150) (24-bit mantissa M) is added as the integer M << (E % 24) to bin E / 24
//             (VG_DET_BINS bins x 2 sub-replicas fill 22 of the 32 replica slots).  Integer addition is associative, so
//             the slots, and everything computed from them, are the same bits on every run.  47-bit addends leave room
//             for 2^16 adds per slot: with the two sub-replicas a launch may add from 2^17 blocks per channel, and every
//             launcher that hands `det` to a kernel checks its block count first (VG_DET_GUARD).  Pieces below 2^-126
//             are dropped.  A partial sum that is not finite (or beyond the f32 range) raises the marker slot
//             VG_DET_NAN_SLOT instead, and the read then returns NaN, as the f64 atomics would have propagated it.
// vg_stats_read sums the slots in a fixed order either way.  (vg_api.h: vg_set_deterministic.)
// ---------------------------------------------------------------------------------------------
#define VG_DET_BINS 11
#define VG_DET_NAN_SLOT (2 * VG_DET_BINS)
#define VG_DET_MAX_BLOCKS (1L << 17)
int vg_det_mode(void);      // host: the library-wide switch (pack.hip)
// host: refuse a launch whose block count could overflow a fixed-point slot (adds per channel <= blocks)
#define VG_DET_GUARD(blocks)                                                                                         \
  VG_REQUIRE(!vg_det_mode() || (long)(blocks) <= VG_DET_MAX_BLOCKS,                                                  \
             "deterministic statistics: %ld blocks add to one accumulator, the fixed-point slots hold %ld (split the batch)", \
             (long)(blocks), VG_DET_MAX_BLOCKS)
// slot0 = replica 0 of the accumulator, rep_stride = distance (in slots) between replicas
__device__ inline void vg_acc_add(double* slot0, size_t rep_stride, unsigned hint, double v, int det) {
  if (!det) {
    unsafeAtomicAdd(&slot0[(size_t)(hint % VG_STATS_REPLICAS) * rep_stride], v);
    return;
  }
  if (!(fabs(v) <= 3.4e38)) {                      // NaN, infinity, or beyond what the f32 pieces can carry
    atomicOr((unsigned long long*)&slot0[(size_t)VG_DET_NAN_SLOT * rep_stride], 1ull);
    return;
  }
#pragma unroll 1
  for (int k = 0; k < 3 && v != 0.0; ++k) {
    const float f = (float)v;
    v -= (double)f;
    int e;
    const float m = frexpf(f, &e);               // f = m * 2^e, |m| in [0.5, 1)
    const int E = e - 24 + 150;                  // f = (m * 2^24) * 2^(E - 150)
    if (E < 0 || f == 0.f) continue;
    const long long M = (long long)(m * 16777216.f);
    const int b = E / 24, sh = E - b * 24;
    const int rep = b + VG_DET_BINS * (int)(hint & 1u);
    atomicAdd((unsigned long long*)&slot0[(size_t)rep * rep_stride], (unsigned long long)(M << sh));
  }
}
__device__ inline double vg_acc_read(const double* slot0, size_t rep_stride, int det) {
  // all loads first (the finalize kernels are one memory latency long, not 32), then the sum in a fixed order
  double s = 0.0;
  if (!det) {
    double v[VG_STATS_REPLICAS];
#pragma unroll
    for (int r = 0; r < VG_STATS_REPLICAS; ++r) v[r] = slot0[(size_t)r * rep_stride];
#pragma unroll
    for (int r = 0; r < VG_STATS_REPLICAS; ++r) s += v[r];
    return s;
  }
  const long long* q = (const long long*)slot0;
  long long t[VG_DET_BINS];
  const long long bad = q[(size_t)VG_DET_NAN_SLOT * rep_stride];
#pragma unroll
  for (int b = 0; b < VG_DET_BINS; ++b) t[b] = q[(size_t)b * rep_stride] + q[(size_t)(b + VG_DET_BINS) * rep_stride];
#pragma unroll
  for (int b = VG_DET_BINS - 1; b >= 0; --b) s += ldexp((double)t[b], 24 * b - 150);
  return bad ? __builtin_nan("") : s;
}
// the [VG_STATS_REPLICAS][2][cpad] layout of the BatchNorm sums (which: 0 = first sum, 1 = second)
__device__ inline void vg_stats_add(double* stats, int cpad, int which, int ch, unsigned hint, double v, int det) {
  vg_acc_add(stats + (size_t)which * cpad + ch, (size_t)2 * cpad, hint, v, det);
}
__device__ inline double vg_stats_read(const double* stats, int cpad, int which, int ch, int det) {
  return vg_acc_read(stats + (size_t)which * cpad + ch, (size_t)2 * cpad, det);
}

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
