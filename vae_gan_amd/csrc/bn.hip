// BatchNorm (training mode) finalize / apply / backward, LeakyReLU / residual fusion and small
// elementwise helpers.  All of these are HBM-bandwidth kernels: 16-byte vector accesses, channel
// index = (element index) % cs, per-channel f64 atomics spread over VG_STATS_REPLICAS copies.
#include "common.h"

// ---------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const double* __restrict__ stats, int c, int cpad, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var, int64_t* num_batches,
                                   int updates, float* mean, float* invstd, float* scale, float* shift, int det) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch == 0 && num_batches != nullptr) *num_batches += updates;
  if (ch >= c) return;
  // everything this thread reads is requested before the sums are used: one memory latency for the whole kernel
  const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
  const float rm0 = running_mean ? running_mean[ch] : 0.f, rv0 = running_mean ? running_var[ch] : 0.f;
  const double s1 = vg_stats_read(stats, cpad, 0, ch, det), s2 = vg_stats_read(stats, cpad, 1, ch, det);
  double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)m;
  invstd[ch] = is;
  float sc = g * is;
  scale[ch] = sc;
  shift[ch] = b - (float)m * sc;
  if (running_mean != nullptr) {
    double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    float rm = rm0, rv = rv0;
    for (int u = 0; u < updates; ++u) {
      rm = (1.f - momentum) * rm + momentum * (float)m;
      rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
    running_mean[ch] = rm;
    running_var[ch] = rv;
  }
}

extern "C" int vg_bn_finalize(const double* stats, int c, int cpad, int64_t count, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches,
                              int updates, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  VG_REQUIRE(stats && mean && invstd && scale && shift && c > 0 && cpad >= c && count > 0 && updates >= 0,
             "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, c, cpad,
                     (double)count, gamma, beta, eps, momentum, running_mean, running_var, num_batches, updates, mean,
                     invstd, scale, shift, vg_det_mode());
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// PIECE consecutive per-channel coefficients starting at channel ch0 (zeros beyond c): two 16-byte loads when the
// piece is whole and aligned, instead of PIECE dependent scalar loads per array at the head of every thread.
template <int PIECE>
__device__ __forceinline__ void load_coef(const float* __restrict__ a, int ch0, int c, float* out) {
  if (ch0 + PIECE <= c && (((uintptr_t)(a + ch0)) & 15) == 0) {
#pragma unroll
    for (int q = 0; q < PIECE / 4; ++q) {
      const f32x4 t = *(const f32x4*)(a + ch0 + 4 * q);
      out[4 * q] = t[0]; out[4 * q + 1] = t[1]; out[4 * q + 2] = t[2]; out[4 * q + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int e = 0; e < PIECE; ++e) out[e] = ch0 + e < c ? a[ch0 + e] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Thread (rr, pc) owns one 16-byte channel piece `pc` for the rows rr, rr+rpb*grid, ...: its PIECE
// scale/shift coefficients stay in registers, and a wave still touches 1 KiB of contiguous memory.
// ACT >= 0: the activation is a compile-time constant (LeakyReLU: every BatchNorm of the training step); ACT < 0: the
// run-time argument decides.  With the run-time test alone the compiler kept four scalar branches PER ELEMENT in the
// unrolled loops (the sigmoid's division sequence behind one of them): instruction issue, not memory, paced the pass.
template <typename T, bool HAS_RES, int ACT>
__global__ __launch_bounds__(256, 4) void bn_act_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int64_t rows, int ppr, int c, int act_rt) {
  constexpr int PIECE = Elt<T>::PIECE;
  const int act = ACT >= 0 ? ACT : act_rt;
  const int rpb = 256 / ppr;
  const int pc = threadIdx.x % ppr, rr = threadIdx.x / ppr;
  if (rr >= rpb) return;
  float sc[PIECE], sh[PIECE];
  load_coef<PIECE>(scale, pc * PIECE, c, sc);
  load_coef<PIECE>(shift, pc * PIECE, c, sh);
  const size_t cs = (size_t)ppr * PIECE;
  auto finish = [&](const Piece<T>& v, const Piece<T>& rs, size_t off) {
    Piece<T> o;
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
      float f = v.get(e) * sc[e] + sh[e];
      if (HAS_RES) f += rs.get(e);
      if (act == VG_ACT_LRELU) f = vg_lrelu(f);
      else if (act == VG_ACT_SIGMOID) f = vg_sigmoid(f);
      if (pc * PIECE + e >= c) f = 0.f;
      o.set(e, f);
    }
    *(decltype(o.v)*)(y + off) = o.v;
  };
  // U rows in flight per thread: all loads of a round are issued before the first use (memory-level parallelism)
  constexpr int U = 4;
  const int64_t stride = rpb;                            // a block walks its own contiguous chunk of rows
  const int64_t chunk = (((rows + gridDim.x - 1) / gridDim.x + rpb * U - 1) / (rpb * U)) * (rpb * U);
  const int64_t rend = min(rows, (int64_t)(blockIdx.x + 1) * chunk);
  int64_t r = (int64_t)blockIdx.x * chunk + rr;
  for (; r + (U - 1) * stride < rend; r += U * stride) {
    Piece<T> v[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * stride) * cs + pc * PIECE;
      v[u].v = *(const decltype(v[u].v)*)(x + off);
      if (HAS_RES) rs[u].v = *(const decltype(rs[u].v)*)(res + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) finish(v[u], rs[u], (size_t)(r + u * stride) * cs + pc * PIECE);
  }
  for (; r < rend; r += stride) {
    const size_t off = (size_t)r * cs + pc * PIECE;
    Piece<T> v, rs;
    v.v = *(const decltype(v.v)*)(x + off);
    if (HAS_RES) rs.v = *(const decltype(rs.v)*)(res + off);
    finish(v, rs, off);
  }
}

static int row_blocks(int64_t rows, int rpb, int rows_per_thread) {
  int64_t b = vg_ceil_div(rows, (int64_t)rpb * rows_per_thread);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

static int ew_blocks(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 2048 * 4) b = 2048 * 4;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int vg_bn_act_fwd(const void* x, const void* res, void* y, const float* scale, const float* shift, int64_t rows,
                             int cs, int c, int act, int dtype, void* stream) {
  VG_REQUIRE(x && y && scale && shift && rows > 0 && cs >= c, "bn_act_fwd: bad arguments");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0 && cs / piece <= 256, "bn_act_fwd: channel stride %d unsupported", cs);
  const int ppr = cs / piece;
  const int64_t np = rows;
  hipStream_t st = (hipStream_t)stream;
  static const int fwd_cap = [] { const char* e = getenv("VG_BN_FWD_CAP"); return e ? atoi(e) : 16384; }();
  static const int fwd_rpt = [] { const char* e = getenv("VG_BN_FWD_RPT"); return e ? atoi(e) : 4; }();
  int nb_fwd = row_blocks(rows, 256 / ppr, fwd_rpt);
  if (nb_fwd > fwd_cap) nb_fwd = fwd_cap;
  dim3 g(nb_fwd), b(256);
#define VG_BN_FWD(T, R, A)                                                                                                  \
  hipLaunchKernelGGL((bn_act_fwd_kernel<T, R, A>), g, b, 0, st, (const T*)x, (const T*)res, (T*)y, scale, shift, np, ppr, c, act)
#define VG_BN_FWD_T(T)                                                                                                       \
  do {                                                                                                                       \
    if (act == VG_ACT_LRELU) { if (res) VG_BN_FWD(T, true, VG_ACT_LRELU); else VG_BN_FWD(T, false, VG_ACT_LRELU); }          \
    else { if (res) VG_BN_FWD(T, true, -1); else VG_BN_FWD(T, false, -1); }                                                  \
  } while (0)
  if (dtype == VG_BF16) VG_BN_FWD_T(bf16_t); else VG_BN_FWD_T(float);
#undef VG_BN_FWD_T
#undef VG_BN_FWD
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: per-channel sums of dz and dz*xhat.
// Block = 256 threads = (256/ppr') rows x ppr' piece-columns; every thread keeps PIECE channel
// accumulators in registers over its rows, then the block combines through LDS and issues one f64
// atomic per channel.
// ---------------------------------------------------------------------------------------------
// XMASK: the LeakyReLU mask is recomputed from x with the forward pass's own scale/shift (same expression, same
// rounding) instead of being read back from y: one tensor pass less.
// WRITE_DZ: the masked gradient dz = dout * act'(.) is also written out (dzout may be dout itself: every element is read
// and written by the same thread), and the sums are those of the value as stored.  The apply pass then needs neither y nor
// the mask, and a residual BatchNorm's skip gradient IS that tensor (see vg_bn_bwd_reduce_dz).
template <typename T, bool XMASK, bool WRITE_DZ, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* dout, const T* __restrict__ y,
                                                            const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, double* __restrict__ sums,
                                                            int64_t rows, int cs, int c, int cpad, int act_rt, int det,
                                                            T* dzout) {
  constexpr int PIECE = Elt<T>::PIECE;
  const int act = ACT >= 0 ? ACT : act_rt;       // see bn_act_fwd_kernel
  __shared__ float red[256 * PIECE * 2];
  const int ppr = cs / PIECE;                        // pieces per row (<= 256 guaranteed by launcher)
  const int rpb = 256 / ppr;                         // rows handled per block pass
  const int pc = threadIdx.x % ppr, rr = threadIdx.x / ppr;
  float a1[PIECE], a2[PIECE], mu[PIECE], is[PIECE], sc[PIECE], sh[PIECE];
#pragma unroll
  for (int e = 0; e < PIECE; ++e) { a1[e] = 0.f; a2[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; }
  load_coef<PIECE>(mean, pc * PIECE, c, mu);
  load_coef<PIECE>(invstd, pc * PIECE, c, is);
  if (XMASK) {
    load_coef<PIECE>(scale, pc * PIECE, c, sc);
    load_coef<PIECE>(shift, pc * PIECE, c, sh);
  }
  double d1[PIECE], d2[PIECE];
#pragma unroll
  for (int e = 0; e < PIECE; ++e) { d1[e] = 0.0; d2[e] = 0.0; }
  if (rr < rpb) {
    int cnt = 0;
    auto take = [&](const Piece<T>& g, const Piece<T>& o, const Piece<T>& v, size_t off) {
      Piece<T> m;
#pragma unroll
      for (int e = 0; e < PIECE; ++e) {
        float dz = g.get(e);
        const float yv = XMASK ? v.get(e) * sc[e] + sh[e] : o.get(e);
        if (act == VG_ACT_LRELU && !(yv > 0.f)) dz *= VG_LRELU_SLOPE;
        if (WRITE_DZ) { m.set(e, dz); dz = m.get(e); }
        const float xh = (v.get(e) - mu[e]) * is[e];
        a1[e] += dz;
        a2[e] += dz * xh;
      }
      if (WRITE_DZ) *(decltype(m.v)*)(dzout + off) = m.v;
      if (++cnt == 32) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) { d1[e] += a1[e]; d2[e] += a2[e]; a1[e] = 0.f; a2[e] = 0.f; }
        cnt = 0;
      }
    };
    constexpr int U = 4;                                 // rows in flight per thread
    const int64_t stride = rpb;                          // a block walks its own contiguous chunk of rows
    const int64_t chunk = (((rows + gridDim.x - 1) / gridDim.x + rpb * U - 1) / (rpb * U)) * (rpb * U);
    const int64_t rend = min(rows, (int64_t)(blockIdx.x + 1) * chunk);
    int64_t r = (int64_t)blockIdx.x * chunk + rr;
    for (; r + (U - 1) * stride < rend; r += U * stride) {
      Piece<T> g[U], o[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t off = (size_t)(r + u * stride) * cs + pc * PIECE;
        g[u].v = *(const decltype(g[u].v)*)(dout + off);
        if (!XMASK) o[u].v = *(const decltype(o[u].v)*)(y + off);
        v[u].v = *(const decltype(v[u].v)*)(x + off);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) take(g[u], o[u], v[u], (size_t)(r + u * stride) * cs + pc * PIECE);
    }
    for (; r < rend; r += stride) {
      const size_t off = (size_t)r * cs + pc * PIECE;
      Piece<T> g, o, v;
      g.v = *(const decltype(g.v)*)(dout + off);
      if (!XMASK) o.v = *(const decltype(o.v)*)(y + off);
      v.v = *(const decltype(v.v)*)(x + off);
      take(g, o, v, off);
    }
  }
  // combine the rpb row-groups of this block through LDS, then one f64 atomic per channel per block
  const int width = ppr * PIECE;
  if (rr < rpb) {
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
      red[(rr * width + pc * PIECE + e) * 2 + 0] = (float)(d1[e] + (double)a1[e]);
      red[(rr * width + pc * PIECE + e) * 2 + 1] = (float)(d2[e] + (double)a2[e]);
    }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < rpb; ++r) { t1 += red[(r * width + ch) * 2]; t2 += red[(r * width + ch) * 2 + 1]; }
    vg_stats_add(sums, cpad, 0, ch, blockIdx.x, t1, det);
    vg_stats_add(sums, cpad, 1, ch, blockIdx.x, t2, det);
  }
}

static int bn_bwd_reduce_launch(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                                const float* scale, const float* shift, double* sums, int64_t rows, int cs, int c, int cpad,
                                int act, int dtype, void* dz, void* stream) {
  VG_REQUIRE(dout && x && mean && invstd && sums && rows > 0 && cs >= c && cpad >= c, "bn_bwd_reduce: bad arguments");
  VG_REQUIRE(y || (scale && shift), "bn_bwd_reduce: need y, or scale and shift to recompute the activation mask from x");
  const bool xmask = y == nullptr && act == VG_ACT_LRELU;
  VG_REQUIRE(y || act == VG_ACT_LRELU || act == VG_ACT_NONE, "bn_bwd_reduce: y is required for this activation");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0 && cs / piece <= 256, "bn_bwd_reduce: channel stride %d unsupported", cs);
  const int rpb = 256 / (cs / piece);
  static const int red_cap = [] { const char* e = getenv("VG_BN_RED_CAP"); return e ? atoi(e) : 2048; }();
  static const int red_rpt = [] { const char* e = getenv("VG_BN_RED_RPT"); return e ? atoi(e) : 16; }();
  int64_t blocks = vg_ceil_div(rows, (int64_t)rpb * red_rpt);
  if (blocks > red_cap) blocks = red_cap;
  hipStream_t st = (hipStream_t)stream;
  if (y == nullptr && !xmask) y = x;    // no activation: the value is never looked at
  const int det = vg_det_mode();
  dim3 g((unsigned)blocks), b(256);
#define VG_BN_REDUCE(T, XM, DZ, A)                                                                                            \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, XM, DZ, A>), g, b, 0, st, (const T*)dout, (const T*)y, (const T*)x, mean, invstd, \
                     scale, shift, sums, rows, cs, c, cpad, act, det, (T*)dz)
#define VG_BN_REDUCE_A(T, XM, DZ)                                                                                             \
  do { if (act == VG_ACT_LRELU) VG_BN_REDUCE(T, XM, DZ, VG_ACT_LRELU); else VG_BN_REDUCE(T, XM, DZ, -1); } while (0)
#define VG_BN_REDUCE_T(T)                                                                                                     \
  do {                                                                                                                        \
    if (dz != nullptr) { if (xmask) VG_BN_REDUCE_A(T, true, true); else VG_BN_REDUCE_A(T, false, true); }                     \
    else { if (xmask) VG_BN_REDUCE_A(T, true, false); else VG_BN_REDUCE_A(T, false, false); }                                 \
  } while (0)
  if (dtype == VG_BF16) VG_BN_REDUCE_T(bf16_t); else VG_BN_REDUCE_T(float);
#undef VG_BN_REDUCE_T
#undef VG_BN_REDUCE_A
#undef VG_BN_REDUCE
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_bn_bwd_reduce(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                                const float* scale, const float* shift, double* sums, int64_t rows, int cs, int c, int cpad,
                                int act, int dtype, void* stream) {
  return bn_bwd_reduce_launch(dout, y, x, mean, invstd, scale, shift, sums, rows, cs, c, cpad, act, dtype, nullptr, stream);
}
extern "C" int vg_bn_bwd_reduce_dz(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, double* sums, void* dz, int64_t rows, int cs, int c,
                                   int cpad, int act, int dtype, void* stream) {
  VG_REQUIRE(dz != nullptr, "bn_bwd_reduce_dz: dz is required");
  return bn_bwd_reduce_launch(dout, y, x, mean, invstd, scale, shift, sums, rows, cs, c, cpad, act, dtype, dz, stream);
}

// ---------------------------------------------------------------------------------------------
// backward pass 2
// ---------------------------------------------------------------------------------------------
__global__ void bn_bwd_coef_kernel(const double* __restrict__ sums, int c, int cpad, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ invstd,
                                   const float* __restrict__ mean,
                                   float* __restrict__ coef /* [3][cpad]: dx = ca*dz - cb - cx*x */,
                                   float* dgamma, float* dbeta, int accumulate, float dgb_scale, float* dxsum, int det) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float gm = gamma ? gamma[ch] : 1.f, is = invstd[ch], mu = mean[ch];     // requested before the sums are used
  const float dg0 = (dgamma && accumulate) ? dgamma[ch] : 0.f, db0 = (dbeta && accumulate) ? dbeta[ch] : 0.f;
  const float dxs0 = dxsum ? dxsum[ch] : 0.f;
  const double s1 = vg_stats_read(sums, cpad, 0, ch, det), s2 = vg_stats_read(sums, cpad, 1, ch, det);
  // d = a*(dz - m1 - xh*m2), xh = (x-mu)*is  ==  a*dz - (a*m1 - a*m2*is*mu) - (a*m2*is)*x
  const float a = gm * is, m1 = (float)(s1 / count), m2 = (float)(s2 / count);
  const float cx = a * m2 * is;
  coef[ch] = a;
  coef[cpad + ch] = a * m1 - cx * mu;
  coef[2 * cpad + ch] = cx;
  const float g2 = (float)(s2 * (double)dgb_scale), g1 = (float)(s1 * (double)dgb_scale);
  if (dgamma) dgamma[ch] = dg0 + g2;
  if (dbeta) dbeta[ch] = db0 + g1;
  // Per-channel sum of dx (the bias gradient of the conv in front of this BatchNorm) in closed form, when the sums are
  // those of the rows this rank holds (count = rows):  sum dx = ca*S1 - rows*cb - cx*Sx  with Sx = rows*mean collapses to
  // a*(S1 - rows*m1).  Mathematically zero; what is left is the rounding of m1 to float, i.e. the same kind of noise the
  // reference's autograd leaves there, but without a pass over dx, without atomics, and reproducible.
  if (dxsum) dxsum[ch] = dxs0 + (float)((double)a * (s1 - count * (double)m1));
}

template <typename T, bool HAS_DRES, bool XMASK, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                           const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ coef,
                                                           T* __restrict__ dx, T* __restrict__ dres,
                                                           float* __restrict__ dxsum, int64_t rows,
                                                           int ppr, int c, int cpad, int act_rt) {
  constexpr int PIECE = Elt<T>::PIECE;
  const int act = ACT >= 0 ? ACT : act_rt;       // see bn_act_fwd_kernel
  __shared__ float red[256 * PIECE];
  const int rpb = 256 / ppr;
  const int pc = threadIdx.x % ppr, rr = threadIdx.x / ppr;
  if (rr >= rpb && dxsum == nullptr) return;
  float acc[PIECE];
#pragma unroll
  for (int e = 0; e < PIECE; ++e) acc[e] = 0.f;
  float ca[PIECE], cb[PIECE], cx[PIECE], sc[PIECE], sh[PIECE];     // dx = ca*dz - cb - cx*x (bn_bwd_coef_kernel)
#pragma unroll
  for (int e = 0; e < PIECE; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
  load_coef<PIECE>(coef, pc * PIECE, c, ca);
  load_coef<PIECE>(coef + cpad, pc * PIECE, c, cb);
  load_coef<PIECE>(coef + 2 * cpad, pc * PIECE, c, cx);
  if (XMASK) {
    load_coef<PIECE>(scale, pc * PIECE, c, sc);
    load_coef<PIECE>(shift, pc * PIECE, c, sh);
  }
  const size_t cs = (size_t)ppr * PIECE;
  auto finish = [&](const Piece<T>& g, const Piece<T>& o, const Piece<T>& v, size_t off) {
    Piece<T> rx, rd;
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
      float dz = g.get(e);
      const float yv = XMASK ? v.get(e) * sc[e] + sh[e] : o.get(e);
      if (act == VG_ACT_LRELU && !(yv > 0.f)) dz *= VG_LRELU_SLOPE;
      if (pc * PIECE + e >= c) dz = 0.f;
      rx.set(e, ca[e] * dz - cb[e] - cx[e] * v.get(e));
      acc[e] += rx.get(e);            // the value as stored (the producing conv's bias gradient sums these)
      if (HAS_DRES) rd.set(e, dz);
    }
    *(decltype(rx.v)*)(dx + off) = rx.v;
    if (HAS_DRES) *(decltype(rd.v)*)(dres + off) = rd.v;
  };
  if (rr < rpb) {
    constexpr int U = 2;                                   // rows in flight per thread (4: -3 % at 64 channels, +5..12 % at 256 / 512)
    const int64_t stride = rpb;                            // a block walks its own contiguous chunk of rows
    const int64_t chunk = (((rows + gridDim.x - 1) / gridDim.x + rpb * U - 1) / (rpb * U)) * (rpb * U);
    const int64_t rend = min(rows, (int64_t)(blockIdx.x + 1) * chunk);
    int64_t r = (int64_t)blockIdx.x * chunk + rr;
    for (; r + (U - 1) * stride < rend; r += U * stride) {
      Piece<T> g[U], o[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t off = (size_t)(r + u * stride) * cs + pc * PIECE;
        g[u].v = *(const decltype(g[u].v)*)(dout + off);
        if (!XMASK) o[u].v = *(const decltype(o[u].v)*)(y + off);
        v[u].v = *(const decltype(v[u].v)*)(x + off);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) finish(g[u], o[u], v[u], (size_t)(r + u * stride) * cs + pc * PIECE);
    }
    for (; r < rend; r += stride) {
      const size_t off = (size_t)r * cs + pc * PIECE;
      Piece<T> g, o, v;
      g.v = *(const decltype(g.v)*)(dout + off);
      if (!XMASK) o.v = *(const decltype(o.v)*)(y + off);
      v.v = *(const decltype(v.v)*)(x + off);
      finish(g, o, v, off);
    }
  }
  if (dxsum != nullptr) {     // block-uniform: per-channel sum of dx over this block's rows, one f32 atomic per channel
    const int width = ppr * PIECE;
    if (rr < rpb) {
#pragma unroll
      for (int e = 0; e < PIECE; ++e) red[rr * width + pc * PIECE + e] = acc[e];
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      float t = 0.f;
      for (int r2 = 0; r2 < rpb; ++r2) t += red[r2 * width + ch];
      unsafeAtomicAdd(&dxsum[ch], t);
    }
  }
}

extern "C" int vg_bn_bwd_apply(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                               const float* scale, const float* shift, const float* gamma, const double* sums, void* dx,
                               void* dres, float* dgamma, float* dbeta, int accumulate, float* dxsum, int64_t rows,
                               int64_t count, float dgb_scale, int cs, int c, int cpad, int act, int dtype, void* stream) {
  VG_REQUIRE(dout && x && mean && invstd && sums && dx && rows > 0 && cs >= c && cpad >= c, "bn_bwd_apply: bad arguments");
  VG_REQUIRE(y || (scale && shift), "bn_bwd_apply: need y, or scale and shift to recompute the activation mask from x");
  VG_REQUIRE(y || act == VG_ACT_LRELU || act == VG_ACT_NONE, "bn_bwd_apply: y is required for this activation");
  // no activation and no y: the kernel variant without the y load (its recomputed mask value is never looked at)
  const bool xmask = y == nullptr && (act == VG_ACT_LRELU || (scale && shift));
  if (y == nullptr && !xmask) y = x;
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0 && cs / piece <= 256, "bn_bwd_apply: channel stride %d unsupported", cs);
  hipStream_t st = (hipStream_t)stream;
  // coefficient scratch lives right behind the f64 sums (caller allocates [REPLICAS*2*cpad] f64 + 3*cpad f32)
  float* coef = (float*)(sums + (size_t)VG_STATS_REPLICAS * 2 * cpad);
  // local statistics (count = rows): the bias gradient comes out of the coefficient kernel in closed form; with
  // global-batch statistics (count > rows) the rank's own sum of dx is not zero and the apply kernel adds it up.
  const bool closed = dxsum != nullptr && (count <= 0 || count == rows);
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((c + 127) / 128), dim3(128), 0, st, sums, c, cpad, (double)(count > 0 ? count : rows), gamma, invstd, mean, coef, dgamma, dbeta, accumulate, dgb_scale, closed ? dxsum : (float*)nullptr, vg_det_mode());
  VG_LAUNCH_CHECK();
  if (closed) dxsum = nullptr;
  const int ppr = cs / piece;
  const int64_t np = rows;
  // Grid: many short blocks.  Round 2 ran <= 2048 blocks of >= 16 rows per thread (sized for the per-block bias-sum atomics
  // that most launches no longer have); 16384 blocks of >= 4 rows are 3-10 % faster per launch at 64-256 channels, as a plain
  // copy is (tests/micro/stream_bw.hip), and 0.3 ms per step (scripts/bn_blocks_step_ab.sh).  VG_BN_*_CAP / _RPT: A/B only.
  static const int apply_cap = [] { const char* e = getenv("VG_BN_APPLY_CAP"); return e ? atoi(e) : 16384; }();
  static const int apply_rpt = [] { const char* e = getenv("VG_BN_APPLY_RPT"); return e ? atoi(e) : 4; }();
  int nblocks = row_blocks(rows, 256 / ppr, apply_rpt);               // >= 16 rows per thread: the coefficient preamble amortises
  if (nblocks > apply_cap) nblocks = apply_cap;
  if (dxsum != nullptr) {                                      // every block ends with c same-address f32 atomics
    const int cap = 262144 / c < 256 ? 256 : 262144 / c;        // <= 256 K atomics per launch (c = 512: 56 -> 48 us)
    if (nblocks > cap) nblocks = cap;
  }
  dim3 g(nblocks), b(256);
#define VG_BN_APPLY(T, DR, XM, A)                                                                                   \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T, DR, XM, A>), g, b, 0, st, (const T*)dout, (const T*)y, (const T*)x, mean, invstd, \
                     scale, shift, coef, (T*)dx, (T*)dres, dxsum, np, ppr, c, cpad, act)
#define VG_BN_APPLY_A(T, DR, XM)                                                                                    \
  do { if (act == VG_ACT_LRELU) VG_BN_APPLY(T, DR, XM, VG_ACT_LRELU); else VG_BN_APPLY(T, DR, XM, -1); } while (0)
#define VG_BN_APPLY_T(T)                                                                                            \
  do {                                                                                                              \
    if (dres) { if (xmask) VG_BN_APPLY_A(T, true, true); else VG_BN_APPLY_A(T, true, false); }                      \
    else { if (xmask) VG_BN_APPLY_A(T, false, true); else VG_BN_APPLY_A(T, false, false); }                         \
  } while (0)
  if (dtype == VG_BF16) VG_BN_APPLY_T(bf16_t); else VG_BN_APPLY_T(float);
#undef VG_BN_APPLY_T
#undef VG_BN_APPLY_A
#undef VG_BN_APPLY
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ y, T* __restrict__ dx,
                                                      int64_t npieces, int act) {
  constexpr int PIECE = Elt<T>::PIECE;
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < npieces; q += (int64_t)gridDim.x * 256) {
    Piece<T> g, o, r;
    g.v = *(const decltype(g.v)*)(dout + q * PIECE);
    o.v = *(const decltype(o.v)*)(y + q * PIECE);
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
      float d = g.get(e), yv = o.get(e);
      if (act == VG_ACT_LRELU) d = yv > 0.f ? d : d * VG_LRELU_SLOPE;
      else if (act == VG_ACT_SIGMOID) d = d * yv * (1.f - yv);
      r.set(e, d);
    }
    *(decltype(r.v)*)(dx + q * PIECE) = r.v;
  }
}

extern "C" int vg_act_bwd(const void* dout, const void* y, void* dx, int64_t n, int act, int dtype, void* stream) {
  VG_REQUIRE(dout && y && dx && n > 0, "act_bwd: bad arguments");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(n % piece == 0, "act_bwd: element count %lld not a multiple of %d", (long long)n, piece);
  const int64_t np = n / piece;
  if (dtype == VG_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(ew_blocks(np)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)y, (bf16_t*)dx, np, act);
  else hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_blocks(np)), dim3(256), 0, (hipStream_t)stream, (const float*)dout, (const float*)y, (float*)dx, np, act);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ a, const T* __restrict__ b, int64_t npieces) {
  constexpr int PIECE = Elt<T>::PIECE;
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < npieces; q += (int64_t)gridDim.x * 256) {
    Piece<T> x, y2, r;
    x.v = *(const decltype(x.v)*)(a + q * PIECE);
    y2.v = *(const decltype(y2.v)*)(b + q * PIECE);
#pragma unroll
    for (int e = 0; e < PIECE; ++e) r.set(e, x.get(e) + y2.get(e));
    *(decltype(r.v)*)(a + q * PIECE) = r.v;
  }
}

extern "C" int vg_add_inplace(void* a, const void* b, int64_t n, int dtype, void* stream) {
  VG_REQUIRE(a && b && n > 0, "add_inplace: bad arguments");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(n % piece == 0, "add_inplace: element count not a multiple of %d", piece);
  const int64_t np = n / piece;
  if (dtype == VG_BF16) hipLaunchKernelGGL(add_inplace_kernel<bf16_t>, dim3(ew_blocks(np)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)a, (const bf16_t*)b, np);
  else hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(ew_blocks(np)), dim3(256), 0, (hipStream_t)stream, (float*)a, (const float*)b, np);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// per-channel sum / sum of squares of an arbitrary NHWC tensor (BatchNorm that does not follow a
// conv, e.g. beta_vae_gans/model.py:117 after the Linear projection).  Same layout as the conv stats.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, int64_t rows,
                                                            int cs, int c, int cpad, int det) {
  constexpr int PIECE = Elt<T>::PIECE;
  __shared__ float red[256 * PIECE * 2];
  const int ppr = cs / PIECE, rpb = 256 / ppr;
  const int pc = threadIdx.x % ppr, rr = threadIdx.x / ppr;
  float a1[PIECE], a2[PIECE];
  double d1[PIECE], d2[PIECE];
#pragma unroll
  for (int e = 0; e < PIECE; ++e) { a1[e] = 0.f; a2[e] = 0.f; d1[e] = 0.0; d2[e] = 0.0; }
  if (rr < rpb) {
    int cnt = 0;
    for (int64_t r = (int64_t)blockIdx.x * rpb + rr; r < rows; r += (int64_t)gridDim.x * rpb) {
      Piece<T> v;
      v.v = *(const decltype(v.v)*)(x + (size_t)r * cs + pc * PIECE);
#pragma unroll
      for (int e = 0; e < PIECE; ++e) { const float f = v.get(e); a1[e] += f; a2[e] += f * f; }
      if (++cnt == 32) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) { d1[e] += a1[e]; d2[e] += a2[e]; a1[e] = 0.f; a2[e] = 0.f; }
        cnt = 0;
      }
    }
  }
  const int width = ppr * PIECE;
  if (rr < rpb) {
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
      red[(rr * width + pc * PIECE + e) * 2 + 0] = (float)(d1[e] + (double)a1[e]);
      red[(rr * width + pc * PIECE + e) * 2 + 1] = (float)(d2[e] + (double)a2[e]);
    }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < rpb; ++r) { t1 += red[(r * width + ch) * 2]; t2 += red[(r * width + ch) * 2 + 1]; }
    vg_stats_add(stats, cpad, 0, ch, blockIdx.x, t1, det);
    vg_stats_add(stats, cpad, 1, ch, blockIdx.x, t2, det);
  }
}

extern "C" int vg_channel_stats(const void* x, double* stats, int64_t rows, int cs, int c, int cpad, int dtype, void* stream) {
  VG_REQUIRE(x && stats && rows > 0 && cs >= c && cpad >= c, "channel_stats: bad arguments");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0 && cs / piece <= 256, "channel_stats: channel stride %d unsupported", cs);
  const int rpb = 256 / (cs / piece);
  static const int red_cap = [] { const char* e = getenv("VG_BN_RED_CAP"); return e ? atoi(e) : 2048; }();
  static const int red_rpt = [] { const char* e = getenv("VG_BN_RED_RPT"); return e ? atoi(e) : 16; }();
  int64_t blocks = vg_ceil_div(rows, (int64_t)rpb * red_rpt);
  if (blocks > red_cap) blocks = red_cap;
  if (dtype == VG_BF16) hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, stats, rows, cs, c, cpad, vg_det_mode());
  else hipLaunchKernelGGL(channel_stats_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, stats, rows, cs, c, cpad, vg_det_mode());
  VG_LAUNCH_CHECK();
  return VG_OK;
}


// ---------------------------------------------------------------------------------------------
// Statistics exchange for data parallel runs with global-batch statistics (train.DataParallel(bn_sync='global')): what
// has to travel between the ranks is ONE value per accumulator (sum x, sum x^2 per channel: SURVEY.md section 5), not the
// 32 replicas / 22 fixed-point bins it is spread over on the device.  vg_stats_collapse reads every accumulator the way
// the finalize kernels do (fixed order) into a dense f64 vector; after the all-reduce vg_stats_load puts the vector
// back: plain mode = replica 0, the others cleared; reproducible mode = cut exactly into the fixed-point bins (a double
// is at most three f32 pieces), so everything downstream reads the accumulator as before.  A non-finite value travels
// as NaN and raises the marker slot again.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_collapse_kernel(const double* __restrict__ stats, int nslots, double* __restrict__ vec, int det) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nslots) vec[i] = vg_acc_read(stats + i, (size_t)nslots, det);
}
__global__ __launch_bounds__(256) void stats_load_kernel(const double* __restrict__ vec, int nslots, double* __restrict__ stats, int det) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nslots) return;
  const double v = vec[i];
#pragma unroll
  for (int r = 0; r < VG_STATS_REPLICAS; ++r) stats[(size_t)r * nslots + i] = 0.0;      // all-zero bits in both formats
  if (!det) stats[i] = v;
  else vg_acc_add(stats + i, (size_t)nslots, 0u, v, det);       // this thread owns accumulator i: its own stores are ordered before its atomics
}
extern "C" int vg_stats_collapse(const double* stats, int nslots, double* vec, void* stream) {
  VG_REQUIRE(stats && vec && nslots > 0, "stats_collapse: bad arguments");
  hipLaunchKernelGGL(stats_collapse_kernel, dim3((nslots + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, nslots, vec, vg_det_mode());
  VG_LAUNCH_CHECK();
  return VG_OK;
}
extern "C" int vg_stats_load(const double* vec, int nslots, double* stats, void* stream) {
  VG_REQUIRE(stats && vec && nslots > 0, "stats_load: bad arguments");
  hipLaunchKernelGGL(stats_load_kernel, dim3((nslots + 255) / 256), dim3(256), 0, (hipStream_t)stream, vec, nslots, stats, vg_det_mode());
  VG_LAUNCH_CHECK();
  return VG_OK;
}
