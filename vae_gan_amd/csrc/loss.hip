// Loss reductions and their gradients, reparameterisation + KL, fused Adam.
// All bandwidth-bound: 16-byte accesses, wavefront shuffle reduction -> one f64 atomic per block.
#include "common.h"

__device__ inline void block_atomic_sum(double local, double* out) {
  __shared__ double red[4];
  double w = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

static int red_blocks(int64_t items) {
  int64_t b = (items + 256 * 8 - 1) / (256 * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- sum (a-b)^2 over valid channels of NHWC tensors -------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sqdiff_sum_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t npieces,
                                                         int ppr, int c, double* __restrict__ out) {
  constexpr int PIECE = Elt<T>::PIECE;
  float local = 0.f;
  double dl = 0.0;
  int cnt = 0;
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < npieces; q += (int64_t)gridDim.x * 256) {
    const int ch0 = (int)(q % ppr) * PIECE;
    if (ch0 >= c) continue;
    Piece<T> x, y;
    x.v = *(const decltype(x.v)*)(a + q * PIECE);
    y.v = *(const decltype(y.v)*)(b + q * PIECE);
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
      const float d = x.get(e) - y.get(e);
      if (ch0 + e < c) local += d * d;
    }
    if (++cnt == 64) { dl += local; local = 0.f; cnt = 0; }
  }
  block_atomic_sum(dl + (double)local, out);
}

extern "C" int vg_sqdiff_sum(const void* a, const void* b, int64_t rows, int cs, int c, int dtype, double* out,
                             void* stream) {
  VG_REQUIRE(a && b && out && rows > 0 && cs >= c, "sqdiff_sum: bad arguments");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0, "sqdiff_sum: channel stride %d unsupported", cs);
  const int ppr = cs / piece;
  const int64_t np = rows * ppr;
  if (dtype == VG_BF16) hipLaunchKernelGGL(sqdiff_sum_kernel<bf16_t>, dim3(red_blocks(np)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, np, ppr, c, out);
  else hipLaunchKernelGGL(sqdiff_sum_kernel<float>, dim3(red_blocks(np)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, np, ppr, c, out);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void sqdiff_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                         const float* __restrict__ g, float coef, int64_t npieces, int ppr,
                                                         int c, T* __restrict__ da) {
  constexpr int PIECE = Elt<T>::PIECE;
  const float s = (g ? *g : 1.f) * coef;
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < npieces; q += (int64_t)gridDim.x * 256) {
    const int ch0 = (int)(q % ppr) * PIECE;
    Piece<T> x, y, o;
    x.v = *(const decltype(x.v)*)(a + q * PIECE);
    y.v = *(const decltype(y.v)*)(b + q * PIECE);
#pragma unroll
    for (int e = 0; e < PIECE; ++e) o.set(e, ch0 + e < c ? s * (x.get(e) - y.get(e)) : 0.f);
    *(decltype(o.v)*)(da + q * PIECE) = o.v;
  }
}

extern "C" int vg_sqdiff_bwd(const void* a, const void* b, const float* g, float coef, int64_t rows, int cs, int c,
                             int dtype, void* da, void* stream) {
  VG_REQUIRE(a && b && da && rows > 0 && cs >= c, "sqdiff_bwd: bad arguments");
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0, "sqdiff_bwd: channel stride %d unsupported", cs);
  const int ppr = cs / piece;
  const int64_t np = rows * ppr;
  int64_t blocks = (np + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == VG_BF16) hipLaunchKernelGGL(sqdiff_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, g, coef, np, ppr, c, (bf16_t*)da);
  else hipLaunchKernelGGL(sqdiff_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, g, coef, np, ppr, c, (float*)da);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- flat f32 reductions: mode 0: x, 1: relu(1-x), 2: relu(1+x) ---------------------------------
__global__ __launch_bounds__(256) void reduce_f32_kernel(const float* __restrict__ x, int64_t n, int mode,
                                                         double* __restrict__ out) {
  double local = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = x[i];
    if (mode == 1) v = fmaxf(1.f - v, 0.f);
    else if (mode == 2) v = fmaxf(1.f + v, 0.f);
    local += v;
  }
  block_atomic_sum(local, out);
}

extern "C" int vg_reduce_f32(const float* x, int64_t n, int mode, double* out, void* stream) {
  VG_REQUIRE(x && out && n > 0 && mode >= 0 && mode <= 2, "reduce_f32: bad arguments");
  hipLaunchKernelGGL(reduce_f32_kernel, dim3(red_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n, mode, out);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void reduce_f32_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             float coef, int64_t n, int mode, float* __restrict__ dx) {
  const float s = (g ? *g : 1.f) * coef;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float d = s;
    if (mode == 1) d = (1.f - x[i] > 0.f) ? -s : 0.f;
    else if (mode == 2) d = (1.f + x[i] > 0.f) ? s : 0.f;
    dx[i] = d;
  }
}

extern "C" int vg_reduce_f32_bwd(const float* x, const float* g, float coef, int64_t n, int mode, float* dx, void* stream) {
  VG_REQUIRE(x && dx && n > 0 && mode >= 0 && mode <= 2, "reduce_f32_bwd: bad arguments");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_f32_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, g, coef, n, mode, dx);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- reparameterisation + KL --------------------------------------------------------------------
// z = mu + eps*exp(0.5*lv);  kl_sum += sum(1 + lv - mu^2 - exp(lv))      (loss = -0.5*mean)
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, int64_t n, float* __restrict__ z,
                                                             double* __restrict__ kl) {
  double local = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float m = mu[i], l = lv[i];
    if (z) z[i] = m + eps[i] * expf(0.5f * l);
    local += (double)(1.f + l - m * m - expf(l));
  }
  if (kl) block_atomic_sum(local, kl);
}

extern "C" int vg_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps, int64_t n, float* z, double* kl,
                                 void* stream) {
  VG_REQUIRE(mu && logvar && n > 0 && (z == nullptr || eps != nullptr), "reparam_kl_fwd: bad arguments");
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(red_blocks(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, n, z, kl);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// dmu = gz + gkl*ckl*(-2 mu);  dlv = gz*eps*0.5*exp(0.5 lv) + gkl*ckl*(1 - exp(lv))
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, const float* __restrict__ gz,
                                                             const float* __restrict__ gkl, float ckl, int64_t n,
                                                             float* __restrict__ dmu, float* __restrict__ dlv) {
  const float gk = gkl ? *gkl * ckl : 0.f;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float m = mu[i], l = lv[i];
    const float g = gz ? gz[i] : 0.f;
    dmu[i] = g + gk * (-2.f * m);
    dlv[i] = (gz ? g * eps[i] * 0.5f * expf(0.5f * l) : 0.f) + gk * (1.f - expf(l));
  }
}

extern "C" int vg_reparam_kl_bwd(const float* mu, const float* logvar, const float* eps, const float* gz, const float* gkl,
                                 float ckl, int64_t n, float* dmu, float* dlogvar, void* stream) {
  VG_REQUIRE(mu && logvar && dmu && dlogvar && n > 0 && (gz == nullptr || eps != nullptr), "reparam_kl_bwd: bad arguments");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, gz, gkl, ckl, n, dmu, dlogvar);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- fused Adam over a flat buffer (torch.optim.Adam arithmetic, eps outside the bias-corrected sqrt) ----
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float lr, float b1, float b2,
                                                   float eps, float bc1, float rsqrt_bc2, float gscale) {
  const float step = lr / bc1;
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    f32x4 pp = *(const f32x4*)(p + q * 4), gg = *(const f32x4*)(g + q * 4);
    f32x4 mm = *(const f32x4*)(m + q * 4), vv = *(const f32x4*)(v + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      mm[e] = mm[e] * b1 + (1.f - b1) * gr;
      vv[e] = vv[e] * b2 + (1.f - b2) * gr * gr;
      const float denom = sqrtf(vv[e]) * rsqrt_bc2 + eps;
      pp[e] = pp[e] - step * (mm[e] / denom);
    }
    *(f32x4*)(p + q * 4) = pp;
    *(f32x4*)(m + q * 4) = mm;
    *(f32x4*)(v + q * 4) = vv;
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gscale;
      const float m1 = m[i] * b1 + (1.f - b1) * gr;
      const float v1 = v[i] * b2 + (1.f - b2) * gr * gr;
      m[i] = m1; v[i] = v1;
      p[i] = p[i] - step * (m1 / (sqrtf(v1) * rsqrt_bc2 + eps));
    }
  }
}

extern "C" int vg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, int step, float grad_scale, void* stream) {
  VG_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  VG_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr, beta1,
                     beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- column sums of a small-row f32 matrix (Linear bias gradients): out[c] = sum_r a[r][c] -------------
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ a, int rows, int64_t cols,
                                                         float* __restrict__ out) {
  for (int64_t c = blockIdx.x * (int64_t)256 + threadIdx.x; c < cols; c += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += a[(size_t)r * cols + c];
    out[c] = s;
  }
}

extern "C" int vg_colsum_f32(const float* a, int rows, int64_t cols, float* out, void* stream) {
  VG_REQUIRE(a && out && rows > 0 && cols > 0, "colsum_f32: bad arguments");
  int64_t blocks = (cols + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, rows, cols, out);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- clamp (torch.clamp(logvar, -20, 20): beta_vae_gans/model.py:91, hierarchical_vae_gan/model.py:168) ----
// forward: y = min(max(x, lo), hi); backward (g != null): y = (lo <= x <= hi) ? g : 0
__global__ __launch_bounds__(256) void clamp_kernel(const float* __restrict__ x, const float* __restrict__ g, float lo, float hi,
                                                    float* __restrict__ y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    y[i] = g ? ((v >= lo && v <= hi) ? g[i] : 0.f) : fminf(fmaxf(v, lo), hi);
  }
}

extern "C" int vg_clamp(const float* x, const float* g, float lo, float hi, float* y, int64_t n, void* stream) {
  VG_REQUIRE(x && y && n > 0 && lo <= hi, "clamp: bad arguments");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, g, lo, hi, y, n);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
