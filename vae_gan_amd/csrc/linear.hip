// Backward of the wide Linear layers (beta-VAE-GAN: fc_mu / fc_logvar map 512*16*16 = 131072.. 524288 features to the
// latent; beta_vae_gans/model.py:126-127, 140-141): y = x W^T + b with a handful of batch rows and a huge weight.
//   dW[o][k] (+)= sum_b gy[b][o] * x[b][k]      (fout x fin floats: 268 MB for 128 x 524288)
//   dx[b][k]   = sum_o gy[b][o] * W[o][k]
// Both are bound by one pass over the [fout][fin] matrix (write dW / read W).  Through the generic pixel-reduction
// weight-gradient kernel (memset + atomics into a packed image + unpack) the weight gradient made four passes over
// 268 MB and took 0.89 ms per layer; here a thread owns two input features k, keeps the batch's x values (or the dx
// accumulators) in registers, and streams dW / W rows with coalesced 8-byte accesses; gy sits in LDS and is read as
// 16-byte broadcasts.  No atomics: the result does not depend on the order blocks run in.
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// BC = batch rows held in registers per pass (a longer batch takes several passes; later ones accumulate)
template <int BC>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           float* __restrict__ dw, int b0, int nb, int fin, int fout,
                                                           int accumulate) {
  extern __shared__ float sgy[];                     // [fout][BC]: gy^T of this pass (zero padded rows past nb)
  for (int i = threadIdx.x; i < fout * BC; i += 256) {
    const int o = i / BC, b = i - o * BC;
    sgy[i] = b < nb ? gy[(size_t)(b0 + b) * fout + o] : 0.f;
  }
  __syncthreads();
  const long k = 2 * ((long)blockIdx.x * 256 + threadIdx.x);
  if (k >= fin) return;
  f32x2 xv[BC];
#pragma unroll
  for (int b = 0; b < BC; ++b) xv[b] = b < nb ? *(const f32x2*)(x + (size_t)(b0 + b) * fin + k) : f32x2{0.f, 0.f};
  for (int o = 0; o < fout; ++o) {
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int b4 = 0; b4 < BC; b4 += 4) {
      const f32x4 g = *(const f32x4*)(sgy + o * BC + b4);         // same address in every lane: a broadcast read
      s += xv[b4] * g[0]; s += xv[b4 + 1] * g[1]; s += xv[b4 + 2] * g[2]; s += xv[b4 + 3] * g[3];
    }
    f32x2* dst = (f32x2*)(dw + (size_t)o * fin + k);
    if (accumulate) s += *dst;
    *dst = s;
  }
}

template <int BC>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           float* __restrict__ dx, int b0, int nb, int fin, int fout) {
  extern __shared__ float sgy[];                     // [fout][BC]
  for (int i = threadIdx.x; i < fout * BC; i += 256) {
    const int o = i / BC, b = i - o * BC;
    sgy[i] = b < nb ? gy[(size_t)(b0 + b) * fout + o] : 0.f;
  }
  __syncthreads();
  const long k = 2 * ((long)blockIdx.x * 256 + threadIdx.x);
  if (k >= fin) return;
  f32x2 acc[BC];
#pragma unroll
  for (int b = 0; b < BC; ++b) acc[b] = f32x2{0.f, 0.f};
  f32x2 wv = *(const f32x2*)(w + k);
  for (int o = 0; o < fout; ++o) {
    const f32x2 wn = o + 1 < fout ? *(const f32x2*)(w + (size_t)(o + 1) * fin + k) : f32x2{0.f, 0.f};   // next row in flight
#pragma unroll
    for (int b4 = 0; b4 < BC; b4 += 4) {
      const f32x4 g = *(const f32x4*)(sgy + o * BC + b4);
      acc[b4] += wv * g[0]; acc[b4 + 1] += wv * g[1]; acc[b4 + 2] += wv * g[2]; acc[b4 + 3] += wv * g[3];
    }
    wv = wn;
  }
#pragma unroll
  for (int b = 0; b < BC; ++b)
    if (b < nb) *(f32x2*)(dx + (size_t)(b0 + b) * fin + k) = acc[b];
}

static const int LIN_BC = 32;

extern "C" int vg_linear_wgrad(const float* x, const float* gy, float* dw, int b, int fin, int fout, int accumulate, void* stream) {
  VG_REQUIRE(x && gy && dw && b > 0 && fin > 0 && fout > 0, "linear_wgrad: bad arguments");
  VG_REQUIRE(fin % 2 == 0, "linear_wgrad: in_features %d must be even", fin);
  const size_t lds = (size_t)fout * LIN_BC * sizeof(float);
  VG_REQUIRE(lds <= 160 * 1024, "linear_wgrad: out_features %d too large for the LDS copy of the output gradient", fout);
  VG_HIP(vg_lds_attr((const void*)linear_wgrad_kernel<LIN_BC>, (int)lds));
  const unsigned blocks = (unsigned)vg_ceil_div(fin / 2, 256);
  for (int b0 = 0; b0 < b; b0 += LIN_BC) {
    const int nb = b - b0 < LIN_BC ? b - b0 : LIN_BC;
    hipLaunchKernelGGL(linear_wgrad_kernel<LIN_BC>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, gy, dw, b0, nb, fin, fout,
                       (accumulate || b0 > 0) ? 1 : 0);
    VG_LAUNCH_CHECK();
  }
  return VG_OK;
}

extern "C" int vg_linear_dgrad(const float* gy, const float* w, float* dx, int b, int fin, int fout, void* stream) {
  VG_REQUIRE(gy && w && dx && b > 0 && fin > 0 && fout > 0, "linear_dgrad: bad arguments");
  VG_REQUIRE(fin % 2 == 0, "linear_dgrad: in_features %d must be even", fin);
  const size_t lds = (size_t)fout * LIN_BC * sizeof(float);
  VG_REQUIRE(lds <= 160 * 1024, "linear_dgrad: out_features %d too large for the LDS copy of the output gradient", fout);
  VG_HIP(vg_lds_attr((const void*)linear_dgrad_kernel<LIN_BC>, (int)lds));
  const unsigned blocks = (unsigned)vg_ceil_div(fin / 2, 256);
  for (int b0 = 0; b0 < b; b0 += LIN_BC) {
    const int nb = b - b0 < LIN_BC ? b - b0 : LIN_BC;
    hipLaunchKernelGGL(linear_dgrad_kernel<LIN_BC>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, gy, w, dx, b0, nb, fin, fout);
    VG_LAUNCH_CHECK();
  }
  return VG_OK;
}
