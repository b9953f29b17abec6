// Spatial self-attention (hierarchical_vae_gan/model.py:24-47): softmax_j(q_i . k_j) without the 1/sqrt(d)
// scale, o_i = sum_j A_ij v_j.  Flash-style: the [HW,HW] attention matrix (64 MiB per image at HW=4096 in
// the reference) never leaves registers.
//
// Round-1 implementation: exact-f32 MFMA (v_mfma_f32_32x32x2_f32) for every product.  The score tile is
// always computed with the softmax axis on the REGISTER index of the accumulator, so max/sum are lane-local
// (+ one shuffle between the two lane halves) and the probability tile feeds the next MFMA as its B operand
// straight from the accumulator registers (no LDS round trip, no cross-lane movement):
//   acc reg t of lane (j = lane&31, h = lane>>5) holds X[row rho(t,h)][col j], rho(t,h) = (t&3) + 8*(t>>2) + 4h;
//   mfma(a = M[rho(t,h)][i], b = X reg t) accumulates  sum_rows M[row][i] * X[row][j].
#include "common.h"

__device__ __forceinline__ int rho(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------
// forward: one wave = 32 queries (lane = query), key tiles of 32; O^T accumulators: rows = channels (registers),
// columns = queries (lanes).  Instantiated per channel-block count so that every accumulator index is static.
// ---------------------------------------------------------------------------------------------
template <int NW, int NCB>
__global__ __launch_bounds__(NW * 64) void attn_fwd_t(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ o,
                                                      float* __restrict__ lse, int HW, int D) {
  constexpr int C = NCB * 32;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LK = D + 1;
  constexpr int LV = C + 1;
  float* sk = sm;
  float* sv = sk + 32 * 33;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const bool wave_active = q0 < HW;
  const float* qb = q + (size_t)b * HW * D;
  const float* kb = k + (size_t)b * HW * D;
  const float* vb = v + (size_t)b * HW * C;
  float qr[16];
  const bool q_ok = wave_active && q0 + l32 < HW;        // ragged HW: the last wave's lanes past the end idle
#pragma unroll
  for (int u = 0; u < 16; ++u) qr[u] = (q_ok && 2 * u + h < D) ? qb[(size_t)(q0 + l32) * D + 2 * u + h] : 0.f;
  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  float m = -__builtin_inff(), l = 0.f;
  for (int kt = 0; kt < HW; kt += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * D; e += NW * 64) { int r = e / D, c = e - r * D; sk[r * LK + c] = kt + r < HW ? kb[(size_t)(kt + r) * D + c] : 0.f; }
    for (int e = tid; e < 32 * C; e += NW * 64) { int r = e / C, c = e - r * C; sv[r * LV + c] = kt + r < HW ? vb[(size_t)(kt + r) * C + c] : 0.f; }
    __syncthreads();
    if (!wave_active) continue;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (2 * u < D) st = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[l32 * LK + 2 * u + h], qr[u], st, 0, 0, 0);
    if (kt + 32 > HW) {                                   // ragged last key tile: keys past the end get zero weight
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + rho(r, h) >= HW) st[r] = -__builtin_inff();
    }
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = expf(st[r] - mn); rs += st[r]; }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] *= alpha;
#pragma unroll
      for (int t = 0; t < 16; ++t)
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[rho(t, h) * LV + cb * 32 + l32], st[t], acc[cb], 0, 0, 0);
    }
  }
  if (!q_ok) return;
  const float inv = 1.f / l;
  float* ob = o + ((size_t)b * HW + q0 + l32) * C;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w = {acc[cb][4 * g] * inv, acc[cb][4 * g + 1] * inv, acc[cb][4 * g + 2] * inv, acc[cb][4 * g + 3] * inv};
      *(f32x4*)(ob + cb * 32 + 8 * g + 4 * h) = w;
    }
  if (h == 0) lse[(size_t)b * HW + q0 + l32] = m + logf(l);
}

// ---------------------------------------------------------------------------------------------
// backward, pass 1: delta[q] = sum_c dO[q][c] * O[q][c]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                         float* __restrict__ delta, int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x * 4L + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += o[r * C + c] * d_o[r * C + c];
    s = wave_sum(s);
    if (lane == 0) delta[r] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// backward, pass 2 (dQ): one wave = 32 queries (lane = query), loop over key tiles
//   S^T = K Q^T ; P = exp(S^T - lse_q) ; dP^T = V dO^T ; dS = P (dP^T - delta_q) ; dQ^T += K^T dS
// ---------------------------------------------------------------------------------------------
template <int NW, int NCB>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_t(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ d_o,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dq, int HW, int D) {
  constexpr int C = NCB * 32, LV = C + 1, LK = 33;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sk = sm;                       // [32][33]   K tile, columns >= D zero
  float* sv = sk + 32 * LK;             // [32][LV]   V tile
  float* sdo = sv + 32 * LV;            // [NW][32][LV] dO rows of each wave's queries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const bool wave_active = q0 < HW;
  const float* qb = q + (size_t)b * HW * D;
  const float* kb = k + (size_t)b * HW * D;
  const float* vb = v + (size_t)b * HW * C;
  float qr[16];
  const bool q_ok = wave_active && q0 + l32 < HW;        // ragged HW: lanes past the end idle
#pragma unroll
  for (int u = 0; u < 16; ++u) qr[u] = (q_ok && 2 * u + h < D) ? qb[(size_t)(q0 + l32) * D + 2 * u + h] : 0.f;
  float* mydo = sdo + wave * 32 * LV;
  if (wave_active)
    for (int e = lane; e < 32 * C; e += 64) { int r = e / C, c = e - r * C; mydo[r * LV + c] = q0 + r < HW ? d_o[((size_t)b * HW + q0 + r) * C + c] : 0.f; }
  const float my_lse = q_ok ? lse[(size_t)b * HW + q0 + l32] : 0.f;
  const float my_delta = q_ok ? delta[(size_t)b * HW + q0 + l32] : 0.f;
  f32x16 dqa;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqa[r] = 0.f;
  for (int kt = 0; kt < HW; kt += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += NW * 64) { int r = e >> 5, c = e & 31; sk[r * LK + c] = (c < D && kt + r < HW) ? kb[(size_t)(kt + r) * D + c] : 0.f; }
    for (int e = tid; e < 32 * C; e += NW * 64) { int r = e / C, c = e - r * C; sv[r * LV + c] = kt + r < HW ? vb[(size_t)(kt + r) * C + c] : 0.f; }
    __syncthreads();
    if (!wave_active) continue;
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (2 * u < D) st = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[l32 * LK + 2 * u + h], qr[u], st, 0, 0, 0);
    // dP^T[key][q] = sum_c V[key][c] dO[q][c]
#pragma unroll 4
    for (int u = 0; u < C / 2; ++u)
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[l32 * LV + 2 * u + h], mydo[l32 * LV + 2 * u + h], dp, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r)                                                         // dS^T[key][q]; keys past a ragged end: 0
      st[r] = kt + rho(r, h) < HW ? expf(st[r] - my_lse) * (dp[r] - my_delta) : 0.f;
    // dQ^T[d][q] += sum_key K[key][d] dS^T[key][q]
#pragma unroll
    for (int t = 0; t < 16; ++t) dqa = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[rho(t, h) * LK + l32], st[t], dqa, 0, 0, 0);
  }
  if (!q_ok) return;
  float* out = dq + ((size_t)b * HW + q0 + l32) * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = rho(r, h);
    if (d < D) out[d] = dqa[r];
  }
}

// ---------------------------------------------------------------------------------------------
// backward, pass 3 (dK, dV): one wave = 32 keys (lane = key), loop over query tiles
//   S = Q K^T ; P = exp(S - lse_q) ; dV^T += dO^T P ; dP = dO V^T ; dS = P (dP - delta_q) ; dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <int NW, int NCB>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_t(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dk, float* __restrict__ dv, int HW, int D) {
  constexpr int C = NCB * 32, LV = C + 1, LK = 33;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sq = sm;                       // [32][33]  Q tile, columns >= D zero
  float* sdo = sq + 32 * LK;            // [32][LV]  dO tile
  float* sls = sdo + 32 * LV;           // [32] lse, [32] delta
  float* svk = sls + 64;                // [NW][32][LV] V rows of each wave's keys
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int k0 = (blockIdx.x * NW + wave) * 32;
  const bool wave_active = k0 < HW;
  const float* qb = q + (size_t)b * HW * D;
  const float* kb = k + (size_t)b * HW * D;
  float kr[16];
  const bool k_ok = wave_active && k0 + l32 < HW;        // ragged HW: lanes (keys) past the end idle
#pragma unroll
  for (int u = 0; u < 16; ++u) kr[u] = (k_ok && 2 * u + h < D) ? kb[(size_t)(k0 + l32) * D + 2 * u + h] : 0.f;
  float* myv = svk + wave * 32 * LV;
  if (wave_active)
    for (int e = lane; e < 32 * C; e += 64) { int r = e / C, c = e - r * C; myv[r * LV + c] = k0 + r < HW ? v[((size_t)b * HW + k0 + r) * C + c] : 0.f; }
  f32x16 dva[NCB], dka;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dva[cb][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) dka[r] = 0.f;
  for (int qt = 0; qt < HW; qt += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += NW * 64) { int r = e >> 5, c = e & 31; sq[r * LK + c] = (c < D && qt + r < HW) ? qb[(size_t)(qt + r) * D + c] : 0.f; }
    for (int e = tid; e < 32 * C; e += NW * 64) { int r = e / C, c = e - r * C; sdo[r * LV + c] = qt + r < HW ? d_o[((size_t)b * HW + qt + r) * C + c] : 0.f; }
    // queries past a ragged end: lse = +inf makes their probability exp(s - lse) exactly 0
    if (tid < 32) sls[tid] = qt + tid < HW ? lse[(size_t)b * HW + qt + tid] : __builtin_inff();
    else if (tid < 64) sls[tid] = qt + tid - 32 < HW ? delta[(size_t)b * HW + qt + tid - 32] : 0.f;
    __syncthreads();
    if (!wave_active) continue;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // S[q][key] = sum_d Q[q][d] K[key][d]   (rows = queries, lane = key)
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (2 * u < D) s = __builtin_amdgcn_mfma_f32_32x32x2f32(sq[l32 * LK + 2 * u + h], kr[u], s, 0, 0, 0);
    // dP[q][key] = sum_c dO[q][c] V[key][c]
#pragma unroll 4
    for (int u = 0; u < C / 2; ++u)
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sdo[l32 * LV + 2 * u + h], myv[l32 * LV + 2 * u + h], dp, 0, 0, 0);
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = rho(r, h);
      p[r] = expf(s[r] - sls[qq]);
      s[r] = p[r] * (dp[r] - sls[32 + qq]);        // dS[q][key]
    }
    // dV^T[c][key] += sum_q dO[q][c] P[q][key]
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        dva[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sdo[rho(t, h) * LV + cb * 32 + l32], p[t], dva[cb], 0, 0, 0);
    // dK^T[d][key] += sum_q Q[q][d] dS[q][key]
#pragma unroll
    for (int t = 0; t < 16; ++t) dka = __builtin_amdgcn_mfma_f32_32x32x2f32(sq[rho(t, h) * LK + l32], s[t], dka, 0, 0, 0);
  }
  if (!k_ok) return;
  float* ov = dv + ((size_t)b * HW + k0 + l32) * C;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w = {dva[cb][4 * g], dva[cb][4 * g + 1], dva[cb][4 * g + 2], dva[cb][4 * g + 3]};
      *(f32x4*)(ov + cb * 32 + 8 * g + 4 * h) = w;
    }
  float* ok = dk + ((size_t)b * HW + k0 + l32) * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = rho(r, h);
    if (d < D) ok[d] = dka[r];
  }
}

// ---------------------------------------------------------------------------------------------
static int attn_check(int b, int hw, int d, int c) {
  VG_REQUIRE(b > 0 && hw > 0, "attention: HW=%d must be positive", hw);
  VG_REQUIRE(d > 0 && d <= 32 && d % 2 == 0, "attention: q/k width %d unsupported (even, <= 32)", d);
  VG_REQUIRE(c == 32 || c == 64 || c == 128 || c == 256, "attention: value width %d unsupported (32/64/128/256)", c);
  return VG_OK;
}

#define ATTN_DISPATCH(KERN, NWV, LDS, GRIDX, ...)                                                            \
  do {                                                                                                        \
    auto kern_ = KERN;                                                                                        \
    VG_HIP(vg_lds_attr((const void*)kern_, 160 * 1024)); \
    hipLaunchKernelGGL(kern_, dim3(GRIDX, b), dim3(NWV * 64), LDS, st, __VA_ARGS__);                          \
  } while (0)

extern "C" int vg_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int b, int hw, int d, int c,
                           void* stream) {
  int rc = attn_check(b, hw, d, c);
  if (rc) return rc;
  VG_REQUIRE(q && k && v && o && lse, "attn_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (32 * 33 + 32 * (c + 1)) * sizeof(float);
  const unsigned gx = (unsigned)vg_ceil_div(hw, 4 * 32);
  switch (c / 32) {
    case 1: ATTN_DISPATCH((attn_fwd_t<4, 1>), 4, lds, gx, q, k, v, o, lse, hw, d); break;
    case 2: ATTN_DISPATCH((attn_fwd_t<4, 2>), 4, lds, gx, q, k, v, o, lse, hw, d); break;
    case 4: ATTN_DISPATCH((attn_fwd_t<4, 4>), 4, lds, gx, q, k, v, o, lse, hw, d); break;
    default: ATTN_DISPATCH((attn_fwd_t<4, 8>), 4, lds, gx, q, k, v, o, lse, hw, d); break;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                           float* dq, float* dk, float* dv, float* delta_ws, int b, int hw, int d, int c, void* stream) {
  int rc = attn_check(b, hw, d, c);
  if (rc) return rc;
  VG_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv && delta_ws, "attn_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)b * hw;
  int dblocks = (int)vg_ceil_div(rows, 4);
  if (dblocks > 4096) dblocks = 4096;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(dblocks), dim3(256), 0, st, o, d_o, delta_ws, rows, c);
  VG_LAUNCH_CHECK();
  const int nw = c <= 128 ? 4 : 2;
  const size_t lds_q = (32 * 33 + 32 * (c + 1) + (size_t)nw * 32 * (c + 1)) * sizeof(float);
  const size_t lds_k = lds_q + 64 * sizeof(float);
  const unsigned gx = (unsigned)vg_ceil_div(hw, nw * 32);
  switch (c / 32) {
    case 1:
      ATTN_DISPATCH((attn_bwd_dq_t<4, 1>), 4, lds_q, gx, q, k, v, d_o, lse, delta_ws, dq, hw, d);
      ATTN_DISPATCH((attn_bwd_dkv_t<4, 1>), 4, lds_k, gx, q, k, v, d_o, lse, delta_ws, dk, dv, hw, d);
      break;
    case 2:
      ATTN_DISPATCH((attn_bwd_dq_t<4, 2>), 4, lds_q, gx, q, k, v, d_o, lse, delta_ws, dq, hw, d);
      ATTN_DISPATCH((attn_bwd_dkv_t<4, 2>), 4, lds_k, gx, q, k, v, d_o, lse, delta_ws, dk, dv, hw, d);
      break;
    case 4:
      ATTN_DISPATCH((attn_bwd_dq_t<4, 4>), 4, lds_q, gx, q, k, v, d_o, lse, delta_ws, dq, hw, d);
      ATTN_DISPATCH((attn_bwd_dkv_t<4, 4>), 4, lds_k, gx, q, k, v, d_o, lse, delta_ws, dk, dv, hw, d);
      break;
    default:
      ATTN_DISPATCH((attn_bwd_dq_t<2, 8>), 2, lds_q, gx, q, k, v, d_o, lse, delta_ws, dq, hw, d);
      ATTN_DISPATCH((attn_bwd_dkv_t<2, 8>), 2, lds_k, gx, q, k, v, d_o, lse, delta_ws, dk, dv, hw, d);
      break;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- y = gamma * o + x (model.py:45) and its backward ----------------------------------------------------
__global__ __launch_bounds__(256) void gamma_residual_kernel(const float* __restrict__ o, const float* __restrict__ x,
                                                             const float* __restrict__ gamma, float* __restrict__ y, int64_t n) {
  const float g = *gamma;
  for (int64_t i = blockIdx.x * 256L + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = g * o[i] + x[i];
}

__global__ __launch_bounds__(256) void gamma_residual_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ o,
                                                                 const float* __restrict__ gamma, float* __restrict__ d_o,
                                                                 double* __restrict__ dgamma, int64_t n) {
  const float g = *gamma;
  double local = 0.0;
  for (int64_t i = blockIdx.x * 256L + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gv = gy[i];
    d_o[i] = g * gv;
    local += (double)(gv * o[i]);
  }
  __shared__ double red[4];
  double w = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(dgamma, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int vg_gamma_residual(const float* o, const float* x, const float* gamma, float* y, int64_t n, void* stream) {
  VG_REQUIRE(o && x && gamma && y && n > 0, "gamma_residual: bad arguments");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gamma_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, o, x, gamma, y, n);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_gamma_residual_bwd(const float* gy, const float* o, const float* gamma, float* d_o, double* dgamma, int64_t n,
                                     void* stream) {
  VG_REQUIRE(gy && o && gamma && d_o && dgamma && n > 0, "gamma_residual_bwd: bad arguments");
  int64_t blocks = (n + 2047) / 2048;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gamma_residual_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gy, o, gamma, d_o, dgamma, n);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- dtype cast between activation storage types (bf16 <-> f32), 16 bytes of the narrower side per thread ----
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, int64_t n8) {
  for (int64_t q = blockIdx.x * 256L + threadIdx.x; q < n8; q += (int64_t)gridDim.x * 256) {
    bf16x8 v = *(const bf16x8*)(s + q * 8);
    f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    *(f32x4*)(d + q * 8) = a;
    *(f32x4*)(d + q * 8 + 4) = b;
  }
}
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n8) {
  for (int64_t q = blockIdx.x * 256L + threadIdx.x; q < n8; q += (int64_t)gridDim.x * 256) {
    f32x4 a = *(const f32x4*)(s + q * 8), b = *(const f32x4*)(s + q * 8 + 4);
    bf16x8 v = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    *(bf16x8*)(d + q * 8) = v;
  }
}

extern "C" int vg_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream) {
  VG_REQUIRE(src && dst && n > 0 && n % 8 == 0 && src_dtype != dst_dtype, "cast: bad arguments");
  int64_t n8 = n / 8, blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (src_dtype == VG_BF16) hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (float*)dst, n8);
  else hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src, (bf16_t*)dst, n8);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
