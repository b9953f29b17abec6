// Spatial self-attention (hierarchical_vae_gan/model.py:24-47) on the bf16 MFMA, for the bf16 compute mode.
//
// Same flash-style decomposition and the same register choreography as attn.hip (the softmax axis sits on the
// accumulator's REGISTER index, so max / sum are lane-local and a probability tile feeds the next product straight
// from the accumulator registers), but every product is a v_mfma_f32_32x32x16_bf16: the 16 probabilities a lane
// holds for its query are exactly the eight k-slots of two MFMAs, in the key order rho(t,h), and the other operand
// ([key][channel] rows in LDS) is brought in with transposed LDS reads (ds_read_b64_tr_b16) whose row addresses
// follow that same key order.  q, k, v arrive as the bf16 NHWC tensors the 1x1 convs wrote (their products are exact
// in f32, so the scores equal the f32 kernel's up to summation order); P and dS are rounded to bf16 once.
// 16x less matrix-pipe time than the exact-f32 MFMA (32 cycles per 16-deep product instead of 64 per 2-deep).
#include "common.h"

__device__ __forceinline__ int rho(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 cvt8(const f32x16& v, int o) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[o + i];
  return r;
}

// Operand fragment whose 32 rows (lane&31) are COLUMNS col0..col0+31 of a row-major [row][col] bf16 LDS tile and whose
// k-slots (h, 0..3 / 4..7) are the tile rows row_lo.. / row_hi.. (+0..3): two transposed 8-byte reads.
typedef __attribute__((address_space(3))) bf16x4* tr_ptr_t;
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int stride, int row_lo, int row_hi, int col0, int lane) {
  const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
  const int cterm = (col0 + 16 * g16 + 4 * tp) * 2;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(tile + (row_lo + tq) * stride + cterm));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(tile + (row_hi + tq) * stride + cterm));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int KSTR = 64 + 16;     // LDS row stride of a [32][32] bf16 q/k tile (bytes)

// stage a [32][32] bf16 tile (64-byte rows) and a [32][C] bf16 tile from global memory
template <int C>
__device__ __forceinline__ void stage_qk(unsigned char* dst, const bf16_t* src, int tid) {
  if (tid < 128) {
    const int r = tid >> 2, pc = tid & 3;
    *(u32x4*)(dst + r * KSTR + pc * 16) = *(const u32x4*)(src + (size_t)r * 32 + pc * 8);
  }
}
template <int C>
__device__ __forceinline__ void stage_rows_bf16(unsigned char* dst, const bf16_t* src, int tid) {
  constexpr int VS = C * 2 + 64, PPR = C / 8;
  for (int e = tid; e < 32 * PPR; e += 256) {
    const int r = e / PPR, pc = e - r * PPR;
    *(u32x4*)(dst + r * VS + pc * 16) = *(const u32x4*)(src + (size_t)r * C + pc * 8);
  }
}
template <int C>
__device__ __forceinline__ void stage_rows_f32(unsigned char* dst, const float* src, int tid) {
  constexpr int VS = C * 2 + 64, PPR = C / 4;
  for (int e = tid; e < 32 * PPR; e += 256) {
    const int r = e / PPR, pc = e - r * PPR;
    const f32x4 v = *(const f32x4*)(src + (size_t)r * C + pc * 4);
    *(bf16x4*)(dst + r * VS + pc * 8) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  }
}

// ---------------------------------------------------------------------------------------------
// forward: wave = 32 queries (lane = query), key tiles of 32; O^T accumulators (rows = channels, columns = queries)
// ---------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, float* __restrict__ o,
                                                            float* __restrict__ lse, int HW) {
  constexpr int C = NCB * 32, VS = C * 2 + 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sk = smem;                 // [32][KSTR]
  unsigned char* sv = smem + 32 * KSTR;     // [32][VS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bf16_t* qb = q + (size_t)b * HW * 32;
  const bf16_t* kb = k + (size_t)b * HW * 32;
  const bf16_t* vb = v + (size_t)b * HW * C;
  const bf16x8 qf0 = *(const bf16x8*)(qb + (size_t)(q0 + l32) * 32 + 8 * h);
  const bf16x8 qf1 = *(const bf16x8*)(qb + (size_t)(q0 + l32) * 32 + 16 + 8 * h);
  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  float m = -__builtin_inff(), l = 0.f;
  for (int kt = 0; kt < HW; kt += 32) {
    __syncthreads();
    stage_qk<C>(sk, kb + (size_t)kt * 32, tid);
    stage_rows_bf16<C>(sv, vb + (size_t)kt * C, tid);
    __syncthreads();
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    st = mma(*(const bf16x8*)(sk + l32 * KSTR + 16 * h), qf0, st);
    st = mma(*(const bf16x8*)(sk + l32 * KSTR + 32 + 16 * h), qf1, st);
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - mn); rs += st[r]; }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mn;
    const bf16x8 p0 = cvt8(st, 0), p1 = cvt8(st, 8);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] *= alpha;
      acc[cb] = mma(tr_frag(sv, VS, 4 * h, 8 + 4 * h, cb * 32, lane), p0, acc[cb]);
      acc[cb] = mma(tr_frag(sv, VS, 16 + 4 * h, 24 + 4 * h, cb * 32, lane), p1, acc[cb]);
    }
  }
  const float inv = 1.f / l;
  float* ob = o + ((size_t)b * HW + q0 + l32) * C;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w = {acc[cb][4 * g] * inv, acc[cb][4 * g + 1] * inv, acc[cb][4 * g + 2] * inv, acc[cb][4 * g + 3] * inv};
      *(f32x4*)(ob + cb * 32 + 8 * g + 4 * h) = w;
    }
  if (h == 0) lse[(size_t)b * HW + q0 + l32] = m + __logf(l);
}

// ---------------------------------------------------------------------------------------------
// backward (dQ): wave = 32 queries (lane = query), loop over key tiles
//   S^T = K Q^T ; P = exp(S^T - lse_q) ; dP^T = V dO^T ; dS = P (dP^T - delta_q) ; dQ^T += K^T dS
// ---------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                               const bf16_t* __restrict__ v, const float* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dq, int HW) {
  constexpr int C = NCB * 32, VS = C * 2 + 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sk = smem;
  unsigned char* sv = smem + 32 * KSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bf16_t* qb = q + (size_t)b * HW * 32;
  const bf16_t* kb = k + (size_t)b * HW * 32;
  const bf16_t* vb = v + (size_t)b * HW * C;
  const size_t qrow = (size_t)b * HW + q0 + l32;
  const bf16x8 qf0 = *(const bf16x8*)(qb + (size_t)(q0 + l32) * 32 + 8 * h);
  const bf16x8 qf1 = *(const bf16x8*)(qb + (size_t)(q0 + l32) * 32 + 16 + 8 * h);
  bf16x8 dof[2 * NCB];             // dO[q][16 j + 8 h .. + 7]
#pragma unroll
  for (int j = 0; j < 2 * NCB; ++j) {
    const f32x4 a = *(const f32x4*)(d_o + qrow * C + 16 * j + 8 * h), c4 = *(const f32x4*)(d_o + qrow * C + 16 * j + 8 * h + 4);
    dof[j] = bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)c4[0], (bf16_t)c4[1], (bf16_t)c4[2], (bf16_t)c4[3]};
  }
  const float my_lse = lse[qrow], my_delta = delta[qrow];
  f32x16 dqa;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqa[r] = 0.f;
  for (int kt = 0; kt < HW; kt += 32) {
    __syncthreads();
    stage_qk<C>(sk, kb + (size_t)kt * 32, tid);
    stage_rows_bf16<C>(sv, vb + (size_t)kt * C, tid);
    __syncthreads();
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
    st = mma(*(const bf16x8*)(sk + l32 * KSTR + 16 * h), qf0, st);
    st = mma(*(const bf16x8*)(sk + l32 * KSTR + 32 + 16 * h), qf1, st);
    // dP^T[key][q] = sum_c V[key][c] dO[q][c]
#pragma unroll
    for (int j = 0; j < 2 * NCB; ++j) dp = mma(*(const bf16x8*)(sv + l32 * VS + (16 * j + 8 * h) * 2), dof[j], dp);
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] - my_lse) * (dp[r] - my_delta);     // dS^T[key][q]
    // dQ^T[d][q] += sum_key K[key][d] dS^T[key][q]
    dqa = mma(tr_frag(sk, KSTR, 4 * h, 8 + 4 * h, 0, lane), cvt8(st, 0), dqa);
    dqa = mma(tr_frag(sk, KSTR, 16 + 4 * h, 24 + 4 * h, 0, lane), cvt8(st, 8), dqa);
  }
  float* out = dq + qrow * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) *(f32x4*)(out + 8 * g + 4 * h) = f32x4{dqa[4 * g], dqa[4 * g + 1], dqa[4 * g + 2], dqa[4 * g + 3]};
}

// ---------------------------------------------------------------------------------------------
// backward (dK, dV): wave = 32 keys (lane = key), loop over query tiles
//   S = Q K^T ; P = exp(S - lse_q) ; dV^T += dO^T P ; dP = dO V^T ; dS = P (dP - delta_q) ; dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(256) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                const bf16_t* __restrict__ v, const float* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dk, float* __restrict__ dv, int HW) {
  constexpr int C = NCB * 32, VS = C * 2 + 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sq = smem;                     // [32][KSTR]  Q tile
  unsigned char* sdo = smem + 32 * KSTR;        // [32][VS]    dO tile (bf16)
  float* sls = (float*)(sdo + 32 * VS);         // [32] lse, [32] delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int k0 = (blockIdx.x * 4 + wave) * 32;
  const bf16_t* qb = q + (size_t)b * HW * 32;
  const bf16_t* kb = k + (size_t)b * HW * 32;
  const size_t krow = (size_t)b * HW + k0 + l32;
  const bf16x8 kf0 = *(const bf16x8*)(kb + (size_t)(k0 + l32) * 32 + 8 * h);
  const bf16x8 kf1 = *(const bf16x8*)(kb + (size_t)(k0 + l32) * 32 + 16 + 8 * h);
  bf16x8 vf[2 * NCB];              // V[key][16 j + 8 h .. + 7]
#pragma unroll
  for (int j = 0; j < 2 * NCB; ++j) vf[j] = *(const bf16x8*)(v + krow * C + 16 * j + 8 * h);
  f32x16 dva[NCB], dka;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dva[cb][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) dka[r] = 0.f;
  for (int qt = 0; qt < HW; qt += 32) {
    __syncthreads();
    stage_qk<C>(sq, qb + (size_t)qt * 32, tid);
    stage_rows_f32<C>(sdo, d_o + ((size_t)b * HW + qt) * C, tid);
    if (tid >= 192 && tid < 224) sls[tid - 192] = lse[(size_t)b * HW + qt + tid - 192];
    else if (tid >= 224) sls[32 + tid - 224] = delta[(size_t)b * HW + qt + tid - 224];
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // S[q][key] (rows = queries, lane = key)
    s = mma(*(const bf16x8*)(sq + l32 * KSTR + 16 * h), kf0, s);
    s = mma(*(const bf16x8*)(sq + l32 * KSTR + 32 + 16 * h), kf1, s);
    // dP[q][key] = sum_c dO[q][c] V[key][c]
#pragma unroll
    for (int j = 0; j < 2 * NCB; ++j) dp = mma(*(const bf16x8*)(sdo + l32 * VS + (16 * j + 8 * h) * 2), vf[j], dp);
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = rho(r, h);
      p[r] = __expf(s[r] - sls[qq]);
      s[r] = p[r] * (dp[r] - sls[32 + qq]);        // dS[q][key]
    }
    const bf16x8 p0 = cvt8(p, 0), p1 = cvt8(p, 8);
    // dV^T[c][key] += sum_q dO[q][c] P[q][key]
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      dva[cb] = mma(tr_frag(sdo, VS, 4 * h, 8 + 4 * h, cb * 32, lane), p0, dva[cb]);
      dva[cb] = mma(tr_frag(sdo, VS, 16 + 4 * h, 24 + 4 * h, cb * 32, lane), p1, dva[cb]);
    }
    // dK^T[d][key] += sum_q Q[q][d] dS[q][key]
    dka = mma(tr_frag(sq, KSTR, 4 * h, 8 + 4 * h, 0, lane), cvt8(s, 0), dka);
    dka = mma(tr_frag(sq, KSTR, 16 + 4 * h, 24 + 4 * h, 0, lane), cvt8(s, 8), dka);
  }
  float* ov = dv + krow * C;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(f32x4*)(ov + cb * 32 + 8 * g + 4 * h) = f32x4{dva[cb][4 * g], dva[cb][4 * g + 1], dva[cb][4 * g + 2], dva[cb][4 * g + 3]};
  float* ok = dk + krow * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) *(f32x4*)(ok + 8 * g + 4 * h) = f32x4{dka[4 * g], dka[4 * g + 1], dka[4 * g + 2], dka[4 * g + 3]};
}

__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                              float* __restrict__ delta, int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x * 4L + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += o[r * C + c] * d_o[r * C + c];
    s = wave_sum(s);
    if (lane == 0) delta[r] = s;
  }
}

static int attn_bf16_check(int b, int hw, int d, int c) {
  VG_REQUIRE(b > 0 && hw > 0 && hw % 128 == 0, "attention(bf16): HW=%d must be a positive multiple of 128", hw);
  VG_REQUIRE(d == 32, "attention(bf16): q/k row stride must be 32 (got %d): channels beyond the width are zero padding", d);
  VG_REQUIRE(c == 128 || c == 256, "attention(bf16): value width %d unsupported (128/256)", c);
  return VG_OK;
}

extern "C" int vg_attn_supported_bf16(int hw, int d, int c) { return hw > 0 && hw % 128 == 0 && d == 32 && (c == 128 || c == 256); }

extern "C" int vg_attn_fwd_bf16(const void* q, const void* k, const void* v, float* o, float* lse, int b, int hw, int d, int c,
                                void* stream) {
  int rc = attn_bf16_check(b, hw, d, c);
  if (rc) return rc;
  VG_REQUIRE(q && k && v && o && lse, "attn_fwd_bf16: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 32 * KSTR + 32 * (c * 2 + 64);
  const dim3 grid(hw / 128, b);
  if (c == 128) hipLaunchKernelGGL(attn_fwd_bf16_kernel<4>, grid, dim3(256), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, o, lse, hw);
  else hipLaunchKernelGGL(attn_fwd_bf16_kernel<8>, grid, dim3(256), lds, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, o, lse, hw);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_attn_bwd_bf16(const void* q, const void* k, const void* v, const float* o, const float* d_o, const float* lse,
                                float* dq, float* dk, float* dv, float* delta_ws, int b, int hw, int d, int c, void* stream) {
  int rc = attn_bf16_check(b, hw, d, c);
  if (rc) return rc;
  VG_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv && delta_ws, "attn_bwd_bf16: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)b * hw;
  int dblocks = (int)vg_ceil_div(rows, 4);
  if (dblocks > 4096) dblocks = 4096;
  hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(dblocks), dim3(256), 0, st, o, d_o, delta_ws, rows, c);
  VG_LAUNCH_CHECK();
  const size_t lds = 32 * KSTR + 32 * (c * 2 + 64) + 64 * sizeof(float);
  const dim3 grid(hw / 128, b);
  const bf16_t *qq = (const bf16_t*)q, *kk = (const bf16_t*)k, *vv = (const bf16_t*)v;
  if (c == 128) {
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<4>, grid, dim3(256), lds, st, qq, kk, vv, d_o, lse, delta_ws, dq, hw);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<4>, grid, dim3(256), lds, st, qq, kk, vv, d_o, lse, delta_ws, dk, dv, hw);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<8>, grid, dim3(256), lds, st, qq, kk, vv, d_o, lse, delta_ws, dq, hw);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<8>, grid, dim3(256), lds, st, qq, kk, vv, d_o, lse, delta_ws, dk, dv, hw);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}
