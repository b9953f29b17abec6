// Vector quantiser kernels (reference: vq_vae_gan/model.py:68-134).
//
// vq_assign: fused  (|z|^2 + |E|^2) - 2 z.E  distance tile on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain) with a running per-lane argmin and one
// wavefront shuffle to merge the two lane halves.  The [rows][K] distance matrix the reference
// materialises (134 MB at B=64) never leaves registers.
#include "common.h"

template <int NW>
__global__ __launch_bounds__(NW * 64) void vq_assign_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            int64_t rows, int K, int D, int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = D + 1;                       // padded row stride: conflict-free 4-byte column reads
  float* sz = sm;                             // [NW*32][LD]
  float* se = sz + NW * 32 * LD;              // [2][32][LD]
  float* sb = se + 2 * 32 * LD;               // [2][32]  |E_k|^2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  constexpr int NT = NW * 64;
  const int64_t row0 = (int64_t)blockIdx.x * (NW * 32);

  // stage the block's z rows
  for (int q = tid; q < NW * 32 * (D / 4); q += NT) {
    int r = q / (D / 4), c4 = q - r * (D / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < rows) v = *(const f32x4*)(z + (size_t)(row0 + r) * D + c4 * 4);
    float* dst = sz + r * LD + c4 * 4;
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
  }
  auto stage_codes = [&](int tile, int buf) {
    for (int q = tid; q < 32 * (D / 4); q += NT) {
      int r = q / (D / 4), c4 = q - r * (D / 4);
      int code = tile * 32 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (code < K) v = *(const f32x4*)(cb + (size_t)code * D + c4 * 4);
      float* dst = se + (buf * 32 + r) * LD + c4 * 4;
      dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
  };
  stage_codes(0, 0);
  __syncthreads();
  // |z_row|^2 for this lane's row (column of the MFMA tile), sequential in k like a plain f32 sum
  const float* zrow = sz + (wave * 32 + l32) * LD;
  float zz = 0.f;
  for (int k = 0; k < D; ++k) zz += zrow[k] * zrow[k];

  float best = __builtin_inff();
  int besti = 0;
  const int ntiles = (K + 31) / 32;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (tid < 32) {   // |E_k|^2 of the current tile
      const float* er = se + (buf * 32 + tid) * LD;
      float s = 0.f;
      for (int k = 0; k < D; ++k) s += er[k] * er[k];
      sb[buf * 32 + tid] = s;
    }
    if (t + 1 < ntiles) stage_codes(t + 1, buf ^ 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ea = se + (buf * 32 + l32) * LD + h;
    const float* zb = zrow + h;
#pragma unroll 8
    for (int s = 0; s < D / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[2 * s], zb[2 * s], acc, 0, 0, 0);
    __syncthreads();   // sb of this tile visible; next tile fully staged
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int code = t * 32 + cl;
      const float dist = (zz + sb[buf * 32 + cl]) - 2.f * acc[r];
      if (code < K && dist < best) { best = dist; besti = code; }
    }
  }
  // merge the two lane halves (codes interleave in groups of 4): smaller distance, then smaller index
  const float ob = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(besti, 32, 64);
  if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  const int64_t row = row0 + wave * 32 + l32;
  if (h == 0 && row < rows) idx[row] = besti;
}

// ---------------------------------------------------------------------------------------------
// vq_assign2: the same arithmetic (same k order, same tie rule, so the same indices bit for bit) laid out for large
// codebooks (K = 8192, D = 256: 275 GFLOP per search at batch 16 x 64 x 64 latents).  The kernel above keeps z and
// the code tile in LDS, reads two scalars per MFMA, and at D = 256 fits one 2-wave block per CU.  Here
//  * a wave's 32 z rows live in REGISTERS for the whole search (D/2 VGPRs: lane (h, row) holds z[row][2s+h]),
//  * the code tile is stored [code][k parity][D/2] so one ds_read_b128 feeds four MFMAs,
//  * |E_k|^2 comes from a one-thread-per-code pre-pass (same sequential sum) instead of 32 lanes per tile,
//  * the next tile's global loads are issued before the MFMA loop and land in LDS after it,
// which leaves 67 KB of LDS per 4-wave block: two blocks per CU, two waves on every SIMD.
// ---------------------------------------------------------------------------------------------
// |z|^2 in k order, one fused multiply-add per element: spelled out so that every kernel that needs the exact search's
// |z|^2 gets the same bits (left to the compiler, `zz += v * v` was contracted in one kernel and not in another)
__device__ __forceinline__ float vq_sq4(float zz, const f32x4 v) {
  zz = __builtin_fmaf(v[0], v[0], zz);
  zz = __builtin_fmaf(v[1], v[1], zz);
  zz = __builtin_fmaf(v[2], v[2], zz);
  return __builtin_fmaf(v[3], v[3], zz);
}

__global__ __launch_bounds__(256) void vq_code_norms_kernel(const float* __restrict__ cb, int K, int D, float* __restrict__ e2) {
  const int code = blockIdx.x * 256 + threadIdx.x;
  if (code >= K) return;
  const float* er = cb + (size_t)code * D;
  float s = 0.f;
  for (int k = 0; k < D; ++k) s += er[k] * er[k];
  e2[code] = s;
}

template <int D>
__global__ __launch_bounds__(256, 2) void vq_assign2_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                         const float* __restrict__ e2, int64_t rows, int K,
                                                         int64_t* __restrict__ idx, const int* __restrict__ guard) {
  constexpr int HD = D / 2, LDW = HD + 4, NV = D / 32;     // NV: f32x4 staging loads per thread per tile
  if (guard != nullptr && *guard == 0) return;             // fallback launch of the shortlist search: nothing to redo
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* se = sm;                              // [2][32][2][LDW]
  float* sb = sm + 2 * 32 * 2 * LDW;           // [2][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + l32;

  float zr[HD];
  float zz = 0.f;
  {
    const float* zp = z + (size_t)(row < rows ? row : rows - 1) * D;
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4) {
      const f32x4 v = *(const f32x4*)(zp + c4 * 4);
      zz = vq_sq4(zz, v);
      zr[2 * c4] = h ? v[1] : v[0];
      zr[2 * c4 + 1] = h ? v[3] : v[2];
      if (c4 % 8 == 7) __builtin_amdgcn_sched_barrier(0);      // keep at most 8 row loads in flight (register pressure)
    }
  }
  const int ntiles = (K + 31) / 32;
  f32x4 pre[NV];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = i * 256 + tid, r = q / (D / 4), c4 = q % (D / 4);
      const int code = tile * 32 + r;
      pre[i] = code < K ? *(const f32x4*)(cb + (size_t)code * D + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto put = [&](int tile, int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = i * 256 + tid, r = q / (D / 4), c4 = q % (D / 4);
      float* d0 = se + ((buf * 32 + r) * 2) * LDW + 2 * c4;
      *(f32x2*)d0 = f32x2{pre[i][0], pre[i][2]};
      *(f32x2*)(d0 + LDW) = f32x2{pre[i][1], pre[i][3]};
    }
    if (tid < 32) sb[buf * 32 + tid] = tile * 32 + tid < K ? e2[tile * 32 + tid] : 0.f;
  };
  fetch(0);
  put(0, 0);
  __syncthreads();

  float best = __builtin_inff();
  int besti = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch(t + 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ea = se + ((buf * 32 + l32) * 2 + h) * LDW;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
      const f32x4 a = *(const f32x4*)(ea + 4 * s4);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], zr[4 * s4 + i], acc, 0, 0, 0);
      if (s4 % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // at most four fragment vectors hoisted ahead
    }
    if (t + 1 < ntiles) put(t + 1, buf ^ 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int code = t * 32 + cl;
      const float dist = (zz + sb[buf * 32 + cl]) - 2.f * acc[r];
      if (code < K && dist < best) { best = dist; besti = code; }
    }
    __syncthreads();
  }
  const float ob = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(besti, 32, 64);
  if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  if (h == 0 && row < rows) idx[row] = besti;
}

template <int D>
static int launch_vq_assign2(const float* z, const float* cb, float* e2, int64_t rows, int k, int64_t* idx, hipStream_t st) {
  constexpr size_t lds = (size_t)(2 * 32 * 2 * (D / 2 + 4) + 64) * sizeof(float);
  VG_HIP(vg_lds_attr((const void*)vq_assign2_kernel<D>, (int)lds));
  hipLaunchKernelGGL(vq_code_norms_kernel, dim3((unsigned)vg_ceil_div(k, 256)), dim3(256), 0, st, cb, k, D, e2);
  VG_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_assign2_kernel<D>, dim3((unsigned)vg_ceil_div(rows, 128)), dim3(256), lds, st, z, cb, (const float*)e2, rows, k, idx,
                     (const int*)nullptr);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_vq_assign(const float* z, const float* codebook, int64_t rows, int k, int d, int64_t* idx,
                            float* norms_ws, void* stream) {
  VG_REQUIRE(z && codebook && idx && rows > 0 && k > 0, "vq_assign: bad arguments");
  VG_REQUIRE(d % 4 == 0 && d >= 4 && d <= 256, "vq_assign: embedding_dim %d unsupported (multiple of 4, <= 256)", d);
  hipStream_t st = (hipStream_t)stream;
  if (norms_ws != nullptr) {
    if (d == 64) return launch_vq_assign2<64>(z, codebook, norms_ws, rows, k, idx, st);
    if (d == 128) return launch_vq_assign2<128>(z, codebook, norms_ws, rows, k, idx, st);
    if (d == 256) return launch_vq_assign2<256>(z, codebook, norms_ws, rows, k, idx, st);
  }
  const int nw = d <= 128 ? 4 : 2;
  const size_t lds = ((size_t)(nw * 32 + 64) * (d + 1) + 64) * sizeof(float);
  const unsigned blocks = (unsigned)vg_ceil_div(rows, nw * 32);
  if (nw == 4) {
    VG_HIP(vg_lds_attr((const void*)vq_assign_kernel<4>, 160 * 1024));
    hipLaunchKernelGGL(vq_assign_kernel<4>, dim3(blocks), dim3(256), lds, st, z, codebook, rows, k, d, idx);
  } else {
    VG_HIP(vg_lds_attr((const void*)vq_assign_kernel<2>, 160 * 1024));
    hipLaunchKernelGGL(vq_assign_kernel<2>, dim3(blocks), dim3(128), lds, st, z, codebook, rows, k, d, idx);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// Shortlist search for large codebooks (K = 8192, D = 256: the exact-f32 MFMA search above is 275 GFLOP on a 157 TF/s
// pipe, 2.6 ms).  The SAME indices, bit for bit, from three steps:
//  1. approximate distances on the bf16 MFMA (16x the rate): dq = |E_k|^2 - 2 zb.eb with z and E rounded to bf16.  For
//     every (row, code) the exact kernel's f32 distance d satisfies |dq + |z|^2 - d| <= B_k with
//       B_k = C1 |z| |E_k| + C2 (|z|^2 + |E_k|^2),   C1 = 1.02 * 2^-6,  C2 = 1.5 * 2^-22
//     (bf16 rounding of both operands: bf16 keeps 8 significant bits, so round-to-nearest moves an element by up to 2^-8
//     of itself, |zb.eb - z.e| <= (2^-7 + 2^-16) |z||E_k| by Cauchy-Schwarz, twice that for the factor 2: (2^-6 + 2^-15)
//     |z||E_k|.  Round 2 had 2^-7 here (a unit roundoff of 2^-9): random data never shows it, because the rounding errors
//     of 64..256 coordinates average out, but with every coordinate of z and of the winning code just beside a bf16
//     midpoint, all errors pushing the same way, the true argmin was dropped in pass 2:
//     tests/test_gpu_vq_shortlist.py::test_shortlist_worst_case_rounding.  f32 accumulation of either
//     kernel: D * 2^-24 |z||E_k| each, twice for the factor 2; the roundings of |z|^2 + |E_k|^2, of d, of dq and of the
//     bound arithmetic: (1 + 2 + 1 + 1) * 2^-24 (|z|^2 + |E_k|^2), using 2 |z||E_k| <= |z|^2 + |E_k|^2; the rest of
//     the 2 % and of the 1.5 is slack for the f32 norms and the bound's own arithmetic).  With latents of order one and a freshly initialised
//     codebook (codes ~ 1/K) the f32 distances themselves are only a few ulp(|z|^2) apart, so a row keeps 2-3
//     candidates on average and up to ~20: the list holds VQS_CAP = 32.  (The bound is taken per code, not per tile of
//     codes: once the EMA has moved some codes to the latents' scale, a tile-wide bound made every small code that
//     shares a tile with a large one a candidate.)
//     pass 1: U = min_k (dq_k + B_k) per row: an upper bound of (exact minimum - |z|^2).
//     pass 2 (the same product again: it is cheap): every code with dq_k - B_k <= U is a candidate.  The exact argmin
//     k* always is: dq_k* - B_k* <= d_k* - |z|^2 <= d_j - |z|^2 <= dq_j + B_j for every j.  So is every code that ties.
//  2. the candidates' distances on the exact-f32 MFMA, the same instruction sequence as vq_assign2 (a wave takes 32
//     rows: A row m = candidate s of row m, B column m = z row m, the diagonal of the 32x32 tile is what is wanted),
//     smallest distance, then smallest index.  Rows with one candidate skip this.
//  3. a row with no candidate or more than VQS_CAP (32) sets a flag; vq_assign2 is launched after it and returns at once
//     unless the flag is set (then it redoes the whole search: correct always, fast when the bound is not hit).
// ---------------------------------------------------------------------------------------------
constexpr int VQS_CAP = 32;
#define VQS_C1 (1.02f / 64.f)
#define VQS_C2 (1.5f / 4194304.f)

// one block per tile of 32 codes: bf16 image of the codes (zero rows beyond K) and |E_k| (rounded up)
__global__ __launch_bounds__(256) void vq_short_prep_kernel(const float* __restrict__ cb, const float* __restrict__ e2, int K, int D,
                                                            bf16_t* __restrict__ eb, float* __restrict__ en) {
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int q = tid; q < 32 * D / 4; q += 256) {
    const int r = q / (D / 4), c4 = q - r * (D / 4);
    const int code = t * 32 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (code < K) v = *(const f32x4*)(cb + (size_t)code * D + c4 * 4);
    *(bf16x4*)(eb + (size_t)code * D + c4 * 4) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  }
  if (tid < 32) {
    const int code = t * 32 + tid;
    en[code] = code < K ? sqrtf(e2[code]) * 1.000001f : 0.f;            // en has ntiles * 32 entries
  }
}

template <int D, int J, int PASS>
__global__ __launch_bounds__(256, 2) void vq_short_kernel(const float* __restrict__ z, const bf16_t* __restrict__ eb,
                                                          const float* __restrict__ e2, const float* __restrict__ en,
                                                          int64_t rows, int K, float* __restrict__ U, int* __restrict__ cnt,
                                                          int* __restrict__ cand) {
  constexpr int NS = D / 16;             // MFMA k steps
  constexpr int STR = D * 2 + 16;        // LDS row stride (bytes): 16-byte fragment reads of 32 rows spread over the banks
  constexpr int NV = D / 64;             // 16-byte staging loads per thread per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smq[];
  unsigned char* st = smq;                         // [2][32][STR]
  float* sb = (float*)(smq + 2 * 32 * STR);        // [2][2][32] |E_k|^2 (1 +- C2) (+inf beyond K), |E_k|
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * J) + l32;

  bf16x8 zf[J][NS];
  float cz[J], c2zz[J], ub[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t row = row0 + 32 * j;
    const float* zp = z + (size_t)(row < rows ? row : rows - 1) * D + 8 * h;
    float n2 = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *(const f32x4*)(zp + 16 * s), b = *(const f32x4*)(zp + 16 * s + 4);
      zf[j][s] = bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
      n2 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // eight row loads in flight, not all of them (registers)
    }
    n2 += __shfl_xor(n2, 32, 64);
    cz[j] = VQS_C1 * sqrtf(n2);
    c2zz[j] = VQS_C2 * n2;
    ub[j] = PASS == 1 ? __builtin_inff() : U[row < rows ? row : rows - 1] + c2zz[j];      // U holds min + C2 |z|^2
  }
  const int ntiles = (K + 31) / 32;
  u32x4 pre[NV];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = i * 256 + tid, r = q / (D / 8), pc = q % (D / 8);
      pre[i] = *(const u32x4*)(eb + ((size_t)tile * 32 + r) * D + pc * 8);       // eb has ntiles * 32 rows
    }
  };
  auto put = [&](int tile, int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = i * 256 + tid, r = q / (D / 8), pc = q % (D / 8);
      *(u32x4*)(st + (buf * 32 + r) * STR + pc * 16) = pre[i];
    }
    if (tid < 32) sb[buf * 64 + tid] = tile * 32 + tid < K ? e2[tile * 32 + tid] * (PASS == 1 ? 1.f + VQS_C2 : 1.f - VQS_C2) : __builtin_inff();
    else if (tid < 64) sb[buf * 64 + tid] = en[tile * 32 + tid - 32];
  };
  fetch(0);
  put(0, 0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch(t + 1);
    f32x16 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned char* base = st + (buf * 32 + l32) * STR + 16 * h;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bf16x8 a = *(const bf16x8*)(base + 32 * s);
#pragma unroll
      for (int j = 0; j < J; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zf[j][s], acc[j], 0, 0, 0);
      if (J * D >= 512 && (s & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // at most four fragment reads hoisted ahead
    }
    if (t + 1 < ntiles) put(t + 1, buf ^ 1);
    // dq_k +- B_k = (|E_k|^2 (1 +- C2) +- C1 |z| |E_k|) - 2 zb.eb +- C2 |z|^2: two FMAs per element, the row term apart
    float e2r[16], enr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
      e2r[r] = sb[buf * 64 + cl];
      enr[r] = sb[buf * 64 + 32 + cl];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (PASS == 1) {
        float m = ub[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fminf(m, __builtin_fmaf(-2.f, acc[j][r], __builtin_fmaf(cz[j], enr[r], e2r[r])));
        ub[j] = m;
      } else {
        unsigned hit = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          hit |= (__builtin_fmaf(-2.f, acc[j][r], __builtin_fmaf(-cz[j], enr[r], e2r[r])) <= ub[j]) ? 1u << r : 0u;
        const int64_t row = row0 + 32 * j;
        if (hit != 0 && row < rows) {
          while (hit) {
            const int r = __builtin_ctz(hit);
            hit &= hit - 1;
            const int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int pos = atomicAdd(&cnt[row], 1);
            if (pos < VQS_CAP) cand[(size_t)row * VQS_CAP + pos] = code;
          }
        }
      }
    }
    __syncthreads();
  }
  if (PASS == 1) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float u = fminf(ub[j], __shfl_xor(ub[j], 32, 64)) + c2zz[j];
      const int64_t row = row0 + 32 * j;
      if (h == 0 && row < rows) U[row] = u;
    }
  }
}

// exact distances of the candidates: the arithmetic of vq_assign2 (same |z|^2 loop, same MFMA chain in k order, same
// distance expression), so the same bits.  A wave takes 32 rows and walks their (row, candidate) PAIRS 32 at a time
// (rows keep 2-3 candidates on average but up to 32: a slot-per-round walk ran as many rounds as the longest list):
// lane m of a round holds pair p = 32 * round + m, A row m = its code, B column m = its z row, and the diagonal of the
// 32x32 tile is the pair's product.  The distances go to LDS; each row then picks among its own.
template <int D>
__global__ __launch_bounds__(256) void vq_short_exact_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                             const float* __restrict__ e2, int64_t rows, int K,
                                                             const int* __restrict__ cnt, const int* __restrict__ cand,
                                                             int64_t* __restrict__ idx, int* __restrict__ flag) {
  __shared__ int spre[4][32];
  __shared__ float sdist[4][32 * VQS_CAP];
  __shared__ int scode[4][32 * VQS_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int64_t row = base + l32;
  const int64_t rc = row < rows ? row : rows - 1;
  int n = cnt[rc];
  if ((n < 1 || n > VQS_CAP) && row < rows && h == 0) atomicOr(flag, 1);
  n = n < 1 ? 1 : (n > VQS_CAP ? VQS_CAP : n);
  int incl = n;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int t = __shfl_up(incl, o, 32);
    if (l32 >= o) incl += t;
  }
  const int excl = incl - n;
  const int P = __shfl(incl, 31, 32);                          // pairs of the wave's 32 rows (the same in both halves)
  if (P == 32) {                                               // wave-uniform: one candidate per row, nothing to compare
    if (h == 0 && row < rows) idx[row] = min(max(cand[(size_t)rc * VQS_CAP], 0), K - 1);
    return;
  }
  if (h == 0) spre[wave][l32] = excl;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int hsel = (l32 >> 2) & 1, rsel = (l32 & 3) + 4 * (l32 >> 3);      // where the diagonal element of column l32 sits
  for (int p0 = 0; p0 < P; p0 += 32) {
    const int p = p0 + l32 < P ? p0 + l32 : P - 1;
    int r = 0;
#pragma unroll
    for (int st = 16; st > 0; st >>= 1)
      if (spre[wave][r + st] <= p) r += st;                    // r + st <= 31
    const int64_t prow = base + r < rows ? base + r : rows - 1;
    int code = cand[(size_t)prow * VQS_CAP + (p - spre[wave][r])];
    code = min(max(code, 0), K - 1);
    const float* zp = z + (size_t)prow * D;
    const float* ep = cb + (size_t)code * D;
    float zz = 0.f;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4) {
      const f32x4 v = *(const f32x4*)(zp + c4 * 4);
      const f32x4 e = *(const f32x4*)(ep + c4 * 4);
      zz = vq_sq4(zz, v);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? e[1] : e[0], h ? v[1] : v[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? e[3] : e[2], h ? v[3] : v[2], acc, 0, 0, 0);
      if (c4 % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    }
    float dv = acc[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) dv = rsel == q ? acc[q] : dv;
    const float ov = __shfl_xor(dv, 32, 64);
    const float sd = h == hsel ? dv : ov;
    const float dist = (zz + e2[code]) - 2.f * sd;
    if (h == 0 && p0 + l32 < P) { sdist[wave][p] = dist; scode[wave][p] = code; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (h == 0 && row < rows) {
    float best = __builtin_inff();
    int besti = 0x7fffffff;
    for (int q = excl; q < excl + n; ++q) {
      const float dist = sdist[wave][q];
      const int code = scode[wave][q];
      if (dist < best || (dist == best && code < besti)) { best = dist; besti = code; }
    }
    idx[row] = besti == 0x7fffffff ? 0 : besti;
  }
}

struct VqShortWs {
  float *e2, *en, *U;
  int *cnt, *flag, *cand;
  bf16_t* eb;
  size_t bytes;
};
static VqShortWs vq_short_layout(void* ws, int64_t rows, int k, int d) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t ntiles = (size_t)(k + 31) / 32;
  unsigned char* b = (unsigned char*)ws;
  VqShortWs w;
  size_t o = 0;
  w.e2 = (float*)(b + o); o += up((size_t)k * 4);
  w.en = (float*)(b + o); o += up(ntiles * 32 * 4);
  w.U = (float*)(b + o); o += up((size_t)rows * 4);
  w.cnt = (int*)(b + o); o += up((size_t)rows * 4);          // cnt and flag are cleared by one memset
  w.flag = (int*)(b + o); o += 256;
  w.cand = (int*)(b + o); o += up((size_t)rows * VQS_CAP * 4);
  w.eb = (bf16_t*)(b + o); o += up(ntiles * 32 * (size_t)d * 2);
  w.bytes = o;
  return w;
}

extern "C" size_t vg_vq_shortlist_ws_bytes(int64_t rows, int k, int d) {
  if (rows <= 0 || k <= 0 || (d != 64 && d != 128 && d != 256)) return 0;
  return vq_short_layout(nullptr, rows, k, d).bytes;
}

template <int D>
static int launch_vq_short(const float* z, const float* cb, int64_t rows, int k, int64_t* idx, const VqShortWs& w, hipStream_t st) {
  const int ntiles = (k + 31) / 32;
  constexpr size_t lds = (size_t)2 * 32 * (D * 2 + 16) + 512;
  hipLaunchKernelGGL(vq_code_norms_kernel, dim3((unsigned)vg_ceil_div(k, 256)), dim3(256), 0, st, cb, k, D, w.e2);
  hipLaunchKernelGGL(vq_short_prep_kernel, dim3(ntiles), dim3(256), 0, st, cb, (const float*)w.e2, k, D, w.eb, w.en);
  VG_HIP(hipMemsetAsync(w.cnt, 0, (size_t)((unsigned char*)w.flag - (unsigned char*)w.cnt) + 256, st));
  // two row sets per wave (each fragment read feeds two MFMAs) once that still gives every CU two blocks; at D = 256
  // the second set of z fragments does not fit the register file
  bool two = false;
  if constexpr (D <= 128) two = rows >= 2 * 256 * 256;
  if constexpr (D <= 128) if (two) {
    const unsigned blocks = (unsigned)vg_ceil_div(rows, 256);
    hipLaunchKernelGGL((vq_short_kernel<D, 2, 1>), dim3(blocks), dim3(256), lds, st, z, (const bf16_t*)w.eb, (const float*)w.e2,
                       (const float*)w.en, rows, k, w.U, w.cnt, w.cand);
    hipLaunchKernelGGL((vq_short_kernel<D, 2, 2>), dim3(blocks), dim3(256), lds, st, z, (const bf16_t*)w.eb, (const float*)w.e2,
                       (const float*)w.en, rows, k, w.U, w.cnt, w.cand);
  }
  if (!two) {
    const unsigned blocks = (unsigned)vg_ceil_div(rows, 128);
    hipLaunchKernelGGL((vq_short_kernel<D, 1, 1>), dim3(blocks), dim3(256), lds, st, z, (const bf16_t*)w.eb, (const float*)w.e2,
                       (const float*)w.en, rows, k, w.U, w.cnt, w.cand);
    hipLaunchKernelGGL((vq_short_kernel<D, 1, 2>), dim3(blocks), dim3(256), lds, st, z, (const bf16_t*)w.eb, (const float*)w.e2,
                       (const float*)w.en, rows, k, w.U, w.cnt, w.cand);
  }
  VG_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_short_exact_kernel<D>, dim3((unsigned)vg_ceil_div(rows, 128)), dim3(256), 0, st, z, cb, (const float*)w.e2, rows, k,
                     (const int*)w.cnt, (const int*)w.cand, idx, w.flag);
  VG_LAUNCH_CHECK();
  constexpr size_t lds2 = (size_t)(2 * 32 * 2 * (D / 2 + 4) + 64) * sizeof(float);
  VG_HIP(vg_lds_attr((const void*)vq_assign2_kernel<D>, (int)lds2));
  hipLaunchKernelGGL(vq_assign2_kernel<D>, dim3((unsigned)vg_ceil_div(rows, 128)), dim3(256), lds2, st, z, cb, (const float*)w.e2, rows, k, idx,
                     (const int*)w.flag);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_vq_assign_shortlist(const float* z, const float* codebook, int64_t rows, int k, int d, int64_t* idx, void* ws,
                                      int* fallback_flag_out, void* stream) {
  VG_REQUIRE(z && codebook && idx && ws && rows > 0 && k > 0, "vq_assign_shortlist: bad arguments");
  VG_REQUIRE(d == 64 || d == 128 || d == 256, "vq_assign_shortlist: embedding_dim %d unsupported (64, 128 or 256)", d);
  VG_REQUIRE(((uintptr_t)ws & 255) == 0, "vq_assign_shortlist: the workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const VqShortWs w = vq_short_layout(ws, rows, k, d);
  int rc;
  if (d == 64) rc = launch_vq_short<64>(z, codebook, rows, k, idx, w, st);
  else if (d == 128) rc = launch_vq_short<128>(z, codebook, rows, k, idx, w, st);
  else rc = launch_vq_short<256>(z, codebook, rows, k, idx, w, st);
  if (rc) return rc;
  if (fallback_flag_out != nullptr) VG_HIP(hipMemcpyAsync(fallback_flag_out, w.flag, sizeof(int), hipMemcpyDeviceToDevice, st));
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// gather + straight-through value + squared error + EMA statistics
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_gather_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                        const int64_t* __restrict__ idx, int64_t rows, int K, int D,
                                                        float* __restrict__ zq, double* __restrict__ sse,
                                                        float* __restrict__ counts, float* __restrict__ esum, int det) {
  const int tpr = D / 4;                       // threads per row
  const int rpb = 256 / tpr;
  const int c4 = threadIdx.x % tpr, rr = threadIdx.x / tpr;
  float local = 0.f;
  if (rr < rpb) {
    for (int64_t r = (int64_t)blockIdx.x * rpb + rr; r < rows; r += (int64_t)gridDim.x * rpb) {
      int code = (int)idx[r];
      if (code < 0) code = 0;
      if (code >= K) code = K - 1;
      const f32x4 zv = *(const f32x4*)(z + (size_t)r * D + c4 * 4);
      const f32x4 qv = *(const f32x4*)(cb + (size_t)code * D + c4 * 4);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float diff = qv[e] - zv[e];
        o[e] = zv[e] + diff;
        local += diff * diff;
      }
      *(f32x4*)(zq + (size_t)r * D + c4 * 4) = o;
      if (esum != nullptr) {
        if (!det) {        // det: vq_esum_ordered_kernel
#pragma unroll
          for (int e = 0; e < 4; ++e) unsafeAtomicAdd(&esum[(size_t)code * D + c4 * 4 + e], zv[e]);
        }
        if (c4 == 0) unsafeAtomicAdd(&counts[code], 1.0f);     // whole numbers below 2^24: exact in any order
      }
    }
  }
  __shared__ double red[4];
  double w = wave_sum_d((double)local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0 && sse != nullptr) unsafeAtomicAdd(sse, red[0] + red[1] + red[2] + red[3]);
}

// The same with the EMA statistics privatised in LDS (k*d + k floats <= 144 KB): the rows of a batch pile onto few
// codes (a fresh codebook: nearly all onto a handful), so global f32 atomics on esum[code][:] serialise (230 us for
// 65,536 x 64 at K = 512).  A block of 1024 threads accumulates its rows with LDS atomics and flushes the non-zero
// entries once (round 3: a wave per row, 256 blocks: 100 -> 47 us).
__global__ __launch_bounds__(1024) void vq_gather_lds_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                             const int64_t* __restrict__ idx, int64_t rows, int K, int D,
                                                             float* __restrict__ zq, double* __restrict__ sse,
                                                             float* __restrict__ counts, float* __restrict__ esum) {
  extern __shared__ float s_acc[];             // [K*D] sums, [K] counts
  float* s_cnt = s_acc + (size_t)K * D;
  const int total = K * D + K;
  for (int i = threadIdx.x; i < total; i += 1024) s_acc[i] = 0.f;
  __syncthreads();
  // A WAVE per row, a lane per channel (c = lane, lane + 64, ...): the 64 lanes of an LDS atomic then hit 64 different
  // banks.  (A row per 16 lanes with four channels each put the four rows of a wave on the same 16 banks, every code row
  // starting at bank 0: f32 LDS atomics four lanes deep per bank were what the kernel spent its time on.)  U rows in
  // flight per wave: a row is two dependent memory latencies (index, then code vector).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int U = 4;
  float local = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t r0 = (int64_t)blockIdx.x * 16 + wave; r0 < rows; r0 += U * stride) {
    int code[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      int c = r < rows ? (int)idx[r] : 0;
      if (c < 0) c = 0;
      if (c >= K) c = K - 1;
      code[u] = c;
    }
    for (int c0 = 0; c0 < D; c0 += 64) {
      const int c = c0 + lane;
      float zv[U], qv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        const bool ok = r < rows && c < D;
        zv[u] = ok ? z[(size_t)r * D + c] : 0.f;
        qv[u] = ok ? cb[(size_t)code[u] * D + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        if (r < rows && c < D) {
          const float diff = qv[u] - zv[u];
          local += diff * diff;
          zq[(size_t)r * D + c] = zv[u] + diff;
          atomicAdd(&s_acc[(size_t)code[u] * D + c], zv[u]);
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r0 + u * stride < rows) atomicAdd(&s_cnt[code[u]], 1.0f);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * D; i += 1024)
    if (s_acc[i] != 0.f) unsafeAtomicAdd(&esum[i], s_acc[i]);
  for (int i = threadIdx.x; i < K; i += 1024)
    if (s_cnt[i] != 0.f) unsafeAtomicAdd(&counts[i], s_cnt[i]);
  __shared__ double red[16];
  double w = wave_sum_d((double)local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0 && sse != nullptr) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    unsafeAtomicAdd(sse, t);
  }
}

// Reproducible EMA sums (vg_set_deterministic): one block per code walks the assignment in row order; a wave owns a
// quarter of the rows, finds its code's rows 64 at a time (ballot) and adds them in ascending order, one lane per
// channel; the four quarter sums are combined in a fixed order.  K blocks re-read idx (rows * 8 bytes each, from L2).
__global__ __launch_bounds__(256) void vq_esum_ordered_kernel(const float* __restrict__ z, const int64_t* __restrict__ idx,
                                                              int64_t rows, int K, int D, float* __restrict__ esum) {
  constexpr int MAXJ = 16;                     // D <= 1024
  __shared__ float part[4][1024];
  const int code = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) acc[j] = 0.f;
  const int64_t q = (rows + 3) / 4;
  const int64_t r1 = min(rows, (int64_t)(wave + 1) * q);
  for (int64_t r0 = (int64_t)wave * q; r0 < r1; r0 += 64) {
    const int64_t r = r0 + lane;
    int c = r < r1 ? (int)idx[r] : -1;
    if (r < r1) { if (c < 0) c = 0; if (c >= K) c = K - 1; }
    unsigned long long m = __ballot(c == code);
    while (m) {
      const int bit = __ffsll((long long)m) - 1;
      m &= m - 1;
      const float* zr = z + (size_t)(r0 + bit) * D;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
        if (lane + 64 * j < D) acc[j] += zr[lane + 64 * j];
    }
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
    if (lane + 64 * j < D) part[wave][lane + 64 * j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) esum[(size_t)code * D + c] += (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

extern "C" int vg_vq_gather(const float* z, const float* codebook, const int64_t* idx, int64_t rows, int k, int d,
                            float* zq, double* sse, float* counts, float* esum, void* stream) {
  VG_REQUIRE(z && codebook && idx && zq && rows > 0 && k > 0, "vq_gather: bad arguments");
  VG_REQUIRE(d % 4 == 0 && d / 4 <= 256, "vq_gather: embedding_dim %d unsupported", d);
  VG_REQUIRE((counts == nullptr) == (esum == nullptr), "vq_gather: counts and esum must be given together");
  const size_t lds = sizeof(float) * ((size_t)k * d + k);
  const int det = vg_det_mode();
  if (esum != nullptr && lds <= 144 * 1024 && rows >= 8192 && !det) {
    VG_HIP(vg_lds_attr((const void*)vq_gather_lds_kernel, 144 * 1024));
    // one block per CU (its LDS holds nothing else), >= 16 rows per wave before a block pays for zeroing and flushing its
    // 131 KB: 65,536 x 64 at K = 512: 64 us with 128 blocks, 47 with 256, 53 with 512
    int64_t nb = vg_ceil_div(rows, (int64_t)16 * 16);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(vq_gather_lds_kernel, dim3((unsigned)nb), dim3(1024), lds, (hipStream_t)stream, z, codebook, idx, rows,
                       k, d, zq, sse, counts, esum);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  const int rpb = 256 / (d / 4);
  int64_t blocks = vg_ceil_div(rows, (int64_t)rpb * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vq_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, codebook, idx, rows, k,
                     d, zq, sse, counts, esum, det);
  VG_LAUNCH_CHECK();
  if (det && esum != nullptr) {
    hipLaunchKernelGGL(vq_esum_ordered_kernel, dim3((unsigned)k), dim3(256), 0, (hipStream_t)stream, z, idx, rows, k, d, esum);
    VG_LAUNCH_CHECK();
  }
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// EMA update: cluster sizes + their sum (one block), then the codebook rows (grid)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vq_ema_cluster_kernel(float* __restrict__ cluster, const float* __restrict__ counts,
                                                              int K, float decay, float* __restrict__ nsum) {
  __shared__ float red[16];
  float local = 0.f;
  for (int i = threadIdx.x; i < K; i += 1024) {
    float v = cluster[i] * decay + (1.f - decay) * counts[i];
    cluster[i] = v;
    local += v;
  }
  float w = wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    *nsum = s;
  }
}

__global__ __launch_bounds__(256) void vq_ema_weight_kernel(const float* __restrict__ cluster, float* __restrict__ ema_w,
                                                            float* __restrict__ cb, const float* __restrict__ esum, int K,
                                                            int D, float decay, const float* __restrict__ nsum) {
  const float n = *nsum;
  const int64_t total = (int64_t)K * D;
  for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int code = (int)(t / D);
    const float w = ema_w[t] * decay + (1.f - decay) * esum[t];
    ema_w[t] = w;
    const float cs = (cluster[code] + 1e-5f) / (n + (float)K * 1e-5f) * n;
    cb[t] = w / cs;
  }
}

extern "C" int vg_vq_ema_update(float* ema_cluster, float* ema_w, float* codebook, const float* counts, const float* esum,
                                int k, int d, float decay, float* nsum, void* stream) {
  VG_REQUIRE(ema_cluster && ema_w && codebook && counts && esum && nsum && k > 0 && d > 0, "vq_ema_update: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vq_ema_cluster_kernel, dim3(1), dim3(1024), 0, st, ema_cluster, counts, k, decay, nsum);
  VG_LAUNCH_CHECK();
  int64_t total = (int64_t)k * d;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vq_ema_weight_kernel, dim3(blocks), dim3(256), 0, st, ema_cluster, ema_w, codebook, esum, k, d, decay, nsum);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// dz = g_zq + (*g_loss)*coef*(z - zq)     (commitment term; straight-through passes g_zq unchanged)
__global__ __launch_bounds__(256) void vq_bwd_kernel(const float* __restrict__ g_zq, const float* __restrict__ g_loss,
                                                     const float* __restrict__ z, const float* __restrict__ zq, int64_t n4,
                                                     float coef, float* __restrict__ dz) {
  const float gl = g_loss ? *g_loss * coef : 0.f;
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const f32x4 a = *(const f32x4*)(z + q * 4), b = *(const f32x4*)(zq + q * 4);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (g_zq) g = *(const f32x4*)(g_zq + q * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] + gl * (a[e] - b[e]);
    *(f32x4*)(dz + q * 4) = o;
  }
}

extern "C" int vg_vq_bwd(const float* g_zq, const float* g_loss, const float* z, const float* zq, int64_t n, float coef,
                         float* dz, void* stream) {
  VG_REQUIRE(z && zq && dz && n > 0 && n % 4 == 0, "vq_bwd: bad arguments");
  int64_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vq_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_zq, g_loss, z, zq, n4, coef, dz);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
