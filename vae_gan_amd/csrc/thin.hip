// Thin layers: convolutions with a 3-channel side (the image-side tensors: encoder / discriminator inputs, decoder
// output and their gradients), bf16.
//
// As ordinary gather-GEMMs these layers pad 3 channels to one 64-byte K chunk (32 channels): ~0.1 % of the step's
// FLOPs took 2.8 ms of a 44 ms step, and the image tensors were 10x their payload in HBM.  Here the 3-channel tensor
// is NHWC with FOUR channels (8 bytes per pixel, channel 3 = 0) and the taps go into the GEMM dimension that would
// otherwise be padding:
//   * thin -> wide  (conv 3->64: forward; input gradient of 64->3): K = (r, s, c) = 4x4x4 = 64, one im2col row per
//     pixel read straight out of the 8-byte-pixel halo in LDS;
//   * wide -> thin  (conv 64->3: forward; input gradient of 3->64, stride 1 and the stride-2 transposed gather):
//     N = (r, s, c): P[q][(r,s,c)] = sum_n in[q][n] W[..] is a plain 1x1 GEMM over the input tile, then a col2im
//     gather-sum over the taps in LDS;
//   * weight gradient: dW[n][(r,s,c)] = sum over pixels, K = pixels, both operands transposed into LDS.
// All three read the master OIHW f32 weights themselves (a few thousand values): no packed weight images.
// Every kernel is bound by the one pass over the 64-channel tensor it touches (HBM).
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

struct ThinParams {
  const bf16_t* thin;   // [N][TH_][TW_][4]
  const bf16_t* wide;   // [N][WH][WW][Ws]
  bf16_t* out;          // the tensor being written (thin or wide)
  const float* w;       // master weight [O][I][R][R]
  const float* bias;
  double* stats;        // thin->wide forward: [VG_STATS_REPLICAS][2][cpad]
  int det;              // statistics through the fixed-point accumulators (common.h: vg_stats_add)
  float* part;          // weight gradient: [blocks][64][64] partial images
  const bf16_t* wpk;    // packed [64][72] weight image (thin_pack_kernel)
  int N, TH_, TW_, WH, WW, Ws;
  int O, I, R, stride, pad;   // pad: of the thin-side gather (see each kernel)
  int wmode, nw, nt, act, cpad;
  int tilesX, tilesY, ntiles;
};

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The [64][72] bf16 weight image a block keeps in LDS (rows of 144 bytes: conflict-free 16-byte fragment reads),
// made once per launch from the master OIHW f32 weights by one small block; the compute blocks copy it with 16-byte
// loads.  (Filled by every block straight from the f32 master, the 16 dependent scattered loads per thread cost more
// than the block's whole MFMA work.)
//   wide_out = 1 (thin_in_kernel): row n, column k = (r*4+s)*4+c:  wmode 0: W[n][c][r][s]; 1: W[c][n][R-1-r][R-1-s]
//   wide_out = 0 (thin_out_kernel): row (r*R+s)*4+c, column n:     wmode 0: W[c][n][r][s]; 1: W[n][c][R-1-r][R-1-s]; 2: W[n][c][r][s]
__global__ __launch_bounds__(256) void thin_pack_kernel(const ThinParams p, bf16_t* __restrict__ wpk, int wide_out) {
  const int R = p.R;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 64 * 72; idx += gridDim.x * 256) {      // one element per thread
    const int row = idx / 72, col = idx - row * 72;
    float v = 0.f;
    if (col < 64) {
      if (wide_out) {
        const int n = row, r = col >> 4, s = (col >> 2) & 3, c = col & 3;
        if (r < R && s < R && c < p.nt && n < p.nw)
          v = p.wmode == 0 ? p.w[(((size_t)n * p.I + c) * R + r) * R + s]
                           : p.w[(((size_t)c * p.I + n) * R + (R - 1 - r)) * R + (R - 1 - s)];
      } else {
        const int n = col, tap = row >> 2, c = row & 3, r = tap / R, s = tap - r * R;
        if (row < R * R * 4 && c < p.nt && n < p.nw) {
          if (p.wmode == 0) v = p.w[(((size_t)c * p.I + n) * R + r) * R + s];
          else if (p.wmode == 1) v = p.w[(((size_t)n * p.I + c) * R + (R - 1 - r)) * R + (R - 1 - s)];
          else v = p.w[(((size_t)n * p.I + c) * R + r) * R + s];
        }
      }
    }
    wpk[idx] = (bf16_t)v;
  }
}
__device__ __forceinline__ void thin_load_weights(bf16_t* wl, const bf16_t* __restrict__ wpk, int tid) {
  for (int q = tid; q < 64 * 72 / 8; q += 256) *(u32x4*)(wl + q * 8) = *(const u32x4*)(wpk + q * 8);
}

// ---------------------------------------------------------------------------------------------
// thin -> wide.  out[pix][n] = sum_{r,s,c} thin[pix*ST - pad + (r,s)][c] * Wk[n][r][s][c],  k = (r*4 + s)*4 + c.
// wmode 0: Wk[n][r][s][c] = W[n][c][r][s] (forward of a thin-input conv); 1: Wk[n][r][s][c] = W[c][n][R-1-r][R-1-s]
// (input gradient of a thin-output conv).  Block = 4 waves over an 8 x 32 output tile; a wave owns two tile rows.
// ---------------------------------------------------------------------------------------------
template <int R, int ST>
__global__ __launch_bounds__(256) void thin_in_kernel(const ThinParams p) {
  constexpr int TH = 8, TW = 32, HH = (TH - 1) * ST + R, HW = (TW - 1) * ST + R, HWP = HW + (HW & 1) + 2;
  __shared__ u32x2 halo[HH * HWP];
  __shared__ __attribute__((aligned(16))) bf16_t wl[64 * 72];
  __shared__ __attribute__((aligned(16))) bf16_t stg[4][32 * 64];
  __shared__ float red[4][64][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int tpi = p.tilesX * p.tilesY;
  const int img = blockIdx.x / tpi, trem = blockIdx.x - img * tpi;
  const int ty0 = (trem / p.tilesX) * TH, tx0 = (trem % p.tilesX) * TW;

  const u32x2* __restrict__ tg = (const u32x2*)p.thin;
  for (int idx = tid; idx < HH * HWP; idx += 256) {
    const int hy = idx / HWP, hx = idx - hy * HWP;
    const int iy = ty0 * ST - p.pad + hy, ix = tx0 * ST - p.pad + hx;
    u32x2 v = {0u, 0u};
    if ((unsigned)iy < (unsigned)p.TH_ && (unsigned)ix < (unsigned)p.TW_) v = tg[((size_t)img * p.TH_ + iy) * p.TW_ + ix];
    halo[idx] = v;
  }
  thin_load_weights(wl, p.wpk, tid);
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int j = 0; j < R; ++j) {                       // k step j = kernel row j: k = 16 j + 8 h + (0..7) = (r = j, s = 2h, 2h+1)
    u32x4 fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int base = ((2 * wave + i) * ST + j) * HWP + l32 * ST + 2 * h;
      const u32x2 lo = halo[base], hi = halo[base + 1];
      fa[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) fb[jj] = *(const u32x4*)(wl + (jj * 32 + l32) * 72 + 16 * j + 8 * h);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[i][jj] = mfma16(fa[i], fb[jj], acc[i][jj]);
  }

  // ---- epilogue: bias, BatchNorm statistics of the bf16-rounded pre-activation, activation; one 32-pixel tile row at a
  // time is transposed through a 4 KB per-wave LDS tile and leaves as 16-byte stores
  bf16_t* wt = stg[wave];          // [32 px][64 ch]
  float bv[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) bv[jj] = (p.bias != nullptr && jj * 32 + l32 < p.nw) ? p.bias[jj * 32 + l32] : 0.f;
  // whole tile, 64 real channels, no sigmoid (every launch of the step): packed f32 arithmetic, two pixels at a time,
  // no per-element bounds checks; activation and statistics are compile-time there
  const bool fast = ty0 + TH <= p.WH && tx0 + TW <= p.WW && p.nw == 64 && p.act != VG_ACT_SIGMOID;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = ty0 + 2 * wave + i;
    const bool rowok = oy < p.WH;
    if (fast) {
      bf16_t* wlp = wt + 4 * h * 64 + l32;
      auto lean = [&](auto LR, auto ST_) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          f32x2 a1 = f32x2{0.f, 0.f}, a2 = f32x2{0.f, 0.f};
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) {
            const int r = 2 * r2, row = (r & 3) + 8 * (r >> 2);
            f32x2 v = f32x2{acc[i][jj][r], acc[i][jj][r + 1]} + bv[jj];
            bf16x2_t q = __builtin_convertvector(v, bf16x2_t);
            if (decltype(ST_)::value || decltype(LR)::value) v = __builtin_convertvector(q, f32x2);
            if (decltype(ST_)::value) { a1 += v; a2 += v * v; }
            if (decltype(LR)::value) {
              const f32x2 w = v * VG_LRELU_SLOPE;
              f32x2 m;
              asm("v_max_f32 %0, %1, %2" : "=v"(m[0]) : "v"(v[0]), "v"(w[0]));
              asm("v_max_f32 %0, %1, %2" : "=v"(m[1]) : "v"(v[1]), "v"(w[1]));
              q = __builtin_convertvector(m, bf16x2_t);
            }
            wlp[row * 64 + jj * 32] = q[0];
            wlp[(row + 1) * 64 + jj * 32] = q[1];
          }
          if (decltype(ST_)::value) { s1[jj] += a1[0] + a1[1]; s2[jj] += a2[0] + a2[1]; }
        }
      };
      typedef std::integral_constant<bool, true> Yes;
      typedef std::integral_constant<bool, false> No;
      if (p.act == VG_ACT_LRELU) { if (p.stats != nullptr) lean(Yes{}, Yes{}); else lean(Yes{}, No{}); }
      else { if (p.stats != nullptr) lean(No{}, Yes{}); else lean(No{}, No{}); }
    } else {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int n = jj * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][jj][r] + bv[jj];
        v = (float)(bf16_t)v;
        if (rowok && tx0 + col < p.WW) { s1[jj] += v; s2[jj] += v * v; }
        if (p.act == VG_ACT_LRELU) v = vg_lrelu(v);
        else if (p.act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
        if (n >= p.nw) v = 0.f;
        wt[col * 64 + n] = (bf16_t)v;
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (rowok) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = lane + 64 * u;
        const int px = q >> 3, pc = q & 7;
        if (tx0 + px < p.WW)
          *(u32x4*)(p.out + (((size_t)img * p.WH + oy) * p.WW + tx0 + px) * p.Ws + pc * 8) = *(const u32x4*)(wt + px * 64 + pc * 8);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (p.stats != nullptr) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const float t1 = s1[jj] + __shfl_xor(s1[jj], 32, 64), t2 = s2[jj] + __shfl_xor(s2[jj], 32, 64);
      if (h == 0) { red[wave][jj * 32 + l32][0] = t1; red[wave][jj * 32 + l32][1] = t2; }
    }
    __syncthreads();
    if (tid < 64 && tid < p.nw) {
      const float t1 = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
      const float t2 = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
      const int rep = blockIdx.x % VG_STATS_REPLICAS;
      vg_stats_add(p.stats, p.cpad, 0, tid, rep, (double)t1, p.det);
      vg_stats_add(p.stats, p.cpad, 1, tid, rep, (double)t2, p.det);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wide -> thin.  P[q][(r,s,c)] = sum_n wide[q][n] * Wl[(r,s,c)][n] over the input tile (6 x 34 pixels), then
//   ST = 1: out[pix][c] = sum_{r,s} P[pix - pad + (r,s)][(r,s,c)]          (R x R gather, output tile 4 x 32)
//   ST = 2: out[Y][X][c] = sum_{r,s: Y+1-r, X+1-s even} P[((Y+1-r)/2, (X+1-s)/2)][(r,s,c)]   (transposed 4x4 stride 2
//           pad 1, output tile 8 x 64 over the 4 x 32 input tile + 1 halo)
// Wl by wmode: 0: W[c][n][r][s] (forward of a thin-output conv); 1: W[n][c][R-1-r][R-1-s] (input gradient of a
// thin-input stride-1 conv, pad = R-1-conv pad); 2: W[n][c][r][s] (ST = 2).
// ---------------------------------------------------------------------------------------------
template <int R, int ST>
__global__ __launch_bounds__(256) void thin_out_kernel(const ThinParams p) {
  constexpr int IH = 6, IW = 34, M = IH * IW, MB = (M + 31) / 32, NC = R * R * 4, PW = NC + 4, NLD = MB * 32 * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  bf16_t* at = (bf16_t*)tsm;                     // [MB*32][72]
  bf16_t* wl = at + MB * 32 * 72;                // [64][72]
  float* P = (float*)(wl + 64 * 72);             // [M][PW], rows 16-byte aligned
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int tpi = p.tilesX * p.tilesY;
  thin_load_weights(wl, p.wpk, tid);
  float bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bv[c] = (p.bias != nullptr && c < p.nt) ? p.bias[c] : 0.f;
  u32x2* __restrict__ og = (u32x2*)p.out;

  // a block walks tiles; the wide input tile of the NEXT tile is requested into registers before the current one is
  // computed, so its HBM latency runs under the MFMA / col2im work (one tile per block was latency-bound: 2 blocks per CU)
  u32x4 nxt[NLD];
  auto request = [&](int tile) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int iy0 = (trem / p.tilesX) * 4 - (ST == 1 ? p.pad : 1), ix0 = (trem % p.tilesX) * 32 - (ST == 1 ? p.pad : 1);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 256 * u;
      const int q = idx >> 3, pc = idx & 7;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < M) {
        const int hy = q / IW, hx = q - hy * IW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        if ((unsigned)iy < (unsigned)p.WH && (unsigned)ix < (unsigned)p.WW && pc * 8 < p.Ws)
          v = *(const u32x4*)(p.wide + (((size_t)img * p.WH + iy) * p.WW + ix) * p.Ws + pc * 8);
      }
      nxt[u] = v;
    }
  };
  int tile = blockIdx.x;
  if (tile < p.ntiles) request(tile);
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int ty0 = (trem / p.tilesX) * 4, tx0 = (trem % p.tilesX) * 32;      // core tile origin on the WIDE grid
    const int iy0 = ty0 - (ST == 1 ? p.pad : 1), ix0 = tx0 - (ST == 1 ? p.pad : 1);
    __syncthreads();                               // the previous tile's fragments and P have been read
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 256 * u;
      *(u32x4*)(at + (idx >> 3) * 72 + (idx & 7) * 8) = nxt[u];
    }
    if (tile + (int)gridDim.x < p.ntiles) request(tile + gridDim.x);
    __syncthreads();
    for (int mb = wave; mb < MB; mb += 4) {
      f32x16 acc[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 fa = *(const u32x4*)(at + (mb * 32 + l32) * 72 + 16 * j + 8 * h);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (jj * 32 < NC) {
            const u32x4 fb = *(const u32x4*)(wl + (jj * 32 + l32) * 72 + 16 * j + 8 * h);
            acc[jj] = mfma16(fa, fb, acc[jj]);
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int col = jj * 32 + l32;
        if (col < NC) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int q = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (q < M) P[q * PW + col] = acc[jj][r];
          }
        }
      }
    }
    __syncthreads();
    auto finish = [&](f32x4 v, int Y, int X) {
      bf16x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float f = (float)(bf16_t)(v[c] + bv[c]);
        if (p.act == VG_ACT_LRELU) f = vg_lrelu(f);
        else if (p.act == VG_ACT_SIGMOID) f = vg_sigmoid(f);
        if (c >= p.nt) f = 0.f;
        o[c] = (bf16_t)f;
      }
      og[((size_t)img * p.TH_ + Y) * p.TW_ + X] = __builtin_bit_cast(u32x2, o);
    };
    if constexpr (ST == 1) {
      if (tid < 128) {                              // one output pixel per thread: 9 16-byte reads of P
        const int oy = tid >> 5, ox = tid & 31;
        const int Y = ty0 + oy, X = tx0 + ox;
        if (Y < p.TH_ && X < p.TW_) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int s = 0; s < R; ++s) v += *(const f32x4*)(P + ((oy + r) * IW + ox + s) * PW + (r * R + s) * 4);
          finish(v, Y, X);
        }
      }
    } else {
      for (int o = tid; o < 8 * 64; o += 256) {
        const int oy = o >> 6, ox = o & 63;
        const int Y = 2 * ty0 + oy, X = 2 * tx0 + ox;
        if (Y >= p.TH_ || X >= p.TW_) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int r0 = (Y + 1) & 1, s0 = (X + 1) & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int r = r0 + 2 * a, s = s0 + 2 * b;
            const int qy = (Y + 1 - r) / 2 - iy0, qx = (X + 1 - s) / 2 - ix0;      // inside the 6 x 34 tile by construction
            v += *(const f32x4*)(P + (qy * IW + qx) * PW + (r * R + s) * 4);
          }
        finish(v, Y, X);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wide -> thin, 3x3 stride 1, without the P image: the taps of a kernel ROW go into the M dimension of a 16x16x32 MFMA.
//   D_hy[(r,c)][px] = sum_{s,n} Wl[(r,s,c)][n] * halo[hy][px + s][n]        A = weights (12 real rows of 16), B = 16 pixels
//   out[y][px][c]   = D_y[(0,c)][px] + D_{y+1}[(1,c)][px] + D_{y+2}[(2,c)][px]
// One MFMA of a halo row serves the three output rows it touches; the accumulator of halo row hy holds its r = 0, 1, 2 parts
// in lanes 0-15, 16-31, 32-47 (register index = c), so an output row is three accumulators met by two 16-lane shifts, and
// lanes 0-15 then hold whole 8-byte output pixels.  Block = 4 waves over an 8 x 32 output tile (halo 10 x 34 pixels x 64
// channels, 144-byte pixel stride: conflict-free 16-byte fragment reads); wave = (16-pixel strip, 4-row half): 36 MFMAs per
// tile, weights in 24 registers.  Persistent blocks, three per CU; the next tile's halo is requested into registers before
// this one is computed.  (The GEMM + col2im kernel above spent three barriers and a 29 KB f32 round trip through LDS per
// 4 x 32 tile: 248 us per launch at 64 x 256 x 256, against ~105 us for its one pass over the 64-channel tensor.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void thin_out3_kernel(const ThinParams p) {
  constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, M = IH * IW, PXB = 144, NLD = (M * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];      // [M][PXB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  const int strip = wave & 1, half = wave >> 1;
  const int tpi = p.tilesX * p.tilesY;
  u32x4 wa[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      wa[s][kc] = u32x4{0u, 0u, 0u, 0u};
      if (l16 < 12) wa[s][kc] = *(const u32x4*)(p.wpk + (((l16 >> 2) * 3 + s) * 4 + (l16 & 3)) * 72 + kc * 32 + kg * 8);
    }
  float bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bv[c] = (p.bias != nullptr && c < p.nt) ? p.bias[c] : 0.f;
  u32x2* __restrict__ og = (u32x2*)p.out;

  u32x4 nxt[NLD];
  auto request = [&](int tile) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int iy0 = (trem / p.tilesX) * TH - p.pad, ix0 = (trem % p.tilesX) * TW - p.pad;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 256 * u;
      const int q = idx >> 3, pc = idx & 7;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < M) {
        const int hy = q / IW, hx = q - hy * IW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        if ((unsigned)iy < (unsigned)p.WH && (unsigned)ix < (unsigned)p.WW && pc * 8 < p.Ws)
          v = *(const u32x4*)(p.wide + (((size_t)img * p.WH + iy) * p.WW + ix) * p.Ws + pc * 8);
      }
      nxt[u] = v;
    }
  };
  int tile = blockIdx.x;
  if (tile < p.ntiles) request(tile);
  const unsigned char* bbase = tsm + ((half * 4 * IW + strip * 16 + l16) * PXB + kg * 16);
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int ty0 = (trem / p.tilesX) * TH, tx0 = (trem % p.tilesX) * TW;
    __syncthreads();                               // the previous tile's fragments have been read
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 256 * u;
      if (idx < M * 8) *(u32x4*)(tsm + (idx >> 3) * PXB + (idx & 7) * 16) = nxt[u];
    }
    if (tile + (int)gridDim.x < p.ntiles) request(tile + gridDim.x);
    __syncthreads();
    f32x4 acc[6];
#pragma unroll
    for (int hr = 0; hr < 6; ++hr) {
      acc[hr] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const u32x4 b = *(const u32x4*)(bbase + (hr * IW + s) * PXB + kc * 64);
          acc[hr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[s][kc]), __builtin_bit_cast(bf16x8, b),
                                                            acc[hr], 0, 0, 0);
        }
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = acc[y][c] + __shfl_down(acc[y + 1][c], 16, 64) + __shfl_down(acc[y + 2][c], 32, 64);
      const int Y = ty0 + half * 4 + y, X = tx0 + strip * 16 + lane;
      if (lane < 16 && Y < p.TH_ && X < p.TW_) {
        bf16x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float f = (float)(bf16_t)(v[c] + bv[c]);
          if (p.act == VG_ACT_LRELU) f = vg_lrelu(f);
          else if (p.act == VG_ACT_SIGMOID) f = vg_sigmoid(f);
          if (c >= p.nt) f = 0.f;
          o[c] = (bf16_t)f;
        }
        og[((size_t)img * p.TH_ + Y) * p.TW_ + X] = __builtin_bit_cast(u32x2, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  acc[n][(r,s,c)] = sum_q wide[q][n] * thin[q*ST - pad + (r,s)][c] over all pixels q of the wide
// tensor: K = pixels.  Persistent blocks walk 8 x 32 pixel tiles.  Both operands stay [pixel][channel] in LDS, as they
// are in memory, and are read with transposed LDS reads (ds_read_b64_tr_b16, as in wgrad3): the wide tile as two
// 32-channel images of plain 64-byte pixel rows, the thin operand straight from the 8-byte pixels of the halo, every
// lane addressing the pixel its tap (r, s) shifts to.  (Round 2 wrote both operands transposed into LDS with 2-byte
// stores, 100 per thread and tile, and was bound by them: 173 us for 570 MB.)
// Every block writes one 64 x 64 partial image; thin_wgrad_finish sums them in a fixed order into OIHW.
// ---------------------------------------------------------------------------------------------
template <int R, int ST>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinParams p) {
  constexpr int TH = 8, TW = 32, HH = (TH - 1) * ST + R, HW = (TW - 1) * ST + R, HWP = HW + (HW & 1) + 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  unsigned char* wI = tsm;                       // [2 chunks of 32 channels][256 pixels][64 B]
  u32x2* halo = (u32x2*)(tsm + 2 * 256 * 64);    // [HH][HWP] 4-channel pixels
  typedef __attribute__((address_space(3))) bf16x4* tr_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int tpi = p.tilesX * p.tilesY;
  const u32x2* __restrict__ tg = (const u32x2*)p.thin;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int mblk = wave >> 1, nblk = wave & 1;
  // transposed-read lane terms: lane (tq, tp) of the 16-lane group (h, g16) addresses pixel 8h + tq of the K step.
  // A: channels 16*g16 + 4*tp .. + 3 of its 32-channel image.  B: the 8-byte pixel that tap (r, s) = (2*nblk + g16, tp)
  // shifts to (columns 16*g16 + 4*tp .. + 3 of the block = that tap's four channels); taps beyond R x R address the
  // last real one (their accumulator columns are never read by the finish kernel).
  const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
  const int a_l = mblk * (256 * 64) + (8 * h + tq) * 64 + (16 * g16 + 4 * tp) * 2;
  const int tr_ = min(2 * nblk + g16, R - 1), ts_ = min(tp, R - 1);
  const int b_l = (tr_ * HWP + (8 * h + tq) * ST + ts_) * 8;
  // the NEXT tile's pixels are requested into registers before the current tile is computed (HBM latency under the
  // LDS writes and MFMAs)
  constexpr int NH = (HH * HWP + 255) / 256;
  u32x4 nw_[8];
  u32x2 nh_[NH];
  auto request = [&](int tile) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int ty0 = (trem / p.tilesX) * TH, tx0 = (trem % p.tilesX) * TW;
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int idx = tid + 256 * u;
      const int hy = idx / HWP, hx = idx - hy * HWP;
      const int iy = ty0 * ST - p.pad + hy, ix = tx0 * ST - p.pad + hx;
      u32x2 v = {0u, 0u};
      if (idx < HH * HWP && (unsigned)iy < (unsigned)p.TH_ && (unsigned)ix < (unsigned)p.TW_) v = tg[((size_t)img * p.TH_ + iy) * p.TW_ + ix];
      nh_[u] = v;
    }
    // wide tile: load j of thread tid is 16-byte piece (j * 256 + tid) % 8 of pixel (j * 256 + tid) / 8: the 64 lanes of
    // a load read 1 KB of consecutive memory (a pixel per thread made every load touch 64 lines for 16 bytes each)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = j * 256 + tid, pix = g >> 3, pc = g & 7;
      const int qy = ty0 + (pix >> 5), qx = tx0 + (pix & 31);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qy < p.WH && qx < p.WW && pc * 8 < p.Ws) v = *(const u32x4*)(p.wide + (((size_t)img * p.WH + qy) * p.WW + qx) * p.Ws + pc * 8);
      nw_[j] = v;
    }
  };
  int tile = blockIdx.x;
  if (tile < p.ntiles) request(tile);
  for (; tile < p.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's fragments have been read
#pragma unroll
    for (int u = 0; u < NH; ++u)
      if (tid + 256 * u < HH * HWP) halo[tid + 256 * u] = nh_[u];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = j * 256 + tid, pix = g >> 3, pc = g & 7;
      *(u32x4*)(wI + (pc >> 2) * (256 * 64) + pix * 64 + (pc & 3) * 16) = nw_[j];
    }
    if (tile + (int)gridDim.x < p.ntiles) request(tile + gridDim.x);
    __syncthreads();
    const unsigned char* hb = (const unsigned char*)halo;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {                  // 16 pixels of tile row ks / 2 per step
      const int ao = a_l + ks * 16 * 64;
      const int bo = b_l + ((ks >> 1) * ST * HWP + (ks & 1) * 16 * ST) * 8;
      const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(wI + ao));
      const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(wI + ao + 4 * 64));
      const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(hb + bo));
      const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tr_ptr_t)(hb + bo + 4 * ST * 8));
      const bf16x8 fa = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
      const bf16x8 fb = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
      acc = mfma16(__builtin_bit_cast(u32x4, fa), __builtin_bit_cast(u32x4, fb), acc);
    }
  }
  float* __restrict__ part = p.part + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) part[(mblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + nblk * 32 + l32] = acc[r];
}

// dW[o][i][r][s] (+)= sum over the partial images, fixed order.  wmode 0: acc[n = o][(r,s,c = i)]; 1: acc[n = i][(R-1-r, R-1-s, c = o)]
// A block takes 16 outputs; its 16 thread groups split the partial images (group g: g, g + 16, ...; eight loads in
// flight), LDS combines the groups in order: a few memory latencies per launch instead of nparts / 8 in a chain.
__global__ __launch_bounds__(256) void thin_wgrad_finish_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw,
                                                                int O, int I, int R, int wmode, int accumulate) {
  __shared__ float red[16][16];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + e;
  const bool ok = t < O * I * R * R;
  float v = 0.f;
  if (ok) {
    const int s = t % R, r = (t / R) % R, i = (t / (R * R)) % I, o = t / (R * R * I);
    const int idx = wmode == 0 ? o * 64 + (r * 4 + s) * 4 + i : i * 64 + ((R - 1 - r) * 4 + (R - 1 - s)) * 4 + o;
    int b = g;
    for (; b + 7 * 16 < nparts; b += 8 * 16) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = part[(size_t)(b + 16 * u) * 4096 + idx];
      v += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (; b < nparts; b += 16) v += part[(size_t)b * 4096 + idx];
  }
  red[g][e] = v;
  __syncthreads();
  if (g == 0 && ok) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += red[q][e];
    dw[t] = accumulate ? dw[t] + sum : sum;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// 0: not a thin layer; 1: thin INPUT (ci <= 3 on a 4-channel NHWC tensor, <= 64 output channels on a 64-stride
// tensor; 3x3 stride 1 pad 1, or 4x4 stride 2 pad 1); 2: thin OUTPUT (co <= 3, <= 64 input channels; 3x3 stride 1 pad 1)
extern "C" int vg_thin_conv_supported(const vg_conv_desc* d) {
  if (!d || d->dtype != VG_BF16 || d->r != d->s || d->n <= 0) return 0;
  const bool k3 = d->r == 3 && d->stride == 1 && d->pad == 1;
  const bool k4 = d->r == 4 && d->stride == 2 && d->pad == 1 && d->h % 2 == 0 && d->w % 2 == 0;
  if (d->ci <= 3 && d->xcs == 4 && d->co <= 64 && d->ycs == 64 && (k3 || k4)) return 1;
  if (d->co <= 3 && d->ycs == 4 && d->ci <= 64 && d->xcs == 64 && k3) return 2;
  return 0;
}

static void thin_fill(const vg_conv_desc* d, int kind, ThinParams& p) {
  memset(&p, 0, sizeof(p));
  p.N = d->n; p.O = d->co; p.I = d->ci; p.R = d->r; p.stride = d->stride; p.Ws = 64;
  if (kind == 1) { p.TH_ = d->h; p.TW_ = d->w; p.WH = d->oh; p.WW = d->ow; p.nw = d->co; p.nt = d->ci; }
  else { p.TH_ = d->oh; p.TW_ = d->ow; p.WH = d->h; p.WW = d->w; p.nw = d->ci; p.nt = d->co; }
}
template <int R> static constexpr int thin_out_lds() { return (7 * 32 * 72 + 64 * 72) * 2 + 6 * 34 * (R * R * 4 + 4) * 4; }
template <int R, int ST> static constexpr int thin_wgrad_lds() {      // the wide tile as two 32-channel images + the thin halo
  return 2 * 256 * 64 + ((8 - 1) * ST + R) * ((32 - 1) * ST + R + (((32 - 1) * ST + R) & 1) + 2) * 8;
}
template <typename K> static int thin_lds_attr(K kern, int bytes) {
  VG_HIP(vg_lds_attr((const void*)kern, bytes));
  return VG_OK;
}
static void thin_tiles(ThinParams& p, int gh, int gw, int th, int tw) {
  p.tilesX = (int)vg_ceil_div(gw, tw); p.tilesY = (int)vg_ceil_div(gh, th);
  p.ntiles = p.N * p.tilesX * p.tilesY;
}

extern "C" size_t vg_thin_conv_ws_bytes(void) { return 64 * 72 * sizeof(bf16_t); }

static int thin_out_blocks(const ThinParams& p) { return p.ntiles < 512 ? p.ntiles : 512; }      // 2 resident blocks per CU

// 3x3 stride-1 wide -> thin: the row-tap kernel (thin_out3_kernel); VG_THIN_OUT_OLD=1 keeps the GEMM + col2im kernel (A/B)
static int launch_thin_out3(ThinParams& p, hipStream_t st) {
  static const bool old = getenv("VG_THIN_OUT_OLD") != nullptr;
  if (old) {
    thin_tiles(p, p.TH_, p.TW_, 4, 32);
    int rc = thin_lds_attr(thin_out_kernel<3, 1>, thin_out_lds<3>());
    if (rc) return rc;
    hipLaunchKernelGGL((thin_out_kernel<3, 1>), dim3(thin_out_blocks(p)), dim3(256), thin_out_lds<3>(), st, p);
    return VG_OK;
  }
  constexpr int lds = 10 * 34 * 144;
  thin_tiles(p, p.TH_, p.TW_, 8, 32);
  int rc = thin_lds_attr(thin_out3_kernel, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(thin_out3_kernel, dim3(p.ntiles < 768 ? p.ntiles : 768), dim3(256), lds, st, p);      // 3 resident blocks per CU
  return VG_OK;
}

extern "C" int vg_thin_conv_fwd(const vg_conv_desc* d, const void* x, const float* w_oihw, const float* bias, void* y,
                                double* stats, int act, void* ws, void* stream) {
  const int kind = vg_thin_conv_supported(d);
  VG_REQUIRE(kind != 0, "thin_conv_fwd: not a supported thin layer");
  VG_REQUIRE(x && w_oihw && y && ws, "thin_conv_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  ThinParams p;
  thin_fill(d, kind, p);
  p.w = w_oihw; p.bias = bias; p.act = act; p.wmode = 0; p.pad = d->pad; p.out = (bf16_t*)y; p.wpk = (const bf16_t*)ws;
  hipLaunchKernelGGL(thin_pack_kernel, dim3(64 * 72 / 256), dim3(256), 0, st, p, (bf16_t*)ws, kind == 1 ? 1 : 0);
  VG_LAUNCH_CHECK();
  if (kind == 1) {
    p.thin = (const bf16_t*)x; p.stats = stats; p.cpad = d->co <= 32 ? 32 : 64; p.det = vg_det_mode();
    thin_tiles(p, p.WH, p.WW, 8, 32);
    if (stats) VG_DET_GUARD(p.ntiles);
    if (d->r == 3) hipLaunchKernelGGL((thin_in_kernel<3, 1>), dim3(p.ntiles), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((thin_in_kernel<4, 2>), dim3(p.ntiles), dim3(256), 0, st, p);
  } else {
    VG_REQUIRE(stats == nullptr, "thin_conv_fwd: statistics are only taken on the wide side");
    p.wide = (const bf16_t*)x;
    int rc = launch_thin_out3(p, st);
    if (rc) return rc;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_thin_conv_dgrad(const vg_conv_desc* d, const void* dy, const float* w_oihw, void* dx, void* ws, void* stream) {
  const int kind = vg_thin_conv_supported(d);
  VG_REQUIRE(kind != 0, "thin_conv_dgrad: not a supported thin layer");
  VG_REQUIRE(dy && w_oihw && dx && ws, "thin_conv_dgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  ThinParams p;
  thin_fill(d, kind, p);
  p.w = w_oihw; p.act = VG_ACT_NONE; p.out = (bf16_t*)dx; p.wpk = (const bf16_t*)ws;
  p.wmode = kind == 2 ? 1 : (d->r == 3 ? 1 : 2);
  hipLaunchKernelGGL(thin_pack_kernel, dim3(64 * 72 / 256), dim3(256), 0, st, p, (bf16_t*)ws, kind == 2 ? 1 : 0);
  VG_LAUNCH_CHECK();
  if (kind == 1) {          // dx thin <- dy wide
    p.wide = (const bf16_t*)dy;
    if (d->r == 3) {
      p.wmode = 1; p.pad = d->r - 1 - d->pad;
      int rc = launch_thin_out3(p, st);
      if (rc) return rc;
    } else {
      p.wmode = 2; p.pad = 1;
      thin_tiles(p, p.WH, p.WW, 4, 32);          // tiles of the dy grid; each covers 8 x 64 pixels of dx
      int rc = thin_lds_attr(thin_out_kernel<4, 2>, thin_out_lds<4>());
      if (rc) return rc;
      hipLaunchKernelGGL((thin_out_kernel<4, 2>), dim3(thin_out_blocks(p)), dim3(256), thin_out_lds<4>(), st, p);
    }
  } else {                  // dx wide <- dy thin
    p.thin = (const bf16_t*)dy; p.wmode = 1; p.pad = d->r - 1 - d->pad;
    thin_tiles(p, p.WH, p.WW, 8, 32);
    hipLaunchKernelGGL((thin_in_kernel<3, 1>), dim3(p.ntiles), dim3(256), 0, st, p);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

static int thin_wgrad_blocks(const ThinParams& p) { return p.ntiles < 512 ? p.ntiles : 512; }      // (768 / 1024: no faster)

extern "C" size_t vg_thin_conv_wgrad_ws_bytes(const vg_conv_desc* d) {
  const int kind = vg_thin_conv_supported(d);
  if (!kind) return 0;
  ThinParams p;
  thin_fill(d, kind, p);
  thin_tiles(p, p.WH, p.WW, 8, 32);
  return (size_t)thin_wgrad_blocks(p) * 4096 * sizeof(float);
}

extern "C" int vg_thin_conv_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws,
                                  int accumulate, void* stream) {
  const int kind = vg_thin_conv_supported(d);
  VG_REQUIRE(kind != 0, "thin_conv_wgrad: not a supported thin layer");
  VG_REQUIRE(x && dy && dw_oihw && ws, "thin_conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  ThinParams p;
  thin_fill(d, kind, p);
  p.part = (float*)ws;
  if (kind == 1) { p.thin = (const bf16_t*)x; p.wide = (const bf16_t*)dy; p.wmode = 0; p.pad = d->pad; }
  else { p.thin = (const bf16_t*)dy; p.wide = (const bf16_t*)x; p.wmode = 1; p.pad = d->r - 1 - d->pad; }
  thin_tiles(p, p.WH, p.WW, 8, 32);
  const int blocks = thin_wgrad_blocks(p);
  if (d->r == 3) {
    constexpr int lds = thin_wgrad_lds<3, 1>();
    int rc = thin_lds_attr(thin_wgrad_kernel<3, 1>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((thin_wgrad_kernel<3, 1>), dim3(blocks), dim3(256), lds, st, p);
  } else {
    constexpr int lds = thin_wgrad_lds<4, 2>();
    int rc = thin_lds_attr(thin_wgrad_kernel<4, 2>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((thin_wgrad_kernel<4, 2>), dim3(blocks), dim3(256), lds, st, p);
  }
  VG_LAUNCH_CHECK();
  const int total = d->co * d->ci * d->r * d->r;
  hipLaunchKernelGGL(thin_wgrad_finish_kernel, dim3((total + 15) / 16), dim3(256), 0, st, (const float*)ws, blocks, dw_oihw,
                     d->co, d->ci, d->r, p.wmode, accumulate);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
