// The PatchGAN tail: Conv2d(C, 1, 4, stride 1, padding 1) on a small map (vq_vae_gan/model.py:221,
// beta_vae_gans/model.py:176, hierarchical_vae_gan/model.py:330: 512 -> 1 on 16x16 at 256x256 inputs), bf16.
//
// As a gather-GEMM this layer pads ONE output channel to a 32-wide tile: 0.24 GFLOP took 58 us forward (split-K over
// an f32 workspace, three launches) and 65 us per weight gradient (memset + per-tap kernel + unpack) - launch and
// latency bound, 0.4 ms of a 38 ms step.  It is a dot product per output pixel, so it runs on the vector ALUs here:
//   forward : a wave owns 8 consecutive output pixels of one row; a lane owns 8 channels (per 512-channel round) and
//             keeps their 16 taps in registers; every input pixel the segment touches is loaded once (4 x 11 loads of
//             1 KB per wave) and feeds up to four outputs; one shuffle tree per output at the end.
//   weight gradient : a wave walks input rows; a lane keeps dW of its 8 channels x 16 taps in 128 registers; the <= 64
//             upstream-gradient values a row needs sit one per lane in two registers and are read with v_readlane.
//             The four waves of a block add up through LDS, every block writes one partial image and
//             conv1_wgrad_finish sums those in a fixed order (bit-reproducible).
// The master OIHW f32 weights are read directly (rounded to bf16 like the packed images of the other kernels).
// The input gradient stays on the gather-GEMM (25 us).
#include "common.h"

struct Conv1Params {
  const bf16_t* x;      // [N][H][W][Cs]
  const bf16_t* dy;     // [N][OH][OW][Ys] (channel 0)
  bf16_t* y;            // [N][OH][OW][Ys]
  const float* w;       // [1][C][4][4]
  const float* bias;
  float* part;          // weight gradient: [blocks][Cr*16]
  int N, H, W, C, Cs, OH, OW, Ys, act, rounds, nwaves;
};

__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16_t)v; }

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const Conv1Params p) {
  const int lane = threadIdx.x & 63;
  const int segs = (p.OW + 7) >> 3;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= p.N * p.OH * segs) return;                       // wave-uniform
  const int seg = wv % segs, oy = (wv / segs) % p.OH, img = wv / (segs * p.OH);
  const int ox0 = seg * 8;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  for (int rd = 0; rd < p.rounds; ++rd) {
    const int c0 = (rd * 64 + lane) * 8;
    const bool cok = c0 < p.C;                               // C % 8 == 0
    float wr[16][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f32x4 t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = cok ? *(const f32x4*)(p.w + (size_t)(c0 + e) * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) wr[k][e] = bf16_round(t[k >> 2][k & 3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iy = oy - 1 + r;
      if ((unsigned)iy >= (unsigned)p.H) continue;           // wave-uniform
      const bf16_t* row = p.x + ((size_t)img * p.H + iy) * p.W * p.Cs + c0;
      bf16x8 v[11];
#pragma unroll
      for (int t = 0; t < 11; ++t) {
        const int ix = ox0 - 1 + t;
        v[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (cok && (unsigned)ix < (unsigned)p.W) v[t] = *(const bf16x8*)(row + (size_t)ix * p.Cs);
      }
#pragma unroll
      for (int t = 0; t < 11; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int o = t - s;
          if (o < 0 || o >= 8) continue;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[o] += (float)v[t][e] * wr[r * 4 + s][e];
        }
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[o] += __shfl_xor(acc[o], m, 64);
  }
  // 16-byte piece q of output pixel o: lane = o * pieces + q; channel 0 carries the value, the padding channels zero
  const int pieces = p.Ys >> 3;
  const int o = lane / pieces, q = lane - o * pieces;
  if (o < 8 && ox0 + o < p.OW) {
    float val = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) val = o == k ? acc[k] : val;
    float f = bf16_round(val + (p.bias ? p.bias[0] : 0.f));
    if (p.act == VG_ACT_LRELU) f = vg_lrelu(f);
    else if (p.act == VG_ACT_SIGMOID) f = vg_sigmoid(f);
    bf16x8 out = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q == 0) out[0] = (bf16_t)f;
    *(bf16x8*)(p.y + (((size_t)img * p.OH + oy) * p.OW + ox0 + o) * p.Ys + q * 8) = out;
  }
}

// dW[c][r][s] = sum_{n,oy,ox} dy[n][oy][ox] * x[n][oy-1+r][ox-1+s][c]: walked by INPUT pixel (iy, ix), which meets
// the taps (r, s) with dy[iy+1-r][ix+1-s].  The four waves of a block add their images through LDS in a fixed order
// (waves 3, 2, 1 onto wave 0): one partial image per block.
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const Conv1Params p) {
  __shared__ float red[128 * 64];                              // [register][lane]: conflict-free
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wv = blockIdx.x * 4 + wib;
  const int rows = p.N * p.H;
  for (int rd = 0; rd < p.rounds; ++rd) {
    const int c0 = (rd * 64 + lane) * 8;
    const bool cok = c0 < p.C;
    float acc[16][8];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    for (int row = wv; row < rows; row += p.nwaves) {
      const int img = row / p.H, iy = row - img * p.H;
      // upstream gradient of the four output rows this input row meets: g0 = rows r = 0, 1 (lane = 32 r + ox), g1 = r = 2, 3
      float g[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int r = 2 * hh + (lane >> 5), ox = lane & 31, oy = iy + 1 - r;
        g[hh] = ((unsigned)oy < (unsigned)p.OH && ox < p.OW) ? (float)p.dy[(((size_t)img * p.OH + oy) * p.OW + ox) * p.Ys] : 0.f;
      }
      const bf16_t* xr = p.x + ((size_t)img * p.H + iy) * p.W * p.Cs + c0;
      // four pixels per group, the next group's loads in flight under this group's FMAs
      bf16x8 cur[4], nxt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        cur[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (cok && u < p.W) cur[u] = *(const bf16x8*)(xr + (size_t)u * p.Cs);
      }
      for (int ix0 = 0; ix0 < p.W; ix0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          nxt[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          if (cok && ix0 + 4 + u < p.W) nxt[u] = *(const bf16x8*)(xr + (size_t)(ix0 + 4 + u) * p.Cs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ix = ix0 + u;                              // pixels past the row end hold zeros
          float vf[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) vf[e] = (float)cur[u][e];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int ox = ix + 1 - s;                       // wave-uniform
              float gv = 0.f;
              if ((unsigned)ox < (unsigned)p.OW)
                gv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g[r >> 1]), (r & 1) * 32 + ox));
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[r * 4 + s][e] += gv * vf[e];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
      }
    }
    for (int src = 3; src >= 1; --src) {
      __syncthreads();
      if (wib == src) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) red[(k * 8 + e) * 64 + lane] = acc[k][e];
      }
      __syncthreads();
      if (wib == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[k][e] += red[(k * 8 + e) * 64 + lane];
      }
    }
    __syncthreads();
    if (cok && wib == 0) {
      float* dst = p.part + (size_t)blockIdx.x * p.rounds * 8192 + (size_t)c0 * 16;      // [block][channel][tap]
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(dst + e * 16 + 4 * q) = f32x4{acc[4 * q][e], acc[4 * q + 1][e], acc[4 * q + 2][e], acc[4 * q + 3][e]};
    }
  }
}

// 64 outputs x 16 groups per block: group q sums partials q, q + 16, ... in order; the 16 group sums meet in LDS and are
// added in order.
__global__ __launch_bounds__(1024) void conv1_wgrad_finish_kernel(const float* __restrict__ part, int nparts, size_t stride,
                                                                  float* __restrict__ dw, int total, int accumulate) {
  __shared__ float red[16][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + o;
  float v = 0.f;
  if (t < total) {
    int b = q;
    for (; b + 48 < nparts; b += 64) {
      const float a0 = part[(size_t)b * stride + t], a1 = part[(size_t)(b + 16) * stride + t];
      const float a2 = part[(size_t)(b + 32) * stride + t], a3 = part[(size_t)(b + 48) * stride + t];
      v += (a0 + a1) + (a2 + a3);
    }
    for (; b < nparts; b += 16) v += part[(size_t)b * stride + t];
  }
  red[q][o] = v;
  __syncthreads();
  if (q == 0 && t < total) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += red[k][o];
    dw[t] = accumulate ? dw[t] + r : r;
  }
}

// ---------------------------------------------------------------------------------------------
extern "C" int vg_conv1_supported(const vg_conv_desc* d) {
  if (!d || d->dtype != VG_BF16 || d->n <= 0) return 0;
  return d->co == 1 && d->r == 4 && d->s == 4 && d->stride == 1 && d->pad == 1 && d->ci % 8 == 0 && d->xcs % 8 == 0 &&
         d->xcs >= d->ci && d->ycs % 8 == 0 && d->ycs >= 8 && d->ycs <= 64 && d->ow <= 32 && d->oh >= 1 && d->ow >= 1;
}

static void conv1_fill(const vg_conv_desc* d, Conv1Params& p) {
  memset(&p, 0, sizeof(p));
  p.N = d->n; p.H = d->h; p.W = d->w; p.C = d->ci; p.Cs = d->xcs; p.OH = d->oh; p.OW = d->ow; p.Ys = d->ycs;
  p.rounds = (d->ci + 511) / 512;
}

extern "C" int vg_conv1_fwd(const vg_conv_desc* d, const void* x, const float* w_oihw, const float* bias, void* y, int act,
                            void* stream) {
  VG_REQUIRE(vg_conv1_supported(d), "conv1_fwd: not a supported one-channel 4x4 stride-1 conv");
  VG_REQUIRE(x && w_oihw && y, "conv1_fwd: null pointer");
  Conv1Params p;
  conv1_fill(d, p);
  p.x = (const bf16_t*)x; p.w = w_oihw; p.bias = bias; p.y = (bf16_t*)y; p.act = act;
  const long waves = (long)p.N * p.OH * ((p.OW + 7) / 8);
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

static int conv1_wgrad_waves(const vg_conv_desc* d) {
  const long rows = (long)d->n * d->h;
  long w = rows < 1024 ? rows : 1024;
  return (int)((w + 3) / 4 * 4);
}

extern "C" size_t vg_conv1_wgrad_ws_bytes(const vg_conv_desc* d) {
  if (!vg_conv1_supported(d)) return 0;
  return (size_t)(conv1_wgrad_waves(d) / 4) * ((d->ci + 511) / 512) * 8192 * sizeof(float);
}

extern "C" int vg_conv1_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws, int accumulate,
                              void* stream) {
  VG_REQUIRE(vg_conv1_supported(d), "conv1_wgrad: not a supported one-channel 4x4 stride-1 conv");
  VG_REQUIRE(x && dy && dw_oihw && ws, "conv1_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  Conv1Params p;
  conv1_fill(d, p);
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.part = (float*)ws; p.nwaves = conv1_wgrad_waves(d);
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(p.nwaves / 4), dim3(256), 0, st, p);
  VG_LAUNCH_CHECK();
  const int total = d->ci * 16;
  hipLaunchKernelGGL(conv1_wgrad_finish_kernel, dim3((total + 63) / 64), dim3(1024), 0, st, (const float*)ws, p.nwaves / 4,
                     (size_t)p.rounds * 8192, dw_oihw, total, accumulate);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
