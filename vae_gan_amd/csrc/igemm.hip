// Implicit-GEMM convolution kernels for gfx950 (MI355X).
//
// One gather-GEMM kernel serves nn.Conv2d forward, its input gradient and nn.ConvTranspose2d
// forward (stride-s transposed convs are split into s*s output phases, each a stride-1 gather
// with its own taps), and one pixel-reduction kernel serves the weight gradient.  Activations
// are NHWC; the GEMM K dimension walks (tap, channel) in 64-byte chunks (32 bf16 / 16 f32), so
// the bf16 path (v_mfma_f32_32x32x16_bf16) and the exact-f32 path (v_mfma_f32_32x32x2_f32) share
// all staging code: both consume 16-byte fragments.
//
// LDS tiles are [row][64 B] with the 16-byte slot XOR-swizzled by (row>>2)&3, which makes both
// the ds_write_b128 staging stores and the ds_read_b128 fragment loads bank-conflict free.
#include "common.h"

#define VG_MAX_PHASES 4
#define VG_MAX_TAPS 16

struct IgemmParams {
  const void* x;
  const void* w;       // [phase][Npad][ntaps*C]
  const float* bias;   // [Nout] or null
  void* y;
  double* stats;       // [VG_STATS_REPLICAS][2][Npad] or null
  int N, H, W, C, Cs;  // gathered tensor: dims, reduction channels (multiple of KC), channel stride
  int OHl, OWl;        // logical output grid of one phase
  int OH, OW, Ks;      // physical output tensor
  int Nout, Npad;      // valid / padded GEMM-N (output channels)
  int in_s, out_s;     // gather stride, scatter stride
  int ntaps, nphases;
  int act, accumulate;
  int ksplit;          // >1: grid.z slices of the K loop; raw f32 partials go to splitws by atomics
  float* splitws;      // [pixels][Npad] f32, zeroed by the launcher
  int M;               // N*OHl*OWl
  int mtiles, ntiles;
  signed char dy[VG_MAX_PHASES][VG_MAX_TAPS];
  signed char dx[VG_MAX_PHASES][VG_MAX_TAPS];
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static inline f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 bytes = 4 f32: lane half h holds k = {4*(2j+h)+i}; A and B use the same assignment, so
  // the four k=2 MFMAs below cover 8 distinct k values exactly once.
  __device__ static inline f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], c, 0, 0, 0);
    return c;
  }
};

__device__ __attribute__((aligned(64))) const uint32_t vg_zero_block[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ int lds_off64(int row, int slot) { return row * 64 + (((slot ^ (row >> 2)) & 3) << 4); }

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // bijective: blocks that share an XCD (bid % 8) get one contiguous range of logical ids
  int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ring depth: as many 64-byte-chunk stages as fit next to a second resident block (<= 80 KiB per block)
__host__ __device__ constexpr int igemm_stages(int bm, int bn) { return (bm + bn) * 64 * 4 <= 65536 ? 4 : 3; }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int EB = sizeof(T), KC = 64 / EB, PIECE = 16 / EB;
  constexpr int AR = BM / 64;                   // A rows staged per thread
  constexpr int BR = (BN + 63) / 64;            // B rows staged per thread (BN=32: threads < 128)
  constexpr int STAGE_A = BM * 64, STAGE_B = BN * 64;
  constexpr int NS = igemm_stages(BM, BN);      // LDS ring depth (chunks resident or in flight)
  constexpr int STAGE = NS * (STAGE_A + STAGE_B);
  constexpr int TILE = BM * BN * 4;
  constexpr int MAIN = STAGE > TILE ? STAGE : TILE;
  constexpr int LPC = AR + (BN >= 64 ? BR : 0); // LDS-DMA instructions every wave issues per chunk (lower bound)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + NS * STAGE_A;
  unsigned* rowpix = (unsigned*)(smem + MAIN);
  float* red = (float*)(smem + MAIN + BM * 4);   // [WM][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l32 = lane & 31;
  const int lid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  const int phase = blockIdx.y;
  const int py = phase / p.out_s, px = phase % p.out_s;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int ohw = p.OHl * p.OWl;

  // ---- per-row gather metadata (rows are fixed for the whole K loop)
  const int srow = tid >> 2, sslot = tid & 3;
  int a_nh[AR], a_iy0[AR], a_ix0[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + srow + 64 * i;
    if (m < p.M) {
      int n = m / ohw, rem = m - n * ohw;
      int oy = rem / p.OWl, ox = rem - oy * p.OWl;
      a_nh[i] = n * p.H; a_iy0[i] = oy * p.in_s; a_ix0[i] = ox * p.in_s;
    } else {
      a_nh[i] = 0; a_iy0[i] = -(1 << 24); a_ix0[i] = 0;
    }
  }
  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r;
    unsigned v = 0xFFFFFFFFu;
    if (m < p.M) {
      int n = m / ohw, rem = m - n * ohw;
      int oy = rem / p.OWl, ox = rem - oy * p.OWl;
      v = (unsigned)((n * p.OH + oy * p.out_s + py) * p.OW + ox * p.out_s + px);
    }
    rowpix[r] = v;
  }

  const T* __restrict__ xg = (const T*)p.x;
  const int cchunks = p.C / KC;
  const int nk_all = p.ntaps * cchunks;
  const int kbeg = (int)((long)nk_all * blockIdx.z / p.ksplit), kend = (int)((long)nk_all * (blockIdx.z + 1) / p.ksplit);
  const int nk = kend - kbeg;
  const size_t ktot = (size_t)p.ntaps * p.C;
  const T* __restrict__ wg = (const T*)p.w + ((size_t)phase * p.Npad + n0) * ktot;

  // Global -> LDS staging uses LDS-DMA (global_load_lds_dwordx4): one wave instruction writes 1 KiB
  // = 16 rows x 64 B linearly, so thread (srow, sslot) lands at physical slot sslot of its row and
  // must FETCH the logical slot sslot ^ ((row>>2)&3) (swizzle applied on the source address).
  // Padding / out-of-range rows read a 64-byte zero block instead.
  const int lslot = (sslot ^ (srow >> 2)) & 3;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  int tap = kbeg / cchunks, cc = kbeg - tap * cchunks;   // position of the NEXT chunk to load
  auto load_chunk = [&](int buf) {
    const int dy = p.dy[phase][tap], dx = p.dx[phase][tap];
    const int coff = cc * KC + lslot * PIECE;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const void* src = ok ? (const void*)(xg + ((size_t)(a_nh[i] + iy) * p.W + ix) * p.Cs + coff) : (const void*)vg_zero_block;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (lds_ptr_t)(sA + buf * STAGE_A + (wave * 16 + 64 * i) * 64), 16, 0, 0);
    }
    const size_t koff = (size_t)tap * p.C + coff;
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      int r = srow + 64 * i;
      if (BN >= 64 || wave < 2)   // BN == 32: only rows 0..31 exist (waves 0 and 1), wave-uniform branch
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)r * ktot + koff),
                                         (lds_ptr_t)(sB + buf * STAGE_B + (wave * 16 + 64 * i) * 64), 16, 0, 0);
    }
    if (++cc == cchunks) { cc = 0; ++tap; }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Ring pipeline: chunks ks .. ks+NS-2 are in flight or resident; the DMA for chunk ks+NS-1 is issued
  // right after the barrier of step ks into the buffer every wave finished reading in step ks-1.
  // Waits are counted (never drained inside the steady state) and the barrier is a raw s_barrier so the
  // outstanding LDS-DMA stays in flight across it.
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) load_chunk(s0);
  int rbuf = 0, wbuf = NS - 1;
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + NS - 2 < nk) wait_vmcnt<(NS - 2) * LPC>();   // chunk ks landed (this wave's share)
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + NS - 1 < nk) load_chunk(wbuf);
    const unsigned char* cA = sA + rbuf * STAGE_A;
    const unsigned char* cB = sB + rbuf * STAGE_B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const u32x4*)(cA + lds_off64((wm * MI + i) * 32 + l32, 2 * j + h));
#pragma unroll
      for (int i = 0; i < NI; ++i) fb[i] = *(const u32x4*)(cB + lds_off64((wn * NI + i) * 32 + l32, 2 * j + h));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) acc[i][jj] = Mma<T>::run(fa[i], fb[jj], acc[i][jj]);
    }
    rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
    wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
  }
  __syncthreads();   // all fragment reads done before the epilogue tile aliases the ring

  if (p.ksplit > 1) {
    // split-K: add this slice's raw partial sums into the f32 workspace (bias/activation happen in the
    // finish kernel); one lane half covers 32 consecutive channels = one 128-byte segment per row
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + (wn * NI + j) * 32 + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned pix = rowpix[(wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
          if (pix != 0xFFFFFFFFu) unsafeAtomicAdd(&p.splitws[(size_t)pix * p.Npad + n], acc[i][j][r]);
        }
      }
    return;
  }

  // ---- epilogue: bias, stats (on dtype-rounded pre-activation), activation, LDS-staged stores
  float* tile = (float*)smem;   // [BM][BN] f32, aliases the staging buffers (all reads done: synced)
  float s1[NI], s2[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = (wn * NI + j) * 32 + l32;
    const int n = n0 + col;
    const float bv = (p.bias != nullptr && n < p.Nout) ? p.bias[n] : 0.f;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][j][r] + bv;
        v = Elt<T>::to_f(Elt<T>::from_f(v));
        if (m0 + row < p.M) { a1 += v; a2 += v * v; }
        if (p.act == VG_ACT_LRELU) v = vg_lrelu(v);
        else if (p.act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
        if (n >= p.Nout) v = 0.f;
        tile[row * BN + col] = v;
      }
    }
    s1[j] = a1 + __shfl_xor(a1, 32, 64);
    s2[j] = a2 + __shfl_xor(a2, 32, 64);
  }
  if (p.stats != nullptr && h == 0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = (wn * NI + j) * 32 + l32;
      red[(wm * BN + col) * 2 + 0] = s1[j];
      red[(wm * BN + col) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  if (p.stats != nullptr && tid < BN && n0 + tid < p.Nout) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
    const int rep = (mtile + phase) % VG_STATS_REPLICAS;
    unsafeAtomicAdd(&p.stats[((size_t)rep * 2 + 0) * p.Npad + n0 + tid], (double)t1);
    unsafeAtomicAdd(&p.stats[((size_t)rep * 2 + 1) * p.Npad + n0 + tid], (double)t2);
  }
  T* __restrict__ yg = (T*)p.y;
  constexpr int PPR = BN / PIECE;
  for (int q = tid; q < BM * PPR; q += 256) {
    const int row = q / PPR, pc = q - row * PPR;
    const unsigned pix = rowpix[row];
    const int n = n0 + pc * PIECE;
    if (pix == 0xFFFFFFFFu || n >= p.Ks) continue;
    Piece<T> o;
    const float* src = tile + row * BN + pc * PIECE;
    T* dst = yg + (size_t)pix * p.Ks + n;
    if (p.accumulate) {
      Piece<T> old;
      old.v = *(const decltype(old.v)*)dst;
#pragma unroll
      for (int e = 0; e < PIECE; ++e) o.set(e, src[e] + old.get(e));
    } else {
#pragma unroll
      for (int e = 0; e < PIECE; ++e) o.set(e, src[e]);
    }
    *(decltype(o.v)*)dst = o.v;
  }
}

template <typename T, int MI, int NI, int WM, int WN>
static int launch_igemm(const IgemmParams& p0, hipStream_t st) {
  IgemmParams p = p0;
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int STAGE = igemm_stages(BM, BN) * (BM * 64 + BN * 64), TILE = BM * BN * 4;
  constexpr int MAIN = STAGE > TILE ? STAGE : TILE;
  const size_t lds = MAIN + BM * 4 + WM * BN * 2 * 4;
  p.mtiles = (int)vg_ceil_div(p.M, BM);
  p.ntiles = p.Npad / BN;
  auto kern = igemm_kernel<T, MI, NI, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    VG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(p.mtiles * p.ntiles, p.nphases, p.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

template <typename T> static int dispatch_igemm(const IgemmParams& p, hipStream_t st) {
  if (p.Npad % 128 == 0) return launch_igemm<T, 2, 2, 2, 2>(p, st);   // 128 x 128
  if (p.Npad == 64) return launch_igemm<T, 2, 2, 4, 1>(p, st);        // 256 x 64
  if (p.Npad == 32) return launch_igemm<T, 2, 1, 4, 1>(p, st);        // 256 x 32
  vg_set_error("igemm: unsupported padded N %d", p.Npad);
  return VG_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// geometry helpers shared with pack.hip
// ---------------------------------------------------------------------------------------------
int vg_npad(int n) { return n <= 32 ? 32 : (n <= 64 ? 64 : (int)vg_ceil_div(n, 128) * 128); }

static int kc_of(int dtype) { return dtype == VG_BF16 ? 32 : 16; }

// Tap table of the transposed (dgrad / convT-forward) formulation: output row Y = y*s + py takes
// input rows iy = y + dy with kernel row r, (py + pad - r) % s == 0, dy = (py + pad - r)/s.
int vg_transposed_taps(const vg_conv_desc* d, int phase, signed char* dy, signed char* dx, int* rs) {
  const int s = d->stride, py = phase / s, px = phase % s;
  int n = 0;
  for (int r = 0; r < d->r; ++r) {
    if ((py + d->pad - r) % s != 0) continue;
    for (int c = 0; c < d->s; ++c) {
      if ((px + d->pad - c) % s != 0) continue;
      if (n >= VG_MAX_TAPS) return -1;
      dy[n] = (signed char)((py + d->pad - r) / s);
      dx[n] = (signed char)((px + d->pad - c) / s);
      if (rs) rs[n] = r * d->s + c;
      ++n;
    }
  }
  return n;
}

static int check_desc(const vg_conv_desc* d) {
  VG_REQUIRE(d != nullptr, "conv desc is null");
  VG_REQUIRE(d->dtype == VG_F32 || d->dtype == VG_BF16, "conv: bad dtype %d", d->dtype);
  const int kc = kc_of(d->dtype);
  VG_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->ci > 0 && d->co > 0, "conv: non-positive dims");
  VG_REQUIRE(d->r > 0 && d->s > 0 && d->r * d->s <= VG_MAX_TAPS, "conv: kernel %dx%d unsupported", d->r, d->s);
  VG_REQUIRE(d->stride >= 1 && d->stride <= 2, "conv: stride %d unsupported", d->stride);
  VG_REQUIRE(d->r % d->stride == 0 || d->stride == 1, "conv: kernel %d not a multiple of stride %d", d->r, d->stride);
  VG_REQUIRE(d->oh == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->ow == (d->w + 2 * d->pad - d->s) / d->stride + 1,
             "conv: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d->oh, d->ow, d->h, d->w, d->r,
             d->stride, d->pad);
  VG_REQUIRE(d->xcs % kc == 0 && d->xcs >= d->ci, "conv: x channel stride %d must be a multiple of %d and >= %d", d->xcs, kc, d->ci);
  VG_REQUIRE(d->ycs % kc == 0 && d->ycs >= d->co, "conv: y channel stride %d must be a multiple of %d and >= %d", d->ycs, kc, d->co);
  VG_REQUIRE((int64_t)d->n * d->h * d->w < (1ll << 31) && (int64_t)d->n * d->oh * d->ow < (1ll << 31), "conv: too many pixels");
  return VG_OK;
}

static int cpad_of(int c, int dtype) { int kc = kc_of(dtype); return (int)vg_ceil_div(c, kc) * kc; }

extern "C" int vg_conv_fwd(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                           double* stats, int act, int accumulate, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(x && w_packed && y, "conv_fwd: null pointer");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w_packed; p.bias = bias; p.y = y; p.stats = stats;
  p.N = d->n; p.H = d->h; p.W = d->w; p.C = cpad_of(d->ci, d->dtype); p.Cs = d->xcs;
  p.OHl = d->oh; p.OWl = d->ow; p.OH = d->oh; p.OW = d->ow; p.Ks = d->ycs;
  p.Nout = d->co; p.Npad = vg_npad(d->co);
  p.in_s = d->stride; p.out_s = 1; p.ntaps = d->r * d->s; p.nphases = 1;
  p.act = act; p.accumulate = accumulate; p.ksplit = 1;
  p.M = d->n * d->oh * d->ow;
  for (int r = 0; r < d->r; ++r)
    for (int s = 0; s < d->s; ++s) { p.dy[0][r * d->s + s] = (signed char)(r - d->pad); p.dx[0][r * d->s + s] = (signed char)(s - d->pad); }
  return d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, (hipStream_t)stream) : dispatch_igemm<float>(p, (hipStream_t)stream);
}

// ---- split-K forward: few output tiles but a long reduction (Linear layers viewed as 1x1 convs over
// [B,1,1,F] tensors, the PatchGAN tail).  Partials are summed in an f32 workspace, then finished.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                            T* __restrict__ y, long pixels, int Npad, int Ks, int Nout, int act) {
  const long total = pixels * Ks;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long pix = t / Ks;
    const int n = (int)(t - pix * Ks);
    float v = 0.f;
    if (n < Nout) {
      v = ws[pix * Npad + n] + (bias ? bias[n] : 0.f);
      v = Elt<T>::to_f(Elt<T>::from_f(v));
      if (act == VG_ACT_LRELU) v = vg_lrelu(v);
      else if (act == VG_ACT_SIGMOID) v = vg_sigmoid(v);
    }
    y[t] = Elt<T>::from_f(v);
  }
}

extern "C" size_t vg_conv_splitk_ws_bytes(const vg_conv_desc* d) {
  if (!d) return 0;
  return (size_t)d->n * d->oh * d->ow * vg_npad(d->co) * sizeof(float);
}

extern "C" int vg_conv_fwd_splitk(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                                  int act, int ksplit, void* ws, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(x && w_packed && y && ws, "conv_fwd_splitk: null pointer");
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w_packed; p.bias = nullptr; p.y = y; p.stats = nullptr;
  p.N = d->n; p.H = d->h; p.W = d->w; p.C = cpad_of(d->ci, d->dtype); p.Cs = d->xcs;
  p.OHl = d->oh; p.OWl = d->ow; p.OH = d->oh; p.OW = d->ow; p.Ks = d->ycs;
  p.Nout = d->co; p.Npad = vg_npad(d->co);
  p.in_s = d->stride; p.out_s = 1; p.ntaps = d->r * d->s; p.nphases = 1;
  p.M = d->n * d->oh * d->ow;
  for (int r = 0; r < d->r; ++r)
    for (int s = 0; s < d->s; ++s) { p.dy[0][r * d->s + s] = (signed char)(r - d->pad); p.dx[0][r * d->s + s] = (signed char)(s - d->pad); }
  const int nk = p.ntaps * (p.C / kc_of(d->dtype));
  if (ksplit <= 0) {   // heuristic: enough blocks for ~2 per CU, at least 8 K-chunks per slice
    const long tiles = vg_ceil_div(p.M, p.Npad % 128 == 0 ? 128 : 256) * (p.Npad % 128 == 0 ? p.Npad / 128 : 1);
    ksplit = (int)vg_ceil_div(512, tiles);
  }
  if (ksplit > nk / 8) ksplit = nk / 8;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 65535) ksplit = 65535;
  p.ksplit = ksplit; p.splitws = (float*)ws; p.act = VG_ACT_NONE;
  if (ksplit == 1) {
    p.bias = bias; p.act = act;
    return d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, st) : dispatch_igemm<float>(p, st);
  }
  VG_HIP(hipMemsetAsync(ws, 0, vg_conv_splitk_ws_bytes(d), st));
  rc = d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, st) : dispatch_igemm<float>(p, st);
  if (rc) return rc;
  const long total = (long)p.M * p.Ks;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (d->dtype == VG_BF16) hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const float*)ws, bias, (bf16_t*)y, (long)p.M, p.Npad, p.Ks, p.Nout, act);
  else hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)ws, bias, (float*)y, (long)p.M, p.Npad, p.Ks, p.Nout, act);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_conv_dgrad(const vg_conv_desc* d, const void* dy, const void* w_packed_t, const float* bias, void* dx,
                             double* stats, int act, int accumulate, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(dy && w_packed_t && dx, "conv_dgrad: null pointer");
  const int s = d->stride;
  VG_REQUIRE(d->h % s == 0 && d->w % s == 0, "conv_dgrad: input %dx%d not a multiple of stride %d", d->h, d->w, s);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = dy; p.w = w_packed_t; p.bias = bias; p.y = dx; p.stats = stats;
  p.N = d->n; p.H = d->oh; p.W = d->ow; p.C = cpad_of(d->co, d->dtype); p.Cs = d->ycs;
  p.OHl = d->h / s; p.OWl = d->w / s; p.OH = d->h; p.OW = d->w; p.Ks = d->xcs;
  p.Nout = d->ci; p.Npad = vg_npad(d->ci);
  p.in_s = 1; p.out_s = s; p.nphases = s * s;
  p.act = act; p.accumulate = accumulate; p.ksplit = 1;
  p.M = d->n * p.OHl * p.OWl;
  int nt = -1;
  for (int ph = 0; ph < s * s; ++ph) {
    int n = vg_transposed_taps(d, ph, p.dy[ph], p.dx[ph], nullptr);
    VG_REQUIRE(n > 0 && (nt < 0 || n == nt), "conv_dgrad: uneven tap count across phases (k%d s%d p%d)", d->r, s, d->pad);
    nt = n;
  }
  p.ntaps = nt;
  return d->dtype == VG_BF16 ? dispatch_igemm<bf16_t>(p, (hipStream_t)stream) : dispatch_igemm<float>(p, (hipStream_t)stream);
}

// =============================================================================================
// weight gradient: dW[o][tap][c] = sum_m dy[m][o] * x[pix(m,tap)][c]   (K dimension = pixels)
// =============================================================================================
struct WgradParams {
  const void* x;
  const void* dy;
  float* dw;            // packed f32 [Opad][ntaps][Cpad], zeroed by the launcher
  int N, H, W, Cpad, Cs;
  int OH, OW, Opad, Os;
  int in_s, ntaps;
  int M, chunks_per_split, nchunks;
  int otiles, ctiles;
  signed char dy_[VG_MAX_TAPS];
  signed char dx_[VG_MAX_TAPS];
};

// 64-byte-segment swizzle for [pixel][channel] tiles read through ds_read_b64_tr_b16 (bf16 only)
template <typename T, int RS> __device__ __forceinline__ int wg_off(int row, int pc) {
  if (sizeof(T) == 4) return row * RS + pc * 16;
  constexpr int NSEG = RS / 64;
  int seg = pc >> 2;
  if (NSEG >= 4) seg ^= (row & 3);
  else if (NSEG == 2) seg ^= ((row >> 1) & 1);
  return row * RS + (seg << 6) + ((pc & 3) << 4);
}

template <typename T, int OI, int CI, int WO, int WC, int WK>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int BO = 32 * OI * WO, BC = 32 * CI * WC;
  constexpr int EB = sizeof(T), PIECE = 16 / EB;
  constexpr int KPW = 64 / EB;          // pixels per wave per chunk (32 bf16 / 16 f32)
  constexpr int KP = KPW * WK;          // pixels per chunk
  constexpr int RSO = BO * EB, RSC = BC * EB;   // tile row strides in bytes
  constexpr int TO = KP * RSO, TC = KP * RSC;
  constexpr int PPRO = BO / PIECE, PPRC = BC / PIECE;
  constexpr int NPO = (KP * PPRO + 255) / 256, NPC = (KP * PPRC + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sO = smem;             // [2][KP][BO]
  unsigned char* sC = smem + 2 * TO;    // [2][KP][BC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WK, wc = (wave / WK) % WC, wo = wave / (WK * WC);
  const int h = lane >> 5, l32 = lane & 31;

  int bid = blockIdx.x;
  const int tap = bid % p.ntaps; bid /= p.ntaps;
  const int ctile = bid % p.ctiles; const int otile = bid / p.ctiles;
  const int o0 = otile * BO, c0 = ctile * BC;
  const int chunk0 = blockIdx.y * p.chunks_per_split;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > p.nchunks) chunk1 = p.nchunks;
  const int tdy = p.dy_[tap], tdx = p.dx_[tap];

  const T* __restrict__ xg = (const T*)p.x;
  const T* __restrict__ dyg = (const T*)p.dy;

  // staging assignment: piece q = tid + 256*i  ->  (row = q / PPR, pc = q % PPR)
  // x rows need (n, oy, ox) of pixel m = chunk*KP + row, advanced incrementally by KP per chunk.
  int xr_n[NPC], xr_oy[NPC], xr_ox[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    int q = tid + 256 * i;
    int row = q / PPRC;
    long m = (long)chunk0 * KP + row;
    int ohw = p.OH * p.OW;
    int n = (int)(m / ohw);
    int rem = (int)(m - (long)n * ohw);
    xr_n[i] = n; xr_oy[i] = rem / p.OW; xr_ox[i] = rem - xr_oy[i] * p.OW;
  }
  u32x4 ro[NPO], rc[NPC];
  int lchunk = chunk0;   // next chunk to load
  auto load_chunk = [&]() {
    const long mbase = (long)lchunk * KP;
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRO, pc = q - row * PPRO;
      u32x4 v = {0u, 0u, 0u, 0u};
      long m = mbase + row;
      if (q < KP * PPRO && m < p.M && o0 + pc * PIECE < p.Os) v = *(const u32x4*)(dyg + (size_t)m * p.Os + o0 + pc * PIECE);
      ro[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRC, pc = q - row * PPRC;
      u32x4 v = {0u, 0u, 0u, 0u};
      int iy = xr_oy[i] * p.in_s + tdy, ix = xr_ox[i] * p.in_s + tdx;
      bool ok = q < KP * PPRC && xr_n[i] < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                c0 + pc * PIECE < p.Cpad;
      if (ok) v = *(const u32x4*)(xg + ((size_t)(xr_n[i] * p.H + iy) * p.W + ix) * p.Cs + c0 + pc * PIECE);
      rc[i] = v;
      // advance this row's pixel by KP for the next chunk
      int ox = xr_ox[i] + KP, oy = xr_oy[i], n = xr_n[i];
      while (ox >= p.OW) { ox -= p.OW; ++oy; }
      while (oy >= p.OH) { oy -= p.OH; ++n; }
      xr_ox[i] = ox; xr_oy[i] = oy; xr_n[i] = n;
    }
    ++lchunk;
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRO, pc = q - row * PPRO;
      if (q < KP * PPRO) *(u32x4*)(sO + buf * TO + wg_off<T, RSO>(row, pc)) = ro[i];
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      int q = tid + 256 * i;
      int row = q / PPRC, pc = q - row * PPRC;
      if (q < KP * PPRC) *(u32x4*)(sC + buf * TC + wg_off<T, RSC>(row, pc)) = rc[i];
    }
  };

  f32x16 acc[OI][CI];
#pragma unroll
  for (int i = 0; i < OI; ++i)
#pragma unroll
    for (int j = 0; j < CI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (chunk0 < chunk1) {
    load_chunk();
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = chunk0; ch < chunk1; ++ch) {
    const int buf = (ch - chunk0) & 1;
    if (ch + 1 < chunk1) load_chunk();
    const unsigned char* cO = sO + buf * TO;
    const unsigned char* cC = sC + buf * TC;
    if constexpr (sizeof(T) == 2) {
      // transposed LDS reads: lane (q = li>>2, pp = li&3) of each 16-lane group supplies the address of
      // row q, columns 4pp..4pp+3 of a 4x16 block; lane li receives column li of the 4 rows.
      const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4 fa[OI], fb[CI];
#pragma unroll
        for (int i = 0; i < OI; ++i) {
          bf16x4 lo, hi;
          const int col = (wo * OI + i) * 32 + 16 * g16 + 4 * tp;   // channel (element) index in tile
          const int r0 = wk * KPW + kk * 16 + 8 * h + tq;
          lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cO + wg_off<T, RSO>(r0, col >> 3) + (col & 7) * 2));
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cO + wg_off<T, RSO>(r0 + 4, col >> 3) + (col & 7) * 2));
          bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fa[i] = __builtin_bit_cast(u32x4, f);
        }
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          bf16x4 lo, hi;
          const int col = (wc * CI + i) * 32 + 16 * g16 + 4 * tp;
          const int r0 = wk * KPW + kk * 16 + 8 * h + tq;
          lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cC + wg_off<T, RSC>(r0, col >> 3) + (col & 7) * 2));
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(cC + wg_off<T, RSC>(r0 + 4, col >> 3) + (col & 7) * 2));
          bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fb[i] = __builtin_bit_cast(u32x4, f);
        }
#pragma unroll
        for (int i = 0; i < OI; ++i)
#pragma unroll
          for (int j = 0; j < CI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      // f32: A[i = channel l32][k = pixel 2*step + h] is a plain 4-byte LDS read
#pragma unroll
      for (int st = 0; st < KPW / 2; ++st) {
        const int row = wk * KPW + 2 * st + h;
        float fa[OI], fb[CI];
#pragma unroll
        for (int i = 0; i < OI; ++i) fa[i] = *(const float*)(cO + row * RSO + ((wo * OI + i) * 32 + l32) * 4);
#pragma unroll
        for (int i = 0; i < CI; ++i) fb[i] = *(const float*)(cC + row * RSC + ((wc * CI + i) * 32 + l32) * 4);
#pragma unroll
        for (int i = 0; i < OI; ++i)
#pragma unroll
          for (int j = 0; j < CI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (ch + 1 < chunk1) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // acc[i][j][r]: row (o) = (r&3) + 8*(r>>2) + 4*h, column (c) = l32
  const size_t rowstride = (size_t)p.ntaps * p.Cpad;
#pragma unroll
  for (int i = 0; i < OI; ++i)
#pragma unroll
    for (int j = 0; j < CI; ++j) {
      const int c = c0 + (wc * CI + j) * 32 + l32;
      if (c >= p.Cpad) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (wo * OI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (o < p.Opad) unsafeAtomicAdd(&p.dw[(size_t)o * rowstride + (size_t)tap * p.Cpad + c], acc[i][j][r]);
      }
    }
}

template <typename T, int OI, int CI, int WO, int WC, int WK>
static int launch_wgrad(const WgradParams& p0, hipStream_t st) {
  WgradParams p = p0;
  constexpr int BO = 32 * OI * WO, BC = 32 * CI * WC, EB = sizeof(T);
  constexpr int KP = (64 / EB) * WK;
  const size_t lds = 2 * (size_t)KP * (BO + BC) * EB;
  p.otiles = (int)vg_ceil_div(p.Opad, BO);
  p.ctiles = (int)vg_ceil_div(p.Cpad, BC);
  p.nchunks = (int)vg_ceil_div(p.M, KP);
  const int base_blocks = p.otiles * p.ctiles * p.ntaps;
  int splits = (int)vg_ceil_div(1024, base_blocks);        // aim for ~4 blocks per CU
  const int min_chunks = 8;
  if (splits > p.nchunks / min_chunks) splits = p.nchunks / min_chunks;
  if (splits < 1) splits = 1;
  p.chunks_per_split = (int)vg_ceil_div(p.nchunks, splits);
  splits = (int)vg_ceil_div(p.nchunks, p.chunks_per_split);
  auto kern = wgrad_kernel<T, OI, CI, WO, WC, WK>;
  static bool attr_set = false;
  if (!attr_set) {
    VG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(base_blocks, splits), dim3(256), lds, st, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

template <typename T> static int dispatch_wgrad(const WgradParams& p, hipStream_t st) {
  const bool o128 = p.Opad >= 128, c128 = p.Cpad >= 128;
  const bool o32 = p.Opad <= 32, c32 = p.Cpad <= 32;
  if (o128 && c128) return launch_wgrad<T, 2, 2, 2, 2, 1>(p, st);      // 128 x 128
  if (o128 && c32) return launch_wgrad<T, 2, 1, 2, 1, 2>(p, st);       // 128 x 32
  if (o128) return launch_wgrad<T, 2, 2, 2, 1, 2>(p, st);              // 128 x 64
  if (o32 && c128) return launch_wgrad<T, 1, 2, 1, 2, 2>(p, st);       // 32 x 128
  if (o32) return launch_wgrad<T, 1, 2, 1, 1, 4>(p, st);               // 32 x 64 (covers c<=64)
  if (c128) return launch_wgrad<T, 2, 2, 1, 2, 2>(p, st);              // 64 x 128
  if (c32) return launch_wgrad<T, 2, 1, 1, 1, 4>(p, st);               // 64 x 32
  return launch_wgrad<T, 2, 2, 1, 1, 4>(p, st);                        // 64 x 64
}

__global__ void wgrad_unpack_kernel(const float* __restrict__ packed, float* __restrict__ dw, int O, int I, int RS,
                                    int Cpad) {
  // dw[o][i][rs] = packed[o][rs][i]
  const long total = (long)O * I * RS;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int rs = (int)(t % RS);
    long oi = t / RS;
    int i = (int)(oi % I);
    int o = (int)(oi / I);
    dw[t] = packed[((size_t)o * RS + rs) * Cpad + i];
  }
}

extern "C" size_t vg_conv_wgrad_ws_bytes(const vg_conv_desc* d) {
  if (!d) return 0;
  int opad = (int)vg_ceil_div(d->co, 32) * 32;
  return (size_t)opad * d->r * d->s * cpad_of(d->ci, d->dtype) * sizeof(float);
}

extern "C" int vg_conv_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws,
                             void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VG_REQUIRE(x && dy && dw_oihw && ws, "conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.dy = dy; p.dw = (float*)ws;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cpad = cpad_of(d->ci, d->dtype); p.Cs = d->xcs;
  p.OH = d->oh; p.OW = d->ow; p.Opad = (int)vg_ceil_div(d->co, 32) * 32; p.Os = d->ycs;
  if (p.Opad > p.Os) p.Opad = p.Os;   // f32 strides are multiples of 16: never read past the row
  p.in_s = d->stride; p.ntaps = d->r * d->s;
  p.M = d->n * d->oh * d->ow;
  for (int r = 0; r < d->r; ++r)
    for (int s = 0; s < d->s; ++s) { p.dy_[r * d->s + s] = (signed char)(r - d->pad); p.dx_[r * d->s + s] = (signed char)(s - d->pad); }
  VG_HIP(hipMemsetAsync(ws, 0, vg_conv_wgrad_ws_bytes(d), st));
  rc = d->dtype == VG_BF16 ? dispatch_wgrad<bf16_t>(p, st) : dispatch_wgrad<float>(p, st);
  if (rc) return rc;
  const long total = (long)d->co * d->ci * d->r * d->s;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw_oihw, d->co, d->ci,
                     d->r * d->s, p.Cpad);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
