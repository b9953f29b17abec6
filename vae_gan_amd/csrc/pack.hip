// Weight packing, NCHW<->NHWC layout conversion and per-channel column sums.
#include <stdarg.h>

#include "common.h"
#include <map>
#include <mutex>
#include <utility>
#include <mutex>

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void vg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vg_last_error(void) { return g_err; }
extern "C" int vg_version(void) { return 100; }

int vg_npad(int n);
int vg_transposed_taps(const vg_conv_desc* d, int phase, signed char* dy, signed char* dx, int* rs);

static int kc_of(int dtype) { return dtype == VG_BF16 ? 32 : 16; }
static int cpad_of(int c, int dtype) { int kc = kc_of(dtype); return (int)vg_ceil_div(c, kc) * kc; }

struct PackParams {
  const float* w;
  void* out;
  int O, I, RS;          // master weight dims [O][I][RS]
  int Npad, Cpad, ntaps, nphases, transposed;
  int rs[4][16];
};

template <typename T> __global__ void pack_kernel(const PackParams p) {
  const long total = (long)p.nphases * p.Npad * p.ntaps * p.Cpad;
  T* out = (T*)p.out;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int c = (int)(t % p.Cpad);
    long r1 = t / p.Cpad;
    int tap = (int)(r1 % p.ntaps);
    long r2 = r1 / p.ntaps;
    int n = (int)(r2 % p.Npad);
    int ph = (int)(r2 / p.Npad);
    int rs = p.rs[ph][tap];
    float v = 0.f;
    if (!p.transposed) {
      if (n < p.O && c < p.I) v = p.w[((size_t)n * p.I + c) * p.RS + rs];
    } else {
      if (n < p.I && c < p.O) v = p.w[((size_t)c * p.I + n) * p.RS + rs];
    }
    out[t] = Elt<T>::from_f(v);
  }
}

static int fill_pack(const vg_conv_desc* d, int transposed, PackParams* p) {
  memset(p, 0, sizeof(*p));
  p->O = d->co; p->I = d->ci; p->RS = d->r * d->s; p->transposed = transposed;
  if (!transposed) {
    p->Npad = vg_npad(d->co); p->Cpad = cpad_of(d->ci, d->dtype); p->ntaps = d->r * d->s; p->nphases = 1;
    for (int i = 0; i < p->ntaps; ++i) p->rs[0][i] = i;
  } else {
    p->Npad = vg_npad(d->ci); p->Cpad = cpad_of(d->co, d->dtype); p->nphases = d->stride * d->stride;
    int nt = -1;
    for (int ph = 0; ph < p->nphases; ++ph) {
      signed char dy[16], dx[16];
      int n = vg_transposed_taps(d, ph, dy, dx, p->rs[ph]);
      if (n <= 0 || (nt >= 0 && n != nt)) return -1;
      nt = n;
    }
    p->ntaps = nt;
  }
  return 0;
}

extern "C" size_t vg_conv_packed_bytes(const vg_conv_desc* d, int transposed) {
  PackParams p;
  if (!d || fill_pack(d, transposed, &p)) return 0;
  return (size_t)p.nphases * p.Npad * p.ntaps * p.Cpad * vg_elt_size(d->dtype);
}

extern "C" int vg_conv_pack_weight(const vg_conv_desc* d, int transposed, const float* w_oihw, void* packed,
                                   void* stream) {
  VG_REQUIRE(d && w_oihw && packed, "pack_weight: null pointer");
  VG_REQUIRE(d->r * d->s <= 16 && d->stride >= 1 && d->stride <= 2, "pack_weight: unsupported geometry");
  PackParams p;
  VG_REQUIRE(fill_pack(d, transposed, &p) == 0, "pack_weight: uneven tap count (k%d s%d p%d)", d->r, d->stride, d->pad);
  p.w = w_oihw; p.out = packed;
  const long total = (long)p.nphases * p.Npad * p.ntaps * p.Cpad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (d->dtype == VG_BF16) hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- batched re-pack: every packed weight image of one optimizer in ONE launch after its step (a training step
// re-packs ~60 images; as separate 7 us launches that was 0.43 ms of a 44 ms step).  The caller keeps a table of
// PackParams in device memory (vg_conv_pack_item fills one on the host); block (x, y) works on item y.
// A block takes 32 x 32 (n, c) entries of the packed image with all their taps: it reads the matching 32 rows of
// 32 * RS consecutive master weights (coalesced; the packed order would read every RS-th float), turns them in LDS and
// writes, per phase and tap, 32 rows of 32 consecutive packed elements.  Padding rows / columns are written as zeros.
template <typename T> __global__ __launch_bounds__(256) void pack_batch_kernel(const PackParams* __restrict__ items) {
  extern __shared__ unsigned char tile_raw[];          // T [32][32 * bs + 2], bs = RS | 1; converted on the way in (half the LDS
  T* tile = (T*)tile_raw;                               // for bf16: four blocks per CU instead of two)
  const PackParams& p = items[blockIdx.y];
  const int RS = p.RS, bs = RS | 1, rowf = 32 * bs + 2;
  const int A = p.O, B = p.I;                           // master weights [A][B][RS]
  const int ctiles = (p.Cpad + 31) / 32, ntiles = ((p.Npad + 31) / 32) * ctiles;
  T* out = (T*)p.out;
  const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  __shared__ unsigned short lut[32 * 16];
  for (int rem = threadIdx.x; rem < 32 * RS; rem += 256) lut[rem] = (unsigned short)((rem / RS) << 4 | (rem % RS));
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n0 = (t / ctiles) * 32, c0 = (t % ctiles) * 32;
    const int a0 = p.transposed ? c0 : n0, b0 = p.transposed ? n0 : c0;
    __syncthreads();
    for (int a = 0; a < 32; ++a) {                       // a row of 32 * RS consecutive master weights at a time
      const bool arow = a0 + a < A;
      const float* src = p.w + ((size_t)(a0 + a) * B + b0) * RS;
      for (int rem = threadIdx.x; rem < 32 * RS; rem += 256) {
        const int b = lut[rem] >> 4, rs = lut[rem] & 15;  // rem = b * RS + rs (no integer division per element)
        float v = 0.f;
        if (arow && b0 + b < B) v = src[rem];
        tile[a * rowf + b * bs + rs] = Elt<T>::from_f(v);
      }
    }
    __syncthreads();
    for (int ph = 0; ph < p.nphases; ++ph)
      for (int tap = 0; tap < p.ntaps; ++tap) {
        const int rs = p.rs[ph][tap];
#pragma unroll
        for (int r = grp; r < 32; r += 8) {             // r: row of the packed tile (n), lane32: its column (c)
          const int n = n0 + r, c = c0 + lane32;
          if (n >= p.Npad || c >= p.Cpad) continue;
          out[(((size_t)ph * p.Npad + n) * p.ntaps + tap) * p.Cpad + c] =
              p.transposed ? tile[lane32 * rowf + r * bs + rs] : tile[r * rowf + lane32 * bs + rs];
        }
      }
  }
}

extern "C" size_t vg_conv_pack_item_bytes(void) { return sizeof(PackParams); }

extern "C" int vg_conv_pack_item(const vg_conv_desc* d, int transposed, const float* w_oihw, void* packed, void* item) {
  VG_REQUIRE(d && w_oihw && packed && item, "pack_item: null pointer");
  VG_REQUIRE(d->r * d->s <= 16 && d->stride >= 1 && d->stride <= 2, "pack_item: unsupported geometry");
  PackParams p;
  VG_REQUIRE(fill_pack(d, transposed, &p) == 0, "pack_item: uneven tap count (k%d s%d p%d)", d->r, d->stride, d->pad);
  p.w = w_oihw; p.out = packed;
  memcpy(item, &p, sizeof(p));
  return VG_OK;
}

extern "C" int vg_conv_pack_batch(const void* items_dev, int n_items, int dtype, void* stream) {
  VG_REQUIRE(items_dev && n_items > 0 && n_items <= 65535, "pack_batch: bad arguments");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "pack_batch: bad dtype %d", dtype);
  dim3 grid(256, n_items);                               // a block per 32 x 32 tile: 512 x 512 channels = 256 tiles
  const size_t lds = (size_t)32 * (32 * 17 + 2) * (dtype == VG_BF16 ? 2 : 4);      // RS <= 16
    VG_HIP(vg_lds_attr((const void*)pack_batch_kernel<bf16_t>, (int)lds));
    VG_HIP(vg_lds_attr((const void*)pack_batch_kernel<float>, (int)lds));
  if (dtype == VG_BF16) hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const PackParams*)items_dev);
  else hipLaunchKernelGGL(pack_batch_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const PackParams*)items_dev);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW f32 <-> NHWC T.  One block handles 64 pixels x all channels of one image through LDS so that
// both sides are coalesced (pixels contiguous in NCHW, channels contiguous in NHWC).
// ---------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                                                  int C, int HW, int Cs) {
  __shared__ float tile[64][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int cc = ty; cc < 32; cc += 4) {
    int c = c0 + cc, pix = p0 + tx;
    float v = 0.f;
    if (c < C && pix < HW) v = src[((size_t)n * C + c) * HW + pix];
    tile[tx][cc] = v;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;   // 32 x 8
  for (int pp = py; pp < 64; pp += 8) {
    int pix = p0 + pp, c = c0 + cx;
    if (pix < HW && c < Cs) dst[((size_t)n * HW + pix) * Cs + c] = Elt<T>::from_f(tile[pp][cx]);
  }
}

template <typename T> __global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                                                  int C, int HW, int Cs) {
  __shared__ float tile[64][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  for (int pp = py; pp < 64; pp += 8) {
    int pix = p0 + pp, c = c0 + cx;
    float v = 0.f;
    if (pix < HW && c < Cs) v = Elt<T>::to_f(src[((size_t)n * HW + pix) * Cs + c]);
    tile[pp][cx] = v;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 32; cc += 4) {
    int c = c0 + cc, pix = p0 + tx;
    if (c < C && pix < HW) dst[((size_t)n * C + c) * HW + pix] = tile[tx][cc];
  }
}

// 4-channel bf16 NHWC (the thin image-side layout, csrc/thin.hip): a thread owns one pixel, plane reads and 8-byte
// pixel writes are both coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int C, long HW) {
  const long pix = blockIdx.x * 256L + threadIdx.x;
  if (pix >= HW) return;
  const int n = blockIdx.y;
  bf16x4 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = (bf16_t)(c < C ? src[((size_t)n * C + c) * HW + pix] : 0.f);
  *(bf16x4*)(dst + ((size_t)n * HW + pix) * 4) = o;
}
__global__ __launch_bounds__(256) void nhwc4_to_nchw_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int C, long HW) {
  const long pix = blockIdx.x * 256L + threadIdx.x;
  if (pix >= HW) return;
  const int n = blockIdx.y;
  const bf16x4 v = *(const bf16x4*)(src + ((size_t)n * HW + pix) * 4);
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < C) dst[((size_t)n * C + c) * HW + pix] = (float)v[c];
}

extern "C" int vg_nchw_to_nhwc(const float* src, void* dst, int n, int c, int h, int w, int cs, int dtype, void* stream) {
  VG_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && cs >= c, "nchw_to_nhwc: bad arguments");
  if (dtype == VG_BF16 && cs == 4) {
    const long hw = (long)h * w;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3((unsigned)vg_ceil_div(hw, 256), n), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, c, hw);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  dim3 grid((unsigned)vg_ceil_div((int64_t)h * w, 64), (unsigned)vg_ceil_div(cs, 32), n);
  if (dtype == VG_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, c, h * w, cs);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, c, h * w, cs);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_nhwc_to_nchw(const void* src, float* dst, int n, int c, int h, int w, int cs, int dtype, void* stream) {
  VG_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && cs >= c, "nhwc_to_nchw: bad arguments");
  if (dtype == VG_BF16 && cs == 4) {
    const long hw = (long)h * w;
    hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3((unsigned)vg_ceil_div(hw, 256), n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, c, hw);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  dim3 grid((unsigned)vg_ceil_div((int64_t)h * w, 64), (unsigned)vg_ceil_div(c, 32), n);
  if (dtype == VG_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, c, h * w, cs);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, c, h * w, cs);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// Library-wide reproducibility switch (vg_api.h: vg_set_deterministic).  Read by every launcher that accumulates
// across blocks: statistics go through the fixed-point accumulators of common.h, kernels that split a reduction
// over blocks with f32 atomics run unsplit or on an ordered variant.
// ---------------------------------------------------------------------------------------------
hipError_t vg_lds_attr(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;      // (kernel, device) -> bytes granted
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({kernel, dev});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[{kernel, dev}] = bytes;
  return e;
}

static int g_det_mode = 0;
int vg_det_mode(void) { return g_det_mode; }
extern "C" void vg_set_deterministic(int on) { g_det_mode = on ? 1 : 0; }
extern "C" int vg_get_deterministic(void) { return g_det_mode; }

// ---------------------------------------------------------------------------------------------
// per-channel column sum (bias gradients): out[c] = sum_m a[m][c]
// thread = one 16-byte piece column over a strided set of rows; the block folds its row groups
// through LDS and issues ONE f64 atomic per channel into one of VG_STATS_REPLICAS copies.
// ---------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ a, int64_t rows, int cs,
                                                                                double* __restrict__ ws, int det) {
  constexpr int PIECE = Elt<T>::PIECE;
  __shared__ float red[256 * PIECE];
  const int ppr = cs / PIECE;
  const int rpb = 256 / ppr;
  const int pc = threadIdx.x % ppr, rr = threadIdx.x / ppr;
  float acc[PIECE];
  double dacc[PIECE];
#pragma unroll
  for (int e = 0; e < PIECE; ++e) { acc[e] = 0.f; dacc[e] = 0.0; }
  if (rr < rpb) {
    int cnt = 0;
    for (int64_t r = (int64_t)blockIdx.x * rpb + rr; r < rows; r += (int64_t)gridDim.x * rpb) {
      Piece<T> v;
      v.v = *(const decltype(v.v)*)(a + (size_t)r * cs + pc * PIECE);
#pragma unroll
      for (int e = 0; e < PIECE; ++e) acc[e] += v.get(e);
      if (++cnt == 64) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) { dacc[e] += acc[e]; acc[e] = 0.f; }
        cnt = 0;
      }
    }
#pragma unroll
    for (int e = 0; e < PIECE; ++e) red[rr * cs + pc * PIECE + e] = (float)(dacc[e] + (double)acc[e]);
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < cs; ch += 256) {
    double t = 0.0;
    for (int r = 0; r < rpb; ++r) t += red[r * cs + ch];
    vg_acc_add(ws + ch, cs, blockIdx.x, t, det);
  }
}

// det (only with replicas == VG_STATS_REPLICAS): the slots are the fixed-point bins of vg_acc_add
__global__ void d2f_kernel(const double* __restrict__ src, float* __restrict__ dst, int n, double scale, int replicas,
                           int stride, int accumulate, int det) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double t = 0.0;
  if (det) t = vg_acc_read(src + i, stride, 1);
  else
    for (int r = 0; r < replicas; ++r) t += src[(size_t)r * stride + i];
  dst[i] = accumulate ? dst[i] + (float)(t * scale) : (float)(t * scale);
}

// 4-channel bf16 rows (the thin image-side layout): a thread walks pixels, 8-byte loads
__global__ __launch_bounds__(256) void channel_sum4_kernel(const bf16_t* __restrict__ a, int64_t rows, double* __restrict__ ws, int det) {
  __shared__ float red[4][4];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  double dacc[4] = {0.0, 0.0, 0.0, 0.0};
  int cnt = 0;
  for (int64_t r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const bf16x4 v = *(const bf16x4*)(a + r * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += (float)v[e];
    if (++cnt == 64) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { dacc[e] += acc[e]; acc[e] = 0.f; }
      cnt = 0;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t = wave_sum((float)(dacc[e] + (double)acc[e]));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int e = threadIdx.x;
    vg_acc_add(ws + e, 4, blockIdx.x, (double)((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])), det);
  }
}

extern "C" int vg_channel_sum(const void* a, int64_t rows, int cs, int c_valid, int dtype, float* out, double* ws,
                              int accumulate, void* stream) {
  VG_REQUIRE(a && out && ws && rows > 0 && cs > 0 && c_valid <= cs, "channel_sum: bad arguments");
  if (dtype == VG_BF16 && cs == 4) {
    hipStream_t st4 = (hipStream_t)stream;
    VG_HIP(hipMemsetAsync(ws, 0, sizeof(double) * 4 * VG_STATS_REPLICAS, st4));
    int blocks = (int)vg_ceil_div(rows, 256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(channel_sum4_kernel, dim3(blocks), dim3(256), 0, st4, (const bf16_t*)a, rows, ws, vg_det_mode());
    VG_LAUNCH_CHECK();
    hipLaunchKernelGGL(d2f_kernel, dim3(1), dim3(64), 0, st4, (const double*)ws, out, c_valid, 1.0, VG_STATS_REPLICAS, 4, accumulate, vg_det_mode());
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  const int piece = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(cs % piece == 0 && cs / piece <= 256, "channel_sum: channel stride %d unsupported", cs);
  hipStream_t st = (hipStream_t)stream;
  VG_HIP(hipMemsetAsync(ws, 0, sizeof(double) * cs * VG_STATS_REPLICAS, st));
  const int ppr = cs / piece, rpb = 256 / ppr;
  int blocks = (int)vg_ceil_div(rows, (int64_t)rpb * 16);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == VG_BF16) hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)a, rows, cs, ws, vg_det_mode());
  else hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)a, rows, cs, ws, vg_det_mode());
  VG_LAUNCH_CHECK();
  hipLaunchKernelGGL(d2f_kernel, dim3((c_valid + 255) / 256), dim3(256), 0, st, (const double*)ws, out, c_valid, 1.0,
                     VG_STATS_REPLICAS, cs, accumulate, vg_det_mode());
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// vg_channel_sum without the caller's workspace and without any allocation: for callers that cannot allocate (code
// under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The f64 replica scratch the kernel
// accumulates into lives in a file-scope device array.  On the host the array's NAME is only a stand-in, not a
// device address: its address is taken ONCE per device with hipGetSymbolAddress (that device current) and only that
// pointer is ever handed to a kernel or a memset.  Every GPU holds its own copy at its own address until the process
// ends.  All calls on one GPU share the array (it keeps the last call's contents: each call clears the part it uses
// first, on its stream), so calls that overlap on one GPU race.
// ---------------------------------------------------------------------------------------------
#define VG_NOALLOC_MAX_CS 1024       // 32 replicas x 1024 channels x 8 B = 256 KiB per GPU
#define VG_MAX_DEVICES 64
static __device__ __align__(256) double vg_scratch[VG_STATS_REPLICAS * VG_NOALLOC_MAX_CS];
static double* g_scratch_ptr[VG_MAX_DEVICES];
static std::mutex g_scratch_mutex;

static int scratch_of_current_device(double** out) {
  int dev = -1;
  VG_HIP(hipGetDevice(&dev));
  VG_REQUIRE(dev >= 0 && dev < VG_MAX_DEVICES, "channel_sum_noalloc: device ordinal %d outside [0, %d)", dev, VG_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    VG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(vg_scratch)));
    VG_REQUIRE(p != nullptr, "channel_sum_noalloc: hipGetSymbolAddress returned a null address");
    g_scratch_ptr[dev] = (double*)p;
  }
  *out = g_scratch_ptr[dev];
  return VG_OK;
}

extern "C" int vg_channel_sum_noalloc(const void* a, int64_t rows, int cs, int c_valid, int dtype, float* out,
                                      int accumulate, void* stream) {
  VG_REQUIRE(a && out && rows > 0 && cs > 0 && c_valid > 0 && c_valid <= cs, "channel_sum_noalloc: bad arguments");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "channel_sum_noalloc: bad dtype %d", dtype);
  VG_REQUIRE(cs <= VG_NOALLOC_MAX_CS, "channel_sum_noalloc: channel stride %d exceeds the static scratch (max %d); use "
             "vg_channel_sum with a caller workspace", cs, VG_NOALLOC_MAX_CS);
  const int piece = dtype == VG_BF16 ? 8 : 4;
  const bool thin4 = dtype == VG_BF16 && cs == 4;        // the 8-byte-pixel image-side layout, as in vg_channel_sum
  VG_REQUIRE(thin4 || (cs % piece == 0 && cs / piece <= 256), "channel_sum_noalloc: channel stride %d unsupported", cs);
  double* ws = nullptr;
  int rc = scratch_of_current_device(&ws);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  VG_HIP(hipMemsetAsync(ws, 0, sizeof(double) * cs * VG_STATS_REPLICAS, st));      // the replicas must start at zero
  if (thin4) {
    int blocks4 = (int)vg_ceil_div(rows, 256 * 16);
    if (blocks4 > 2048) blocks4 = 2048;
    hipLaunchKernelGGL(channel_sum4_kernel, dim3(blocks4), dim3(256), 0, st, (const bf16_t*)a, rows, ws, vg_det_mode());
    VG_LAUNCH_CHECK();
    hipLaunchKernelGGL(d2f_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, out, c_valid, 1.0, VG_STATS_REPLICAS, 4, accumulate, vg_det_mode());
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  const int ppr = cs / piece, rpb = 256 / ppr;
  int blocks = (int)vg_ceil_div(rows, (int64_t)rpb * 16);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == VG_BF16) hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)a, rows, cs, ws, vg_det_mode());
  else hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)a, rows, cs, ws, vg_det_mode());
  VG_LAUNCH_CHECK();
  hipLaunchKernelGGL(d2f_kernel, dim3((c_valid + 255) / 256), dim3(256), 0, st, (const double*)ws, out, c_valid, 1.0,
                     VG_STATS_REPLICAS, cs, accumulate, vg_det_mode());
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_scalar_finish(const double* src, double scale, float* dst, void* stream) {
  VG_REQUIRE(src && dst, "scalar_finish: null pointer");
  hipLaunchKernelGGL(d2f_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, dst, 1, scale, 1, 0, 0, 0);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// 2x2 average pooling of an NCHW f32 tensor (the non-reference 'area' target of shim D1: the beta decoder
// emits 128x128 for a 256x256 input, beta_vae_gans/model.py:104,121-126)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t planes,
                                                       int h, int w) {
  const int oh = h / 2, ow = w / 2;
  const int64_t total = planes * oh * ow;
  for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int ox = (int)(t % ow);
    const int64_t r = t / ow;
    const int oy = (int)(r % oh);
    const int64_t pl = r / oh;
    const float* s = src + (pl * h + 2 * oy) * w + 2 * ox;
    dst[t] = 0.25f * ((s[0] + s[1]) + (s[w] + s[w + 1]));
  }
}

extern "C" int vg_avgpool2_nchw(const float* src, float* dst, int64_t planes, int h, int w, void* stream) {
  VG_REQUIRE(src && dst && planes > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "avgpool2: bad arguments");
  int64_t total = planes * (h / 2) * (w / 2);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, planes, h, w);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
