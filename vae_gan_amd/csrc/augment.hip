// Training / eval input transform on the device (beta_vae_gans/data.py:106-119):
//   RandomHorizontalFlip -> RandomVerticalFlip -> RandomResizedCrop(S, scale=(0.8,1)) -> ToTensor   (train)
//   Resize((S,S)) -> ToTensor                                                                       (eval)
// for a batch of decoded uint8 HWC images of different sizes, with the random draws made by the caller.  The pixel
// arithmetic is Pillow's Image.resize(BILINEAR) (ImagingResample, 8 bits per channel), reproduced bit for bit:
// double-precision antialiased triangle coefficients, normalised and rounded to 22-bit fixed point; a horizontal
// pass rounded to uint8, then a vertical pass; ToTensor = uint8 / 255 in float32, CHW.  Flips and the crop are
// folded into the addressing of the horizontal pass.  Byte work, HBM-bound: each source row is staged once in LDS
// with aligned dword loads, the horizontal result (rows x S x 3 bytes) goes through a caller-owned workspace.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int ITEM = 12;          // int64 words per image in the device table
constexpr int PREC = 32 - 8 - 2;  // Pillow's PRECISION_BITS

struct Axis {
  double scale, support, ss;
  int ksize;
};
__host__ __device__ inline Axis make_axis(int in_size, int out_size) {
  Axis a;
  a.scale = (double)in_size / out_size;
  const double fs = a.scale < 1.0 ? 1.0 : a.scale;
  a.support = fs;                          // bilinear support 1.0 x filterscale
  a.ss = 1.0 / fs;
  a.ksize = (int)ceil(a.support) * 2 + 1;
  return a;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output position xx.  No FMA contraction (pragma at the
// top of the file): Pillow is built for baseline x86-64, every product and sum rounds separately.
__device__ inline void axis_bounds(const Axis& a, int in_size, int xx, int& xmin, int& n) {
  const double center = 0.0 + (xx + 0.5) * a.scale;
  xmin = (int)(center - a.support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + a.support + 0.5);
  if (xmax > in_size) xmax = in_size;
  n = xmax - xmin;
}
__device__ inline void axis_coeffs(const Axis& a, int in_size, int xx, int* k, int& xmin, int& n) {
  axis_bounds(a, in_size, xx, xmin, n);
  const double center = 0.0 + (xx + 0.5) * a.scale;
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    double t = (x + xmin - center + 0.5) * a.ss;
    if (t < 0.0) t = -t;
    ww += t < 1.0 ? 1.0 - t : 0.0;
  }
  for (int x = 0; x < n; ++x) {
    double t = (x + xmin - center + 0.5) * a.ss;
    if (t < 0.0) t = -t;
    double w = t < 1.0 ? 1.0 - t : 0.0;
    if (ww != 0.0) w /= ww;
    const double v = w * (double)(1 << PREC);
    k[x] = w < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
  }
}

__device__ inline int clip8(int v) {
  v >>= PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct Item {
  const uint8_t* src;
  int H, W, fh, fv, ci, cj, ch, cw;
  int64_t tmp_off;
};
__device__ inline Item load_item(const int64_t* items, int i) {
  const int64_t* p = items + (int64_t)i * ITEM;
  Item t;
  t.src = (const uint8_t*)p[0];
  t.H = (int)p[1]; t.W = (int)p[2]; t.fh = (int)p[3]; t.fv = (int)p[4];
  t.ci = (int)p[5]; t.cj = (int)p[6]; t.ch = (int)p[7]; t.cw = (int)p[8];
  t.tmp_off = p[9];
  return t;
}

// grid (images, row chunks); LDS: kh[OW*kmax] ints, bnd[2*OW] ints, row buffer.
__global__ __launch_bounds__(256) void aug_h_kernel(const int64_t* __restrict__ items, int OH, int S, uint8_t* __restrict__ ws,
                                                    int rows_per_block, int kmax, int row_dwords) {
  extern __shared__ int lds[];
  int* kh = lds;
  int* bnd = lds + S * kmax;
  uint32_t* rowbuf = (uint32_t*)(bnd + 2 * S);
  const int tid = threadIdx.x;
  const Item it = load_item(items, blockIdx.x);
  const Axis av = make_axis(it.ch, OH), ah = make_axis(it.cw, S);      // S = output width
  int first, n0, lmin, ln;
  axis_bounds(av, it.ch, 0, first, n0);
  axis_bounds(av, it.ch, OH - 1, lmin, ln);
  const int rows = lmin + ln - first;                      // crop rows the vertical pass will read
  const int r0 = blockIdx.y * rows_per_block;
  if (r0 >= rows) return;
  for (int xx = tid; xx < S; xx += 256) {
    int xmin, n;
    axis_coeffs(ah, it.cw, xx, kh + xx * kmax, xmin, n);
    bnd[2 * xx] = xmin;
    bnd[2 * xx + 1] = n;
  }
  uint8_t* tmp = ws + it.tmp_off;
  const int start = it.fh ? it.W - it.cj - it.cw : it.cj;  // first original column of the crop
  const int r1 = min(r0 + rows_per_block, rows);
  for (int r = r0; r < r1; ++r) {
    const int fy = it.ci + first + r;
    const int oy = it.fv ? it.H - 1 - fy : fy;
    const uint8_t* a = it.src + ((int64_t)oy * it.W + start) * 3;
    const int shift = (int)((uintptr_t)a & 3);
    const uint32_t* a0 = (const uint32_t*)(a - shift);     // aligned dwords: a dword holding one valid byte cannot fault
    const int ndw = (shift + it.cw * 3 + 3) >> 2;
    __syncthreads();
    for (int d = tid; d < ndw; d += 256) rowbuf[d] = a0[d];
    __syncthreads();
    const uint8_t* px = (const uint8_t*)rowbuf + shift;
    for (int xx = tid; xx < S; xx += 256) {
      const int x0 = bnd[2 * xx], n = bnd[2 * xx + 1];
      const int* k = kh + xx * kmax;
      int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
      for (int x = 0; x < n; ++x) {
        const int p = it.fh ? it.cw - 1 - (x0 + x) : x0 + x;
        const int kv = k[x];
        s0 += (int)px[p * 3] * kv;
        s1 += (int)px[p * 3 + 1] * kv;
        s2 += (int)px[p * 3 + 2] * kv;
      }
      uint8_t* o = tmp + ((int64_t)r * S + xx) * 3;
      o[0] = (uint8_t)clip8(s0);
      o[1] = (uint8_t)clip8(s1);
      o[2] = (uint8_t)clip8(s2);
    }
  }
}

// grid (images, output row chunks); LDS: kv[rows_per_block*kmax], bnd[2*rows_per_block].
__global__ __launch_bounds__(256) void aug_v_kernel(const int64_t* __restrict__ items, int OH, int S, const uint8_t* __restrict__ ws,
                                                    float* __restrict__ out, int rows_per_block, int kmax) {
  extern __shared__ int lds[];
  int* kv = lds;
  int* bnd = lds + rows_per_block * kmax;
  const int tid = threadIdx.x;
  const Item it = load_item(items, blockIdx.x);
  const Axis av = make_axis(it.ch, OH);
  int first, n0;
  axis_bounds(av, it.ch, 0, first, n0);
  const int y0 = blockIdx.y * rows_per_block;
  for (int t = tid; t < rows_per_block && y0 + t < OH; t += 256) {
    int ymin, n;
    axis_coeffs(av, it.ch, y0 + t, kv + t * kmax, ymin, n);
    bnd[2 * t] = ymin - first;
    bnd[2 * t + 1] = n;
  }
  __syncthreads();
  const uint8_t* tmp = ws + it.tmp_off;
  float* o = out + (int64_t)blockIdx.x * 3 * OH * S;
  const int row_bytes = S * 3;
  for (int t = 0; t < rows_per_block && y0 + t < OH; ++t) {
    const int ymin = bnd[2 * t], n = bnd[2 * t + 1];
    const int* k = kv + t * kmax;
    for (int f = tid; f < row_bytes; f += 256) {
      int s = 1 << (PREC - 1);
      for (int y = 0; y < n; ++y) s += (int)tmp[(int64_t)(ymin + y) * row_bytes + f] * k[y];
      const int xx = f / 3, c = f - xx * 3;
      o[((int64_t)c * OH + (y0 + t)) * S + xx] = (float)clip8(s) / 255.0f;      // ToTensor
    }
  }
}

struct HostPlan {
  int kmax, max_rows, max_cw;
  size_t table_bytes, total_bytes;
};

int plan(const vg_aug_item* items, int n, int OH, int S, HostPlan& p, int64_t* table) {
  VG_REQUIRE(items && n > 0, "augment: empty batch");
  VG_REQUIRE(S >= 1 && S <= 8192 && OH >= 1 && OH <= 8192, "augment: output size %dx%d unsupported (1..8192)", OH, S);
  p.kmax = 3;
  p.max_rows = 1;
  p.max_cw = 1;
  p.table_bytes = ((size_t)n * ITEM * sizeof(int64_t) + 255) & ~(size_t)255;
  size_t off = p.table_bytes;
  for (int i = 0; i < n; ++i) {
    const vg_aug_item& t = items[i];
    VG_REQUIRE(t.src && t.height > 0 && t.width > 0, "augment: image %d is empty", i);
    VG_REQUIRE(t.crop_h > 0 && t.crop_w > 0 && t.crop_i >= 0 && t.crop_j >= 0 && t.crop_i + t.crop_h <= t.height &&
                   t.crop_j + t.crop_w <= t.width,
               "augment: image %d: crop (i=%d, j=%d, h=%d, w=%d) outside the %dx%d image", i, t.crop_i, t.crop_j, t.crop_h,
               t.crop_w, t.height, t.width);
    VG_REQUIRE(t.crop_h <= 16384 && t.crop_w <= 16384, "augment: image %d: crop larger than 16384", i);
    const Axis av = make_axis(t.crop_h, OH), ah = make_axis(t.crop_w, S);
    p.kmax = p.kmax > av.ksize ? p.kmax : av.ksize;
    p.kmax = p.kmax > ah.ksize ? p.kmax : ah.ksize;
    p.max_rows = p.max_rows > t.crop_h ? p.max_rows : t.crop_h;
    p.max_cw = p.max_cw > t.crop_w ? p.max_cw : t.crop_w;
    if (table) {
      int64_t* w = table + (size_t)i * ITEM;
      w[0] = (int64_t)(uintptr_t)t.src; w[1] = t.height; w[2] = t.width; w[3] = t.flip_h != 0; w[4] = t.flip_v != 0;
      w[5] = t.crop_i; w[6] = t.crop_j; w[7] = t.crop_h; w[8] = t.crop_w; w[9] = (int64_t)off; w[10] = w[11] = 0;
    }
    off += ((size_t)t.crop_h * S * 3 + 255) & ~(size_t)255;
  }
  p.total_bytes = off;
  return VG_OK;
}

}  // namespace

extern "C" {

size_t vg_augment_ws_bytes(const vg_aug_item* items, int n, int out_h, int out_w) {
  HostPlan p;
  if (plan(items, n, out_h, out_w, p, nullptr) != VG_OK) return 0;
  return p.total_bytes;
}

int vg_augment_batch(const vg_aug_item* items, int n, int out_h, int out_w, void* ws, size_t ws_bytes, float* out, void* stream) {
  VG_REQUIRE(ws && out, "augment: null pointer");
  HostPlan p;
  VG_REQUIRE(n > 0 && n <= 65535, "augment: batch of %d unsupported (1..65535)", n);
  // the image table goes to the device from a pinned staging buffer of this thread; an event guards its reuse
  struct Stage { int64_t* buf = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
  static thread_local Stage st;
  if (st.pending) { VG_HIP(hipEventSynchronize(st.ev)); st.pending = false; }
  if ((size_t)n * ITEM > st.cap) {
    if (st.buf) VG_HIP(hipHostFree(st.buf));
    st.buf = nullptr;
    st.cap = 0;
    VG_HIP(hipHostMalloc((void**)&st.buf, (size_t)n * ITEM * sizeof(int64_t), hipHostMallocDefault));
    st.cap = (size_t)n * ITEM;
  }
  if (!st.ev) VG_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
  int64_t* table = st.buf;
  int rc = plan(items, n, out_h, out_w, p, table);
  if (rc != VG_OK) return rc;
  VG_REQUIRE(ws_bytes >= p.total_bytes, "augment: workspace of %zu bytes, %zu needed (vg_augment_ws_bytes)", ws_bytes, p.total_bytes);
  const int S = out_w, OH = out_h;
  const int row_dwords = (p.max_cw * 3 + 3 + 3) / 4 + 1;
  const size_t lds_h = sizeof(int) * ((size_t)S * p.kmax + 2 * (size_t)S) + 4 * (size_t)row_dwords;
  const int rpb_v = 8;
  const size_t lds_v = sizeof(int) * ((size_t)rpb_v * p.kmax + 2 * rpb_v);
  VG_REQUIRE(lds_h <= 150 * 1024, "augment: down-scaling factor too large for one pass (%zu bytes of LDS needed, 153600 available)", lds_h);
  hipStream_t s = (hipStream_t)stream;
  VG_HIP(hipMemcpyAsync(ws, table, (size_t)n * ITEM * sizeof(int64_t), hipMemcpyHostToDevice, s));
  VG_HIP(hipEventRecord(st.ev, s));
  st.pending = true;
  VG_HIP(vg_lds_attr((const void*)aug_h_kernel, 150 * 1024));
  const int rpb_h = 16;
  dim3 gh(n, (p.max_rows + rpb_h - 1) / rpb_h);
  aug_h_kernel<<<gh, 256, lds_h, s>>>((const int64_t*)ws, OH, S, (uint8_t*)ws, rpb_h, p.kmax, row_dwords);
  VG_LAUNCH_CHECK();
  dim3 gv(n, (OH + rpb_v - 1) / rpb_v);
  aug_v_kernel<<<gv, 256, lds_v, s>>>((const int64_t*)ws, OH, S, (const uint8_t*)ws, out, rpb_v, p.kmax);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // extern "C"
