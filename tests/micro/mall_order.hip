// Does the traversal ORDER of a consumer kernel matter for the 256 MB Infinity Cache?  K1 streams a tensor out in
// ascending block order; K2 (y = f(x), the shape of a BatchNorm apply pass) reads it in ascending or DESCENDING order.
// Build: hipcc --offload-arch=gfx950 -O3 -o tests/micro/mall_order tests/micro/mall_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// a block owns a contiguous chunk of `per` 16-byte pieces, like the BatchNorm kernels (bn.hip)
__global__ __launch_bounds__(256) void produce(u32x4* __restrict__ out, size_t per, size_t n, unsigned seed) {
  const size_t b = blockIdx.x;
  const size_t lo = b * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) out[i] = u32x4{(unsigned)i, seed, 3u, 4u};
}
__global__ __launch_bounds__(256) void consume(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t per, size_t n, int reverse) {
  const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const size_t lo = b * per, hi = lo + per < n ? lo + per : n;
  if (lo >= hi) return;
  if (reverse) {
    for (long i = (long)hi - 256 + threadIdx.x; i >= (long)lo; i -= 256) { u32x4 v = in[i]; v[1] += 1u; out[i] = v; }   // descending inside the chunk too
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) { u32x4 v = in[i]; v[1] += 1u; out[i] = v; }
  }
}
__global__ __launch_bounds__(256) void reduce_only(const u32x4* __restrict__ in, unsigned* __restrict__ sink, size_t per, size_t n, int reverse) {
  const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const size_t lo = b * per, hi = lo + per < n ? lo + per : n;
  unsigned s = 0;
  if (reverse) { for (long i = (long)hi - 256 + threadIdx.x; i >= (long)lo; i -= 256) { u32x4 v = in[i]; s += v[0] ^ v[1]; } }
  else { for (size_t i = lo + threadIdx.x; i < hi; i += 256) { u32x4 v = in[i]; s += v[0] ^ v[1]; } }
  if (s == 0x12345u) sink[0] = s;
}

int main() {
  const size_t sizes_mb[] = {1074, 537, 268, 134};
  u32x4 *a, *b; unsigned* sink;
  CK(hipMalloc(&a, 1074ull << 20)); CK(hipMalloc(&b, 1074ull << 20)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t mb : sizes_mb) {
    const size_t n = (mb << 20) / 16;
    const int blocks = 2048;
    const size_t per = ((n + blocks - 1) / blocks + 255) / 256 * 256;
    for (int kind = 0; kind < 2; ++kind) {          // 0: y = f(x) (read + write), 1: read only (a reduction pass)
      for (int reverse = 0; reverse < 2; ++reverse) {
        float best = 1e9f, sum = 0.f;
        const int reps = 6;
        for (int r = 0; r < reps; ++r) {
          hipLaunchKernelGGL(produce, dim3(blocks), dim3(256), 0, 0, a, per, n, (unsigned)r);
          CK(hipEventRecord(e0, 0));
          if (kind == 0) hipLaunchKernelGGL(consume, dim3(blocks), dim3(256), 0, 0, a, b, per, n, reverse);
          else hipLaunchKernelGGL(reduce_only, dim3(blocks), dim3(256), 0, 0, a, sink, per, n, reverse);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0) { sum += ms; best = ms < best ? ms : best; }
        }
        const double bytes = (double)n * 16 * (kind == 0 ? 2 : 1);
        printf("%4zu MB tensor, %s, consumer %s: avg %.3f ms (best %.3f) = %.2f TB/s\n", mb, kind == 0 ? "y=f(x) " : "read only",
               reverse ? "DESCENDING" : "ascending ", sum / (reps - 1), best, bytes / (sum / (reps - 1)) / 1e9);
      }
    }
  }
  CK(hipGetLastError());
  return 0;
}
