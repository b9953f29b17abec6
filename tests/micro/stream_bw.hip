// What does a plain streaming pass reach on this box?  The BatchNorm tensor passes are measured against it.
// Mixes: copy (1 read + 1 write), 2 reads + 1 write (backward apply), 2 reads (backward sums), 3 reads + 2 writes.
// Build: hipcc -O3 --offload-arch=gfx950 stream_bw.hip -o stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// NR reads, NW writes; CONTIG: a block walks its own contiguous chunk (as the BatchNorm kernels do), else grid-stride
template <int NR, int NW, int U, bool CONTIG, bool NT>
__global__ __launch_bounds__(256) void pass_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const f32x4* __restrict__ c,
                                                   f32x4* __restrict__ o, f32x4* __restrict__ o2, size_t n, float* sink) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  size_t i, end, step;
  if (CONTIG) { i = blockIdx.x * per + threadIdx.x; end = min(n, (size_t)(blockIdx.x + 1) * per); step = 256; }
  else { i = (size_t)blockIdx.x * 256 + threadIdx.x; end = n; step = (size_t)gridDim.x * 256; }
  f32x4 accum = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * step < end; i += U * step) {
    f32x4 va[U], vb[U], vc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      va[u] = NT ? __builtin_nontemporal_load(a + i + u * step) : a[i + u * step];
      if (NR > 1) vb[u] = NT ? __builtin_nontemporal_load(b + i + u * step) : b[i + u * step];
      if (NR > 2) vc[u] = NT ? __builtin_nontemporal_load(c + i + u * step) : c[i + u * step];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 v = va[u];
      if (NR > 1) v = v * 1.5f + vb[u];
      if (NR > 2) v = v + vc[u] * 0.5f;
      if (NW > 0) { if (NT) __builtin_nontemporal_store(v, o + i + u * step); else o[i + u * step] = v; }
      if (NW > 1) { if (NT) __builtin_nontemporal_store(v * 2.f, o2 + i + u * step); else o2[i + u * step] = v * 2.f; }
      if (NW == 0) accum += v;
    }
  }
  for (; i < end; i += step) {
    f32x4 v = a[i];
    if (NR > 1) v = v * 1.5f + b[i];
    if (NR > 2) v = v + c[i] * 0.5f;
    if (NW > 0) o[i] = v;
    if (NW > 1) o2[i] = v * 2.f;
    if (NW == 0) accum += v;
  }
  if (NW == 0 && accum[0] + accum[1] + accum[2] + accum[3] == 12345.678f) *sink = 1.f;
}

template <int NR, int NW, int U, bool CONTIG, bool NT>
static double run(f32x4** buf, size_t n, int blocks, float* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((pass_kernel<NR, NW, U, CONTIG, NT>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n, sink);
  CK(hipEventRecord(e0));
  const int it = 10;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((pass_kernel<NR, NW, U, CONTIG, NT>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)(NR + NW) * n * 16 / (ms / it * 1e-3) / 1e12;
}

template <int NR, int NW>
static void mix(const char* name, f32x4** buf, size_t n, float* sink) {
  const int grids[] = {1024, 2048, 4096, 8192, 16384};
  printf("%s, %zu MB per tensor\n", name, n * 16 >> 20);
  for (int g : grids) {
    printf("  blocks %5d: contig U2 %.2f U4 %.2f U8 %.2f | stride U1 %.2f U2 %.2f U4 %.2f | nt contig U4 %.2f stride U2 %.2f TB/s\n", g,
           run<NR, NW, 2, true, false>(buf, n, g, sink), run<NR, NW, 4, true, false>(buf, n, g, sink), run<NR, NW, 8, true, false>(buf, n, g, sink),
           run<NR, NW, 1, false, false>(buf, n, g, sink), run<NR, NW, 2, false, false>(buf, n, g, sink), run<NR, NW, 4, false, false>(buf, n, g, sink),
           run<NR, NW, 4, true, true>(buf, n, g, sink), run<NR, NW, 2, false, true>(buf, n, g, sink));
    fflush(stdout);
  }
}

int main() {
  float* sink; CK(hipMalloc(&sink, 4));
  for (size_t mb : {194u, 537u}) {
    const size_t n = mb * 1000000 / 16;
    f32x4* buf[5];
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&buf[i], n * 16)); CK(hipMemset(buf[i], 0, n * 16)); }
    mix<1, 1>("copy (1r + 1w)", buf, n, sink);
    mix<2, 1>("2r + 1w", buf, n, sink);
    mix<2, 0>("2r", buf, n, sink);
    mix<3, 2>("3r + 2w", buf, n, sink);
    for (int i = 0; i < 5; ++i) CK(hipFree(buf[i]));
  }
  return 0;
}
