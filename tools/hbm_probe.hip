// Achievable HBM rates on this box: read-only (sum), write-only (fill) and copy kernels over 8 GiB with 16-byte
// accesses, plus a random 8 KB-row gather like the DISGAT kernels issue.  Build + run: hipcc --offload-arch=gfx950 -O3
// tools/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ p, size_t n, float* out) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_fill(f4* __restrict__ p, size_t n) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
// one wave per row id: gather a 8 KB row (64 lanes x 8 x 16 B), like the att-3 column operand
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ tab, const int* __restrict__ idx, int m, float* out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  f4 acc = {0, 0, 0, 0};
  for (int i = wave; i < m; i += nw) {
    const f4* r = tab + (size_t)idx[i] * 512 + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += r[j * 64];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}

template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  std::vector<float> t;
  for (int i = 0; i < 5; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[2];
}

int main() {
  const size_t bytes = (size_t)8 << 30, n = bytes / 16;
  f4 *a, *b; float* out; int* idx;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
  const int m = 4000000;
  std::vector<int> h(m); uint64_t s = 88172645463325252ull;
  for (int i = 0; i < m; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (bytes / 8192)); }
  hipMalloc(&idx, m * 4); hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice);
  hipMemset(a, 0, bytes);
  for (int grid : {2048, 8192, 32768}) {
    float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
    float w = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n); });
    float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    float g = timeit([&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, a, idx, m, out); });
    printf("grid %6d: read %.2f TB/s  fill %.2f TB/s  copy %.2f TB/s (read+write)  random 8KB-row gather %.2f TB/s\n", grid,
           bytes / r / 1e9, bytes / w / 1e9, 2.0 * bytes / c / 1e9, (double)m * 8192 / g / 1e9);
  }
  return 0;
}
