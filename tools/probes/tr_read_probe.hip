#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
  __shared__ __fp16 img[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) img[i] = (__fp16)(float)((i / 128) * 100 + (i % 128));
  __syncthreads();
  const int l = threadIdx.x, q = (l & 15) >> 2, p = l & 3, g = l >> 4;
  fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(&img[(8 * g + q) * 128 + 4 * p]));
  out[l * 4 + 0] = (float)v[0]; out[l * 4 + 1] = (float)v[1]; out[l * 4 + 2] = (float)v[2]; out[l * 4 + 3] = (float)v[3];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 5, 15, 16, 17, 33, 63}) printf("lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
