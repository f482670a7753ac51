// Helpers shared by the f16x3 GEMM kernels (gemm_split.hip, gemm_planes.hip) and by the kernels that PRODUCE
// their operands already split (edge_fwd.hip: the aggregated Z rows; gemm_planes.hip: the head buffer, the fuser output).
//
// The f16x3 operand format ("planes"): t = v * s with s = f16_scale(bound), bound >= max |v| over the operand;
//   hi = fp16(t),  lo = fp16((t - hi) * 2^11)       both round-to-nearest, t - hi exact in fp32
// so t = hi + lo * 2^-11 to 2^-23 |t| for every element whose hi is a normal fp16.  A producer and its consumer agree
// on s through ONE device scalar holding the bound: both evaluate f16_scale() on it.
#pragma once
#include "disgat_common.h"

namespace disgat {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// power-of-two scale that places amax in [2^13, 2^14); 1 for zero / non-finite input
__device__ __forceinline__ float f16_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);                       // amax = m * 2^e, m in [0.5, 1)
  e = max(-100, min(100, e));
  return ldexpf(1.0f, 14 - e);
}

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  const f16x2 h = {(_Float16)a, (_Float16)b};
  return *reinterpret_cast<const uint32_t*>(&h);
}

// split 4 floats (already scaled) into hi / lo fp16 planes, packed 2 x uint32 each
__device__ __forceinline__ void split4h(const f32x4 t, u32x2& hi, u32x2& lo) {
  const _Float16 h0 = (_Float16)t.x, h1 = (_Float16)t.y, h2 = (_Float16)t.z, h3 = (_Float16)t.w;
  const f16x2 a = {h0, h1}, b = {h2, h3};
  hi = u32x2{*reinterpret_cast<const uint32_t*>(&a), *reinterpret_cast<const uint32_t*>(&b)};
  lo = u32x2{pack_f16((t.x - (float)h0) * 2048.f, (t.y - (float)h1) * 2048.f),
             pack_f16((t.z - (float)h2) * 2048.f, (t.w - (float)h3) * 2048.f)};
}

template <int ACT>
__device__ __forceinline__ float act_ct(float v, float slope) {
  if (ACT == 1) return v > 0.f ? v : __expf(v) - 1.0f;
  if (ACT == 2) return v > 0.f ? v : slope * v;
  return v;
}


// Arguments of the fp32-operand f16x3 launchers (disgat_gemm_f16x3: gemm_split.hip, gemm_rs.hip)
struct GemmHArgs {
  const float* A;
  int64_t lda, a_bs;
  const uint16_t* Bt;    // [batch][2][N][K] fp16: hi, lo of (B^T * s_B), k contiguous
  const float* a_amax;   // device scalar: max |A| over the whole operand (disgat_amax)
  const float* b_scale;  // device scalar: s_B used for Bt
  const float* bias;
  const float* init;
  int64_t ldi, i_bs;
  float* C;
  int64_t ldc, c_bs;
  int M, N, K, batch;
  int act;
  float slope;
  int mt, nt;
};

// gemm_rs.hip: the register-stationary kernel for K in {64, 128, 256} (host side)
bool gemm_rs_takes(int N, int K);
int launch_gemm_f16x3_rs(const GemmHArgs& G, hipStream_t st);

}  // namespace disgat
