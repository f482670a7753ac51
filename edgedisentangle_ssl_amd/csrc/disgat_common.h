// Shared device helpers for the DISGAT gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DISGAT_WAVES_PER_BLOCK 4
#define DISGAT_BLOCK (64 * DISGAT_WAVES_PER_BLOCK)

namespace disgat {

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Sum over the 2^GL consecutive lanes that share the upper lane bits; every lane of the
// group ends with the same total (the pairing tree is symmetric, fp add is commutative).
// Steps 1-4 are DPP (no LDS traffic); 32/64-lane groups finish through ds_bpermute.
template <int GL>
__device__ __forceinline__ float group_sum(float v) {
  if (GL >= 1) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  if (GL >= 2) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  if (GL >= 3) v += dpp_f<0x141>(v);   // row_half_mirror      : i <-> 7-i  within 8
  if (GL >= 4) v += dpp_f<0x140>(v);   // row_mirror           : i <-> 15-i within 16
  if (GL >= 5) v += __shfl_xor(v, 16, 64);
  if (GL >= 6) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float e) { return 1.0f / (1.0f + expf(-e)); }

// softmax numerator of the reference: exp(sigmoid(e) - shift).  sigmoid(e) is in (0,1), so no
// max-shift is needed for range; utils.py:194 subtracts a GLOBAL max, which cancels exactly in
// the ratio, and its +1e-10 (utils.py:198) is below half an ulp of any non-empty row's sum.
__device__ __forceinline__ float softmax_num(float e) { return expf(sigmoidf_(e)); }

__device__ __forceinline__ float lrelu001(float z) { return fmaxf(z, 0.01f * z); }

__device__ __forceinline__ float dot4_lrelu(const f32x4 a, const f32x4 p, const f32x4 q, float acc) {
  acc = fmaf(a.x, lrelu001(p.x + q.x), acc);
  acc = fmaf(a.y, lrelu001(p.y + q.y), acc);
  acc = fmaf(a.z, lrelu001(p.z + q.z), acc);
  acc = fmaf(a.w, lrelu001(p.w + q.w), acc);
  return acc;
}

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}

// Attention dropout (layers.py:394: F.dropout on the normalised attention, training only).
// Counter-based: one splitmix64 hash of (seed, edge*H + head); keep iff the top 32 bits >= thresh
// (thresh = p * 2^32).  Returns the multiplier mask/(1-p).  Forward and backward regenerate the
// same mask from the same (seed, edge, head).
struct DropCfg {
  uint64_t seed;
  uint32_t thresh;   // 0 = dropout off
  float scale;       // 1/(1-p)
};
__device__ __forceinline__ float drop_mult(const DropCfg& d, int64_t k, int h, int H) {
  if (d.thresh == 0u) return 1.0f;
  uint64_t z = d.seed + (uint64_t)(k * H + h) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((uint32_t)(z >> 32) >= d.thresh) ? d.scale : 0.0f;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace disgat
