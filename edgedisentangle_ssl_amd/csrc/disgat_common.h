// Shared device helpers for the DISGAT gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DISGAT_WAVES_PER_BLOCK
#define DISGAT_WAVES_PER_BLOCK 4
#endif
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

// Transposed multi-value reduction: every lane holds H = 2^HL partial sums v[0..H-1]; afterwards the
// lanes of head group h = lane >> (6-HL) all hold the full 64-lane sum of v[h].  Each butterfly step
// halves the number of live values: a lane keeps the half its lane bit selects and adds its
// partner's partial of the same values, so H values cost H-1 exchanges plus the (6-HL) in-group
// steps instead of H full reductions.  All exchanges stay on the VALU (no LDS round trips, which
// made a ds_bpermute version slower than H independent DPP reductions): gfx950's
// v_permlane32_swap / v_permlane16_swap do the 32- and 16-lane steps without selects; the 8- and
// 4-lane steps pair lane i with its mirror in the 16- / 8-lane row (DPP), any partner with the
// opposite lane bit works.
__device__ __forceinline__ float swap_add32(float a, float b) {   // lower half: a + a[lane+32]; upper: b + b[lane-32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {   // even 16-rows: a + a[lane+16]; odd rows: b + b[lane-16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int HL>
__device__ __forceinline__ float multi_reduce(float (&v)[1 << HL]) {
  constexpr int H = 1 << HL;
  const int lane = threadIdx.x & 63;
  int n = H;
  if constexpr (HL >= 1) {
    n >>= 1;
#pragma unroll
    for (int j = 0; j < (H >> 1); ++j) v[j] = swap_add32(v[j], v[(H >> 1) + j]);
  }
  if constexpr (HL >= 2) {
#pragma unroll
    for (int j = 0; j < (H >> 2); ++j) v[j] = swap_add16(v[j], v[(H >> 2) + j]);
  }
  if constexpr (HL >= 3) {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int j = 0; j < (H >> 3); ++j) {
      const float keep = up ? v[(H >> 3) + j] : v[j];
      const float send = up ? v[j] : v[(H >> 3) + j];
      v[j] = keep + dpp_f<0x140>(send);        // row_mirror: i <-> 15-i flips lane bit 3
    }
  }
  if constexpr (HL >= 4) {
    const bool up = (lane & 4) != 0;
    const float keep = up ? v[1] : v[0];
    const float send = up ? v[0] : v[1];
    v[0] = keep + dpp_f<0x141>(send);          // row_half_mirror: i <-> 7-i flips lane bit 2
  }
  (void)n;
  return group_sum<6 - HL>(v[0]);
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

// att 3 with a sign record for the backward pass: one bit per feature, (z > 0) = which slope leaky_relu
// took at z = P[r] + Q[c].  With those bits the score's backward needs no operand gather at all
// (edge_bwd.hip: seg_grad_sign_kernel).
//
// Record of one (edge | pair): 64 uint32 words in lane order.  Lane (h,g) owns features h*F_out + (j*G+g)*4 + k
// (j < QN float4 groups, k < 4 components).  Its word is eight NIBBLES: nibble k + 4*(j & 1) holds component k of the
// even (j & 1 = 0) or odd groups, group pair p = j >> 1 at bit QH-1-p of the nibble (QH = max(1, QN/2) <= 4):
//     bit 4*(k + 4*(j & 1)) + QH-1-(j >> 1).
// One shift and one and (0x11111111) then isolate the 8 bits of a group pair as eight 0 / 1 nibbles, which gfx950's
// v_cvt_scalef32_pk_f32_fp4 reads as fp4 values (0x1 = 0.5) and turns into floats two per instruction: 6 VALU ops per 8
// features in front of the 4 v_pk_fma_f32 that accumulate them (round 2's byte-per-component layout with
// v_cvt_f32_ubyteN: 12).
//
// sign_push: shift left and append (z > 0).  z > 0 as an integer test on the float's bits s: s > 0 as int32
// (negative floats and -0.0 carry the sign bit, +0.0 is 0), i.e. the sign bit of the saturating 0 - s
// (saturation keeps -0.0 = INT_MIN on the "not positive" side); v_alignbit appends it: 2 VALU ops per feature.
__device__ __forceinline__ uint32_t sign_push(uint32_t bits, float z) {
  const int t = __builtin_elementwise_sub_sat(0, __float_as_int(z));
  return __builtin_amdgcn_alignbit(bits, (uint32_t)t, 31);
}
struct SignAcc {
  uint32_t n[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};       // one accumulator per nibble: a group pair's bit per push
  __device__ __forceinline__ uint32_t word() const {
    return n[0] | (n[1] << 4) | (n[2] << 8) | (n[3] << 12) | (n[4] << 16) | (n[5] << 20) | (n[6] << 24) | (n[7] << 28);
  }
};
// dot4_lrelu that also pushes the 4 signs of float4 group j (a compile-time index after unrolling; kept scalar per
// feature: routed through a float4 temporary the compiler forgets that z is the canonical result of an add and spends a
// second v_max per feature on fmaxf)
__device__ __forceinline__ float dot4_lrelu_sign(const f32x4 a, const f32x4 p, const f32x4 q, float acc, SignAcc& s, int j) {
  const int o = 4 * (j & 1);
  float z;
  z = p.x + q.x; acc = fmaf(a.x, lrelu001(z), acc); s.n[o + 0] = sign_push(s.n[o + 0], z);
  z = p.y + q.y; acc = fmaf(a.y, lrelu001(z), acc); s.n[o + 1] = sign_push(s.n[o + 1], z);
  z = p.z + q.z; acc = fmaf(a.z, lrelu001(z), acc); s.n[o + 2] = sign_push(s.n[o + 2], z);
  z = p.w + q.w; acc = fmaf(a.w, lrelu001(z), acc); s.n[o + 3] = sign_push(s.n[o + 3], z);
  return acc;
}
// The record bits of group pair p (groups 2p and 2p+1) of a sign word as 0.0 / sign_unit(): ev = group 2p, od = group 2p+1.
__device__ __forceinline__ float sign_unit() { return __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(0x00000001u, 1.0f, 0)[0]; }
template <int QN>
__device__ __forceinline__ void sign_floats2(uint32_t w, int p, f32x4& ev, f32x4& od) {
  constexpr int QH = QN >= 2 ? QN / 2 : 1;
  uint32_t t = (w >> (QH - 1 - p)) & 0x11111111u;
  asm("" : "+v"(t));   // opaque: otherwise the nibbles are re-derived as single-bit extracts (shift + and + cvt each)
  const auto e0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(t, 1.0f, 0);
  const auto e1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(t, 1.0f, 1);
  ev = f32x4{e0[0], e0[1], e1[0], e1[1]};
  if (QN >= 2) {
    const auto o0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(t, 1.0f, 2);
    const auto o1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(t, 1.0f, 3);
    od = f32x4{o0[0], o0[1], o1[0], o1[1]};
  }
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
// same mask from the same (seed, edge, head).  seed_dev (or NULL): a device counter added to the seed - a step captured
// in a HIP graph bakes `seed` in, so the per-step variation comes from memory the graph advances itself.
struct DropCfg {
  uint64_t seed;
  uint32_t thresh;   // 0 = dropout off
  float scale;       // 1/(1-p)
  const uint64_t* seed_dev;
};
__device__ __forceinline__ float drop_mult(const DropCfg& d, int64_t k, int h, int H) {
  if (d.thresh == 0u) return 1.0f;
  const uint64_t seed = d.seed + (d.seed_dev ? *d.seed_dev : 0ull);        // wave-uniform: one scalar load, hoisted
  uint64_t z = seed + (uint64_t)(k * H + h) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((uint32_t)(z >> 32) >= d.thresh) ? d.scale : 0.0f;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// gathered operand rows that are read once per use and never hit a cache (8 GB tables, uniformly random rows): A/B builds
// (-DDISGAT_NT_GATHER=1) fetch them with the non-temporal hint
__device__ __forceinline__ f32x4 ld4g(const float* p) {
#ifdef DISGAT_NT_GATHER
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace disgat
