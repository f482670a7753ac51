// fp32-accurate dense GEMMs on the 16-bit matrix cores of gfx950 (CDNA4 has no xf32/TF32 and its
// f32-input MFMA runs at the vector rate, 1/16 of bf16 / fp16).
//
//   C[b] = act( A[b] (M x K, fp32) * B[b] (K x N) + bias[b] + init[b] )        b = 0 .. batch-1
//
// Two operand-splitting schemes live in this file: (1) below, three bf16 planes / six products, operands exact
// (`DISGAT_GEMM=split6`); (2) further down, two scaled fp16 planes / three products, the default - its tiled
// kernel for K > 256 and its A-stationary kernel for K <= 256 - plus the helpers (max |A|, weight preparation,
// activation backward).
//
// Scheme 1.  Every fp32 value v is written exactly as hi + mid + lo with three bf16 numbers (8 + 8 + 8
// mantissa bits); a product a*b then needs the six partial
// products whose weight is >= 2^-16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi), each exact in
// the MFMA's fp32 accumulator; the dropped terms are <= 2^-25 relative, below fp32 rounding.
// Measured error vs fp64 equals hipBLASLt's fp32 GEMM (tests/test_gpu_gemm.py).  `terms` = 3
// keeps only hi*hi, hi*mid, mid*hi (~5e-6 of the tensor maximum: opt-in).
//
// These are the dense contractions of the DISGAT path (SURVEY 8d "MFMA" rows): the per-node score
// operands P/Q = x W (layers.py:350, 363, 376), the per-head output projections (layers.py:398,
// 110, 39), FuseLayer's Linear (layers.py:905) and DifHead's MLP (models.py:538).
//
// Tried and dropped (measured on MI355X, M = 1e6): a wave-specialised persistent variant (4 loader
// waves + 4 MFMA waves, LDS double buffer, one block per CU) ran 3-8 % slower: with the A operand
// split on the fly the loader path (VALU split + 18 ds_writes per thread and K-step) costs about as
// much as the 96 MFMAs and shares their SIMD issue slots; an ablation without MFMAs still needed
// ~1.0 us per K-step.  A DMA-staged variant (global_load_lds into a 3-stage ring, counted vmcnt across
// raw barriers, A split from its LDS fragments, 4 or 8 waves per block, one block per CU) was correct
// but reached only 114-167 TFLOP/s against this kernel's 147-187.  For scale: the 6-product scheme's
// ceiling is 417 TFLOP/s at the nominal MFMA rate, ~330 at the clock bf16 MFMA loops hold on random
// data, and the guide's hand-tuned 8-phase bf16 GEMM sustains ~55 % of peak, i.e. ~230 TFLOP/s here.
//
// Structure: 128x128 output tile per 256-thread block (4 waves as 2x2, 64x64 each = 4x4 MFMA
// tiles of 16x16x32), BK = 32.  A is read as fp32 (coalesced float4), split in registers and
// written to three LDS planes; B arrives pre-split and pre-transposed ([3][N][K] bf16, k
// contiguous) so both operands are one ds_read_b128 per fragment; 64-B rows with an XOR chunk
// swizzle make the 16-row x 16-B fragment reads conflict-free.  Next tile's global loads are issued
// before the 96 MFMAs of the current one (two-tile-deep register prefetch, single LDS stage, 48 KB, 2
// blocks/CU).  Block ids are remapped so that all N-tiles of one M-tile run on one XCD (A tile
// served by that XCD's L2).  Epilogue fuses bias, an additive init matrix and ELU / leaky-ReLU.
#include "gemm_common.h"
#include <type_traits>

namespace disgat {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmArgs {
  const float* A;
  int64_t lda, a_bs;
  const uint16_t* Bt;   // [batch][3][N][K]
  const float* bias;    // [batch][N] or null
  const float* init;    // or null
  int64_t ldi, i_bs;
  float* C;
  int64_t ldc, c_bs;
  int M, N, K, batch;
  int act;              // 0 none, 1 elu, 2 leaky relu(slope)
  float slope;
  int terms;            // 6 or 3
  int mt, nt;           // tiles along M, N
};

constexpr int GBM = 128, GBN = 128, GBK = 32, GLD = 32;   // LDS rows: 32 bf16 = 64 B, four 16-B chunks
constexpr int PLANE = GBM * GLD;                          // bf16 elements per plane (A and B tiles equal)

// Exact three-way split of an fp32 value by truncation: hi = top 8 mantissa bits, mid = the next 8,
// lo = the last 8 (v == hi + mid + lo exactly; each residual subtraction is exact).  Truncation costs
// 5 VALU ops per value instead of ~14 for round-to-nearest; the dropped cross terms (mid*lo, lo*mid,
// lo*lo) stay below 2^-22 of the product and share its sign, i.e. a relative scaling of ~1e-7.
__device__ __forceinline__ void split1(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(v) & 0xFFFF0000u;
  const float r1 = v - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m)) & 0xFFFF0000u;
}

// two bf16 (upper halves of a and b) -> one dword, a in the low half
// Epilogue activation, branch-free (a per-element expm1f call drags a branchy libm body into every
// unrolled store): 0 none, 1 ELU(alpha 1) with exp(v)-1 (absolute error ~1e-7), 2 leaky ReLU.
__device__ __forceinline__ float act_fn(float v, int act, float slope) {
  const float neg = (act == 1) ? (__expf(v) - 1.0f) : ((act == 2) ? slope * v : v);
  return v > 0.f ? v : neg;
}

// LDS element offset of 16-B chunk `c` (0..3) of tile row `r`: chunk index XOR-swizzled with
// (r >> 1) & 3, which makes the 16-row x 16-B MFMA fragment reads conflict-free for all four lane
// groups of ds_read_b128 (unswizzled 64-B rows are 2-way conflicted) without padding.
__device__ __forceinline__ int sw(int r, int c) { return r * GLD + ((c ^ ((r >> 1) & 3)) << 3); }

__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// split 4 floats into 3 planes of 4 bf16 (packed 2 x uint32 per plane; element 0 in the low half)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
  split1(v.x, h0, m0, l0);
  split1(v.y, h1, m1, l1);
  split1(v.z, h2, m2, l2);
  split1(v.w, h3, m3, l3);
  hi = u32x2{pack_hi(h0, h1), pack_hi(h2, h3)};
  mid = u32x2{pack_hi(m0, m1), pack_hi(m2, m3)};
  lo = u32x2{pack_hi(l0, l1), pack_hi(l2, l3)};
}

template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmArgs G) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[6 * PLANE];   // A planes 0..2, B planes 3..5
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD; give each XCD whole M-tiles
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int m_tile = (q / G.nt) * 8 + xcd;
  const int n_tile = q % G.nt;
  if (m_tile >= G.mt) return;
  const int bz = blockIdx.y;
  const int m0 = m_tile * GBM, n0 = n_tile * GBN;

  const float* A = G.A + (int64_t)bz * G.a_bs;
  const uint16_t* Bt = G.Bt + (int64_t)bz * 3 * G.N * G.K;
  float* C = G.C + (int64_t)bz * G.c_bs;

  // global -> register staging maps
  const int a_row = tid >> 3, a_col = (tid & 7) * 4;          // + 32*i rows, i < 4
  const int b_row = tid >> 2, b_col = (tid & 3) * 8;          // + 64*i rows, i < 2, per plane
  f32x4 a_s0[4], a_s1[4];
  u32x4 b_s0[6], b_s1[6];

  auto load_tiles = [&](f32x4(&a_st)[4], u32x4(&b_st)[6], int kt) {
    const int k0 = kt * GBK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = m0 + a_row + 32 * i;
      a_st[i] = (r < G.M) ? ld4(A + (int64_t)r * G.lda + k0 + a_col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + b_row + 64 * i;
        b_st[p * 2 + i] = *reinterpret_cast<const u32x4*>(Bt + ((int64_t)p * G.N + n) * G.K + k0 + b_col);
      }
  };
  auto store_tiles = [&](const f32x4(&a_st)[4], const u32x4(&b_st)[6]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 h, m, l;
      split4(a_st[i], h, m, l);
      const int o = sw(a_row + 32 * i, a_col >> 3) + (a_col & 4);
      *reinterpret_cast<u32x2*>(&lds[0 * PLANE + o]) = h;
      *reinterpret_cast<u32x2*>(&lds[1 * PLANE + o]) = m;
      *reinterpret_cast<u32x2*>(&lds[2 * PLANE + o]) = l;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<u32x4*>(&lds[(3 + p) * PLANE + sw(b_row + 64 * i, b_col >> 3)]) = b_st[p * 2 + i];
  };

  f32x4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int frag_off = sw(lane & 15, lane >> 4);      // tile bases are multiples of 16 rows: swizzle unchanged
  const int KT = G.K / GBK;

  auto compute = [&]() {
    bf16x8 af[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[p][i] = *reinterpret_cast<const bf16x8*>(&lds[p * PLANE + (wm * 64 + i * 16) * GLD + frag_off]);
#pragma unroll
    for (int pb = 0; pb < (TERMS == 3 ? 2 : 3); ++pb) {
      bf16x8 bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&lds[(3 + pb) * PLANE + (wn * 64 + j * 16) * GLD + frag_off]);
      // plane pairs (pa,pb) kept: pb=0: pa 0,1,2 ; pb=1: pa 0,1 ; pb=2: pa 0   (TERMS==3: (0,0),(1,0),(0,1))
      const int npa = (TERMS == 3) ? (pb == 0 ? 2 : 1) : (3 - pb);
#pragma unroll
      for (int pa = 0; pa < npa; ++pa)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pa][i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  // pipeline: tile kt is in LDS, tile kt+1 sits in one register set (loaded during the previous
  // step), tile kt+2 is being loaded into the other set while the MFMAs of tile kt run
  load_tiles(a_s0, b_s0, 0);
  store_tiles(a_s0, b_s0);
  if (KT > 1) load_tiles(a_s1, b_s1, 1);
  __syncthreads();
  for (int kt = 0; kt < KT; kt += 2) {
    if (kt + 2 < KT) load_tiles(a_s0, b_s0, kt + 2);
    compute();
    __syncthreads();
    if (kt + 1 < KT) {
      store_tiles(a_s1, b_s1);
      __syncthreads();
      if (kt + 3 < KT) load_tiles(a_s1, b_s1, kt + 3);
      compute();
      __syncthreads();
      if (kt + 2 < KT) {
        store_tiles(a_s0, b_s0);
        __syncthreads();
      }
    }
  }

  // epilogue: C/D layout of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg.  The tile goes
  // through LDS in two 64-row halves (wm = 0, then wm = 1) so that every thread stores float4s of
  // full 512-B rows instead of 64 scattered dwords.
  const float* bias = G.bias ? G.bias + (int64_t)bz * G.N : nullptr;
  const float* init = G.init ? G.init + (int64_t)bz * G.i_bs : nullptr;
  float* stage = reinterpret_cast<float*>(lds);             // 64 rows x (128 + 4) floats = 33 KB
  constexpr int SLD = GBN + 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(i * 16 + (lane >> 4) * 4 + r) * SLD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
    const int c4 = (tid & 31) * 4;                          // 32 threads x float4 = one 128-float row
    const int col = n0 + c4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = ld4(bias + col);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int lr = (tid >> 5) + rr * 8;
      const int row = m0 + half * 64 + lr;
      if (row < G.M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&stage[lr * SLD + c4]) + bv;
        if (init) v += ld4(init + (int64_t)row * G.ldi + col);
        v.x = act_fn(v.x, G.act, G.slope); v.y = act_fn(v.y, G.act, G.slope);
        v.z = act_fn(v.z, G.act, G.slope); v.w = act_fn(v.w, G.act, G.slope);
        st4(C + (int64_t)row * G.ldc + col, v);
      }
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------
// Second scheme, half the MFMA work: two fp16 planes per operand instead of three bf16 planes.
//   t = v * s (s = power of two placing the operand's largest magnitude in [2^13, 2^14)),
//   hi = fp16(t), lo = fp16((t - hi) * 2^11)         both round-to-nearest, t - hi exact in fp32
// so t = hi + lo * 2^-11 up to 2^-23 |t| for every element whose hi is a normal fp16 (within 2^-27 of the
// operand's maximum) - one bit short of fp32's own 2^-24 representation error.  A product needs hi*hi and the
// two cross terms (lo*lo is 2^-24 relative): 3 MFMAs instead of 6.  The cross terms carry a 2^-11 weight
// and accumulate in their own fp32 accumulator (folding the weight into lo would push lo into fp16's
// subnormal range for all but the largest elements).  C = (acc_hh + 2^-11 acc_x) / (s_A s_B).
// Measured error vs fp64: tests/test_gpu_gemm.py (yardstick hipBLASLt fp32).

// struct GemmHArgs: gemm_common.h (shared with gemm_rs.hip)

__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(const GemmHArgs G) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[17408];      // 4 planes (32 KB) | epilogue stage (33.8 KB)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int m_tile = (q / G.nt) * 8 + xcd;
  const int n_tile = q % G.nt;
  if (m_tile >= G.mt) return;
  const int bz = blockIdx.y;
  const int m0 = m_tile * GBM, n0 = n_tile * GBN;

  const float* A = G.A + (int64_t)bz * G.a_bs;
  const uint16_t* Bt = G.Bt + (int64_t)bz * 2 * G.N * G.K;
  float* C = G.C + (int64_t)bz * G.c_bs;
  const float sA = f16_scale(*G.a_amax);
  const float inv = 1.0f / (sA * *G.b_scale);

  const int a_row = tid >> 3, a_col = (tid & 7) * 4;
  const int b_row = tid >> 2, b_col = (tid & 3) * 8;
  // A (HBM) is prefetched two tiles ahead, B (the small weight, L2-resident) one tile ahead: 48 staging registers
  f32x4 a_s0[4], a_s1[4];
  u32x4 b_s[4];

  auto load_a = [&](f32x4(&a_st)[4], int kt) {
    const int k0 = kt * GBK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = m0 + a_row + 32 * i;
      a_st[i] = (r < G.M) ? ld4(A + (int64_t)r * G.lda + k0 + a_col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto load_b = [&](u32x4(&b_st)[4], int kt) {
    const int k0 = kt * GBK;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + b_row + 64 * i;
        b_st[p * 2 + i] = *reinterpret_cast<const u32x4*>(Bt + ((int64_t)p * G.N + n) * G.K + k0 + b_col);
      }
  };
  auto store_tiles = [&](const f32x4(&a_st)[4], const u32x4(&b_st)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 h, l;
      split4h(a_st[i] * sA, h, l);
      const int o = sw(a_row + 32 * i, a_col >> 3) + (a_col & 4);
      *reinterpret_cast<u32x2*>(&lds[0 * PLANE + o]) = h;
      *reinterpret_cast<u32x2*>(&lds[1 * PLANE + o]) = l;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<u32x4*>(&lds[(2 + p) * PLANE + sw(b_row + 64 * i, b_col >> 3)]) = b_st[p * 2 + i];
  };

  f32x4v acc[4][4], acx[4][4];      // hi*hi | hi*lo + lo*hi (weight 2^-11)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      acx[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }

  const int frag_off = sw(lane & 15, lane >> 4);
  const int KT = G.K / GBK;

  auto compute = [&]() {
    // the hi fragments stay live for all three products; the lo fragments are streamed one at a time
    f16x8 ah[4], bh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(&lds[0 * PLANE + (wm * 64 + i * 16) * GLD + frag_off]);
      bh[i] = *reinterpret_cast<const f16x8*>(&lds[2 * PLANE + (wn * 64 + i * 16) * GLD + frag_off]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x8 bl = *reinterpret_cast<const f16x8*>(&lds[3 * PLANE + (wn * 64 + j * 16) * GLD + frag_off]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acx[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f16x8 al = *reinterpret_cast<const f16x8*>(&lds[1 * PLANE + (wm * 64 + i * 16) * GLD + frag_off]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acx[i][j], 0, 0, 0);
    }
  };

  load_a(a_s0, 0);
  load_b(b_s, 0);
  store_tiles(a_s0, b_s);
  if (KT > 1) load_a(a_s1, 1);
  __syncthreads();
  for (int kt = 0; kt < KT; kt += 2) {
    if (kt + 2 < KT) load_a(a_s0, kt + 2);
    if (kt + 1 < KT) load_b(b_s, kt + 1);
    compute();
    __syncthreads();
    if (kt + 1 < KT) {
      store_tiles(a_s1, b_s);
      __syncthreads();
      if (kt + 3 < KT) load_a(a_s1, kt + 3);
      if (kt + 2 < KT) load_b(b_s, kt + 2);
      compute();
      __syncthreads();
      if (kt + 2 < KT) {
        store_tiles(a_s0, b_s);
        __syncthreads();
      }
    }
  }

  const float* bias = G.bias ? G.bias + (int64_t)bz * G.N : nullptr;
  const float* init = G.init ? G.init + (int64_t)bz * G.i_bs : nullptr;
  float* stage = reinterpret_cast<float*>(lds);
  constexpr int SLD = GBN + 4;
  const float xw = inv * (1.0f / 2048.f);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(i * 16 + (lane >> 4) * 4 + r) * SLD + wn * 64 + j * 16 + (lane & 15)] =
                fmaf(acx[i][j][r], xw, acc[i][j][r] * inv);
    }
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    const int col = n0 + c4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = ld4(bias + col);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int lr = (tid >> 5) + rr * 8;
      const int row = m0 + half * 64 + lr;
      if (row < G.M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&stage[lr * SLD + c4]) + bv;
        if (init) v += ld4(init + (int64_t)row * G.ldi + col);
        v.x = act_fn(v.x, G.act, G.slope); v.y = act_fn(v.y, G.act, G.slope);
        v.z = act_fn(v.z, G.act, G.slope); v.w = act_fn(v.w, G.act, G.slope);
        st4(C + (int64_t)row * G.ldc + col, v);
      }
    }
    __syncthreads();
  }
}

// A-stationary variant for K <= 256 (the P/Q, per-head projection and classifier GEMMs).  Ablation of the tiled
// kernel above on [1e6,256] x [256,2048] (6.0 ms): the MFMAs account for 0.5 ms, re-reading the A tile once per
// 128-column tile for 1.9 ms, the LDS-staged epilogue for 2.5 ms, and the phases hardly overlap.  Here a 512-thread
// block keeps its whole 128 x K A tile, split once, in LDS (2 planes x 128 x (K+8) fp16 = 132 KB at K = 256) and
// sweeps all N columns in steps of 256: wave w owns the 128 x 32 strip [32w, 32w+32) of the step, reads A fragments
// from LDS and B fragments straight from global memory (the weight planes are L2-resident, k contiguous = the
// MFMA B layout; one k-step prefetched), and stores its results directly - after one v_permlane32_swap per register
// pair every store instruction writes two full 128-B rows.  No barrier after the A tile is in place.
// What still bounds it (ablations at [1e6,256] x [256,2048], 3.8 ms): the 8 GB of results leave each CU at its share
// of the HBM write rate (~21 GB/s): a wave's 64 stores per N-step block in the issue queue, and the other wave's B
// fragment loads queue behind them in the same vector-memory path (no epilogue: -1.5 ms; no B reloads: -1.0 ms).
// Tried without gain: issuing all LDS fragment reads ahead of the MFMAs (the two waves per SIMD already hide that
// latency), a per-SIMD token that forces the two waves into anti-phase (MFMA loop vs stores), and a variant with
// 64 x 32 wave tiles and two accumulator sets that retires the previous step's results 4 stores per k-step inside
// the next MFMA loop (correct, 4.6 ms against 3.5: twice the B-fragment traffic and twice the steps cost more than
// the overlap returns).  Round 2, same-box A/B at [1e6,256] x [256,2048] (3.43-3.50 ms): accumulating the product transposed so
// that the epilogue is 16 sixteen-byte stores per step instead of 64 four-byte ones (kept: simpler, same time, 3.50 vs
// 3.51); the conflict-free LDS pad below (kept, <1 %); a 64-row tile with two blocks per CU for N <= 256 (5.75 vs 5.10 ms on
// the 8-head projection: half the reuse of every weight fragment); a variant that keeps the wave's whole weight strip in
// 128 VGPRs per column step, walks the row tiles two at a time and drips its stores, second wave of each SIMD half a step
// out of phase (3.80-3.83 vs 3.43-3.45 ms); a chained projection -> ELU -> fuser kernel that never materialises the
// [M, H*N1] head buffer (12.5 vs 5.1 + 4.2 ms: 64-row tiles again, weight planes through the 64 B/clk vector-memory path);
// a persistent form (one block per CU walking the tiles, the next tile's fp32 rows requested into registers between the
// current tile's last stores, LDS-only barriers): 4.70 vs 5.10 ms on the 8-head projection alone, but the whole bench
// step did not get faster with it (T_iter 575.7 vs 568.2 ms on one box) - profiles/r02/experiments.md.
// LDS row = K + 16 fp16 (K % 128 == 0 or K % 32 == 0 rows of 64 B multiples): row stride = 2 sixteen-byte slots mod 16.
// ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59},
// {36-43,48-51,60-63} (MI355X_MICROARCH.md, LDS): with fragment row = lane & 15 and k-chunk = lane >> 4 a group mixes
// rows {0-3,12-15} of one chunk with rows {4-11} of the next, so a one-slot pad (K + 8, the round-1 layout) puts rows 11
// and 12 on one slot - one extra LDS cycle per group, 8 instead of 4 per read (SQ_LDS_BANK_CONFLICT = 3.9 per LDS
// instruction, profiles/r02/gemm_f16x3_pmc.md).  Pads of 2, 6, 10, 14 slots are conflict-free for all four groups.
constexpr int AS_BM = 128, AS_PAD = 16;

template <int ACT>
__global__ __launch_bounds__(512, 1) void gemm_f16x3_as_kernel(const GemmHArgs G) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_as[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bz = blockIdx.y;
  const int m0 = blockIdx.x * AS_BM;
  const int K = G.K, KP = K + AS_PAD, K4 = K >> 2;
  const int plane = AS_BM * KP;

  const float* A = G.A + (int64_t)bz * G.a_bs;
  const uint16_t* Bt = G.Bt + (int64_t)bz * 2 * G.N * K;
  float* C = G.C + (int64_t)bz * G.c_bs;
  const float* bias = G.bias ? G.bias + (int64_t)bz * G.N : nullptr;
  const float* init = G.init ? G.init + (int64_t)bz * G.i_bs : nullptr;
  const float sA = f16_scale(*G.a_amax);
  const float inv = 1.0f / (sA * *G.b_scale);
  const float xw = inv * (1.0f / 2048.f);

  // ---- A tile: fp32 -> scaled hi / lo planes in LDS, 8 float4 loads in flight per thread
  const int n_f4 = AS_BM * K4;
  for (int base = 0; base < n_f4; base += 512 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 512 + tid;
      const int r = idx / K4, c4 = idx - r * K4;
      v[u] = (idx < n_f4 && m0 + r < G.M) ? ld4(A + (int64_t)(m0 + r) * G.lda + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 512 + tid;
      if (idx < n_f4) {
        const int r = idx / K4, c4 = idx - r * K4;
        u32x2 h, l;
        split4h(v[u] * sA, h, l);
        *reinterpret_cast<u32x2*>(&lds_as[r * KP + c4 * 4]) = h;
        *reinterpret_cast<u32x2*>(&lds_as[plane + r * KP + c4 * 4]) = l;
      }
    }
  }
  __syncthreads();

  const int KT = K / GBK;
  const int a_off = (lane & 15) * KP + (lane >> 4) * 8;            // + i*16*KP + t*32 (+ plane)
  const int64_t b_off = (int64_t)lane * 8;                            // fragment-major planes: + ((n/16 + j)*KT + t)*512 (+ plane N*K)
  const int64_t b_plane = (int64_t)G.N * K;

  // Blocks walk the column steps in rotated order (block b starts at step b mod n_steps): every block streams the SAME
  // weight planes from its XCD's L2, and blocks that start together would otherwise ask for the same lines at the same
  // time, all on one L2 channel
  const int n_steps = (G.N + 255) / 256;
  const int rot = blockIdx.x % n_steps;
  // The first k-step's weight fragments of a column step are requested BEFORE the previous step's results are stored:
  // gfx9's vmcnt counts loads and stores in issue order, so a fragment load issued after the 16 stores could only be
  // waited for by waiting for those stores to be acknowledged by memory - every step began with a full store drain.
  f16x8 bh_n[2], bl_n[2];
  auto first_frags = [&](int it_) {
    const int n_w_ = ((it_ + rot) % n_steps) * 256 + wave * 32;
    if (it_ < n_steps && n_w_ < G.N) {
      const uint16_t* bp_ = Bt + (int64_t)(n_w_ >> 4) * KT * 512 + b_off;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh_n[j] = *reinterpret_cast<const f16x8*>(bp_ + (int64_t)j * KT * 512);
        bl_n[j] = *reinterpret_cast<const f16x8*>(bp_ + b_plane + (int64_t)j * KT * 512);
      }
    }
  };
  first_frags(0);
  for (int it = 0; it < n_steps; ++it) {
    const int ns = (it + rot) % n_steps;
    const int n_w = ns * 256 + wave * 32;
    if (n_w >= G.N) {                                               // N % 256 == 128: waves 4-7 idle in the last step
      first_frags(it + 1);
      continue;
    }
    // bias and the additive init matrix seed the hi*hi accumulator (times s_A s_B, a power of two: exact) instead
    // of being added per element in the epilogue, where 64 dependent loads per lane sat between the stores
    f32x4v acc[8][2], acx[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        acx[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
    if (bias || init) {
      const float seed = sA * *G.b_scale;
      // transposed accumulators (see the MFMA calls): acc[i][j][r] = C[m0 + 16i + (lane & 15)][n_w + 16j + 4*(lane >> 4) + r]
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cj = n_w + 16 * j + 4 * (lane >> 4);
        const f32x4 bv = bias ? ld4(bias + cj) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = m0 + 16 * i + (lane & 15);
          const f32x4 iv = (init && row < G.M) ? ld4(init + (int64_t)row * G.ldi + cj) : f32x4{0.f, 0.f, 0.f, 0.f};
          acc[i][j] = f32x4v{(bv.x + iv.x) * seed, (bv.y + iv.y) * seed, (bv.z + iv.z) * seed, (bv.w + iv.w) * seed};
        }
      }
    }
    const uint16_t* bp = Bt + (int64_t)(n_w >> 4) * KT * 512 + b_off;
    for (int t = 0; t < KT; ++t) {
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = bh_n[j];
        bl[j] = bl_n[j];
      }
      if (t + 1 < KT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh_n[j] = *reinterpret_cast<const f16x8*>(bp + (int64_t)(j * KT + t + 1) * 512);
          bl_n[j] = *reinterpret_cast<const f16x8*>(bp + b_plane + (int64_t)(j * KT + t + 1) * 512);
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {          // 4 row tiles at a time
        // Hand-ordered (sched_barriers pin it): left to itself the compiler sinks every ds_read next to its first
        // use and waits for it with nothing else in flight.  The lo fragments are fetched while the hi MFMAs run.
        auto lda = [&](int pl, int i) {
          return *reinterpret_cast<const f16x8*>(&lds_as[pl * plane + a_off + (half * 4 + i) * 16 * KP + t * GBK]);
        };
        // Operand order (weight fragment, A fragment): the product is accumulated TRANSPOSED, so a lane ends up with 4
        // consecutive columns of one output row (C^T tile: register r = column 4*(lane>>4)+r, lane&15 = row) and the
        // epilogue is one 16-byte store per tile - 16 store instructions per column step and wave instead of 64.
#define DISGAT_MF(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(b_, a_, c_, 0, 0, 0)
        auto hi2 = [&](const f16x8& a, int i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[half * 4 + i][j] = DISGAT_MF(a, bh[j], acc[half * 4 + i][j]);
            acx[half * 4 + i][j] = DISGAT_MF(a, bl[j], acx[half * 4 + i][j]);
          }
        };
        auto lo2 = [&](const f16x8& a, int i) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acx[half * 4 + i][j] = DISGAT_MF(a, bh[j], acx[half * 4 + i][j]);
        };
#undef DISGAT_MF
        const f16x8 ah0 = lda(0, 0), ah1 = lda(0, 1), ah2 = lda(0, 2), ah3 = lda(0, 3);
        __builtin_amdgcn_sched_barrier(0);
        hi2(ah0, 0);
        hi2(ah1, 1);
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 al0 = lda(1, 0), al1 = lda(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        hi2(ah2, 2);
        hi2(ah3, 3);
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 al2 = lda(1, 2), al3 = lda(1, 3);
        __builtin_amdgcn_sched_barrier(0);
        lo2(al0, 0);
        lo2(al1, 1);
        lo2(al2, 2);
        lo2(al3, 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // Epilogue cost matters here: per N-step a wave retires 384 MFMAs and 128 outputs per lane.  Addresses are two
    // per-lane pointers (rows +0 and +8 of each 16-row tile) advanced by the uniform row stride - one 64-bit add per
    // store, no multiplies; the activation is a template parameter (a run-time switch evaluated every branch for every
    // element); ragged last tiles take the checked copy of the loop.
    first_frags(it + 1);                       // next step's first fragments: in flight before the stores below
    const bool full = m0 + AS_BM <= G.M;
    int64_t ldc = G.ldc;
    asm volatile("" : "+s"(ldc));
    // lane (lane & 15, lane >> 4) stores C[m0 + 16i + (lane & 15)][n_w + 16j + 4*(lane >> 4) .. +3]: 16 B per lane, one
    // 64-lane instruction = 16 rows x 64 B; the pointer advances by 16 rows per tile (one 64-bit add, no multiplies)
    float* pr = C + (int64_t)(m0 + (lane & 15)) * ldc + n_w + 4 * (lane >> 4);
    auto emit = [&](auto checked) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + 16 * i + (lane & 15);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 v;
          v.x = act_ct<ACT>(fmaf(acx[i][j][0], xw, acc[i][j][0] * inv), G.slope);
          v.y = act_ct<ACT>(fmaf(acx[i][j][1], xw, acc[i][j][1] * inv), G.slope);
          v.z = act_ct<ACT>(fmaf(acx[i][j][2], xw, acc[i][j][2] * inv), G.slope);
          v.w = act_ct<ACT>(fmaf(acx[i][j][3], xw, acc[i][j][3] * inv), G.slope);
          if (!decltype(checked)::value || row < G.M) st4(pr + 16 * j, v);
        }
        pr += 16 * ldc;
      }
    };
    if (full) emit(std::false_type{});
    else emit(std::true_type{});
  }
}

// Weight gradient dW[Ka][N] = A^T G with A [M][Ka], G [M][N] both fp32 and both "k-major" (the reduction index M
// is the ROW of either operand), M ~ 1e6: split-K over row ranges, deterministic partial sums reduced by the host.
// Both operands are split on the fly (same two-plane fp16 scheme, scales from their max magnitudes) into LDS images
// [32 k-rows][128 cols] and consumed through gfx950's transposing LDS read `ds_read_b64_tr_b16`: per 16-lane group
// it takes a 4 x 16 block and hands lane i the 4 k-values of column i - two of them are one 16x16x32 operand
// fragment.  256-byte image rows with the XOR chunk swizzle of the CDNA4 guide (T10, image (b)) keep both the 8-byte
// stores and the transposed reads conflict-free.
typedef __fp16 trh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct GemmTNArgs {
  const float* A;
  int64_t lda, a_bs;
  const float* G;
  int64_t ldg, g_bs;
  const float* a_amax;
  const float* g_amax;
  float* part;            // [batch][S][Ka][N]
  int M, Ka, N, batch, S, rows_per_split;
};

__device__ __forceinline__ int tn_off(int row, int ch) {        // byte offset of 16-byte chunk ch of image row `row`
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

__global__ __launch_bounds__(256, 2) void gemm_f16x3_tn_kernel(const GemmTNArgs T) {
  __shared__ __attribute__((aligned(16))) unsigned char img[4 * 8192];     // A hi, A lo, G hi, G lo: 32 rows x 256 B
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = T.N / 128;
  // XCD-aware block -> (tile, split) map.  All tiles of one split read the same row range of both operands; workgroups go
  // to the 8 XCDs round-robin in launch order, so with the plain map the 32 tiles of a split sat on 8 different L2s and
  // each fetched its own copy: 23.7 GB through the fabric per launch for 9 GB of operands.  Here XCD c takes whole splits
  // c, c + 8, ...: 13.6 GB (rocprofv3 --pmc FETCH_SIZE, tools/wgrad_bench.py).  The time moves by 2-5 % only - the kernel
  // is bound by its own split + LDS + MFMA phases, not by the fabric - but half the traffic is half the traffic.
  int tile = blockIdx.x, sp = blockIdx.y;
  {
    const int nx = gridDim.x, S = gridDim.y;
    if ((S & 7) == 0) {
      const int L = blockIdx.x + nx * blockIdx.y;      // launch order within this batch slab (nx * S is a multiple of 8)
      const int q = L >> 3;
      tile = q % nx;
      sp = (q / nx) * 8 + (L & 7);
    }
  }
  const int ka0 = (tile / nt) * 128, n0 = (tile % nt) * 128;
  const int bz = blockIdx.z;
  const int k_begin = sp * T.rows_per_split;
  const int k_end = min(T.M, k_begin + T.rows_per_split);

  const float* A = T.A + (int64_t)bz * T.a_bs + ka0;
  const float* G = T.G + (int64_t)bz * T.g_bs + n0;
  const float sA = f16_scale(*T.a_amax), sG = f16_scale(*T.g_amax);
  const float inv = 1.0f / (sA * sG);

  // staging: thread t covers rows (t >> 5) + 8 i (i < 4), columns 4 (t & 31) .. +3 of both 32 x 128 tiles
  const int s_row = tid >> 5, s_col = (tid & 31) * 4;
  f32x4 a_s[4], g_s[4];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + s_row + 8 * i;
      const bool ok = k < k_end;
      a_s[i] = ok ? ld4(A + (int64_t)k * T.lda + s_col) : f32x4{0.f, 0.f, 0.f, 0.f};
      g_s[i] = ok ? ld4(G + (int64_t)k * T.ldg + s_col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = tn_off(s_row + 8 * i, s_col >> 3) + (s_col & 4) * 2;
      u32x2 h, l;
      split4h(a_s[i] * sA, h, l);
      *reinterpret_cast<u32x2*>(img + o) = h;
      *reinterpret_cast<u32x2*>(img + 8192 + o) = l;
      split4h(g_s[i] * sG, h, l);
      *reinterpret_cast<u32x2*>(img + 16384 + o) = h;
      *reinterpret_cast<u32x2*>(img + 24576 + o) = l;
    }
  };

  // transposed fragment read: lane (group g, q, p) addresses row r0 + q, columns c0 + 4p .. +3; lane i of the group
  // receives column c0 + i of rows r0 .. r0+3.  Two reads (r0 = 8g, 8g + 4) = the 8 k-values of a 16x16x32 operand.
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  auto frag = [&](int image, int ch0) {
    typedef __attribute__((address_space(3))) trh4* lptr;
    const int o0 = image * 8192 + tn_off(8 * g4 + q4, ch0 + (p4 >> 1)) + 8 * (p4 & 1);
    const int o1 = image * 8192 + tn_off(8 * g4 + 4 + q4, ch0 + (p4 >> 1)) + 8 * (p4 & 1);
    const trh4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(img + o0));
    const trh4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(img + o1));
    f16x8 r;
    r[0] = (_Float16)lo4[0]; r[1] = (_Float16)lo4[1]; r[2] = (_Float16)lo4[2]; r[3] = (_Float16)lo4[3];
    r[4] = (_Float16)hi4[0]; r[5] = (_Float16)hi4[1]; r[6] = (_Float16)hi4[2]; r[7] = (_Float16)hi4[3];
    return r;
  };

  f32x4v acc[4][4], acx[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      acx[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }

  auto compute = [&]() {
    f16x8 ah[4], gh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = frag(0, 8 * wm + 2 * i);
      gh[i] = frag(2, 8 * wn + 2 * i);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], gh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x8 gl = frag(3, 8 * wn + 2 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], gl, acx[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f16x8 al = frag(1, 8 * wm + 2 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, gh[j], acx[i][j], 0, 0, 0);
    }
  };

  if (k_begin < k_end) {
    load(k_begin);
    store();
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
      const bool more = k0 + 32 < k_end;
      if (more) load(k0 + 32);
      compute();
      __syncthreads();
      if (more) {
        store();
        __syncthreads();
      }
    }
  }

  float* P = T.part + (((int64_t)bz * T.S + sp) * T.Ka + ka0 + 64 * wm) * T.N + n0 + 64 * wn;
  const float xw = inv * (1.0f / 2048.f);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        P[(int64_t)(16 * i + 4 * (lane >> 4) + r) * T.N + 16 * j + (lane & 15)] = fmaf(acx[i][j][r], xw, acc[i][j][r] * inv);
}

// max over the block of a per-thread bit pattern (non-negative floats order like unsigned integers), then ONE atomicMax per
// block - and only when the block's value can still raise *out.  One atomic per WAVE, as before, serialises at the L2: the
// first few thousand waves of a launch all start against *out == 0 and queue up on the same address (an amax over 21 MB took
// 70 us, 0.64 ms over 1 GB).  Call with all 256 threads.
__device__ __forceinline__ void block_atomic_max(uint32_t m, uint32_t* __restrict__ out) {
  __shared__ uint32_t wave_max[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
  }
}

// Weight preparation for the f16x3 scheme in two small launches (the torch formulation took ~15): max |W| over an
// arbitrarily strided [batch][K][N] weight, then hi / lo planes of (W^T * s) as [batch][2][N][K] fp16 and s itself.
__global__ __launch_bounds__(256) void wamax_kernel(const float* __restrict__ W, int64_t sb, int64_t sk, int64_t sn, int K,
                                                    int N, int64_t total, uint32_t* __restrict__ out) {
  uint32_t m = 0u;
  const int64_t kn = (int64_t)K * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / kn, r = i - b * kn;
    const int64_t k = r / N, n = r - k * N;
    m = max(m, __float_as_uint(fabsf(W[b * sb + k * sk + n * sn])));
  }
  block_atomic_max(m, out);
}

// One element of the split: position i of the [batch][N][K] output order (coalesced plane writes), scale s.
// I: the integer type of the index arithmetic (the one-launch form of small weights divides in 32 bits).
template <typename I>
__device__ __forceinline__ void wsplit_one(const float* __restrict__ W, int64_t sb, int64_t sk, int64_t sn, int K, int N,
                                           I kn, I i, float s, uint16_t* __restrict__ planes, int frag) {
  const I b = i / kn, r = i - b * kn;
  const I n = r / (I)K, k = r - n * (I)K;
  const float t = W[b * sb + k * sk + n * sn] * s;
  const _Float16 h = (_Float16)t;
  const _Float16 l = (_Float16)((t - (float)h) * 2048.f);
  // K <= 256 (operands of the A-stationary kernel): fragment-major - the 16 x 32 block of (n-tile, k-step) is stored
  // in MFMA lane order (lane = 16*(k%32/8) + n%16, 8 halfs each), so a wave's fragment load is one contiguous KB
  int64_t o = r;
  if (frag == 1) {
    o = ((((n >> 4) * (K >> 5) + (k >> 5)) * 64 + (((k & 31) >> 3) << 4) + (n & 15)) << 3) + (k & 7);
  } else if (frag == 2) {
    // DMA-tiled (disgat_gemm_planes): per (256-column step, k-step) one 16 KB block that IS the LDS image of the weight
    // tile - LDS row rho holds weight row wperm(rho) (gemm_planes.hip), 16-byte chunk c of it at slot c ^ ((rho >> 1) & 3)
    // - so a DMA piece (16 LDS rows) is 1 KB of consecutive memory: 8 full 128-byte lines instead of 16 half lines
    const int64_t ns = n >> 8, nl = n & 255, x = nl & 31;
    const int64_t rho = (nl & ~31) + (((x >> 2) & 1) << 4) + ((x >> 3) << 2) + (x & 3);
    const int64_t t2 = k >> 5, c = ((k & 31) >> 3) ^ ((rho >> 1) & 3);
    o = (((ns * (K >> 5) + t2) * 256 + rho) << 5) + (c << 3) + (k & 7);
  }
  planes[(b * 2 + 0) * kn + o] = *reinterpret_cast<const uint16_t*>(&h);
  planes[(b * 2 + 1) * kn + o] = *reinterpret_cast<const uint16_t*>(&l);
}

__global__ __launch_bounds__(256) void wsplit_kernel(const float* __restrict__ W, int64_t sb, int64_t sk, int64_t sn, int K,
                                                     int N, int64_t total, const float* __restrict__ amax,
                                                     uint16_t* __restrict__ planes, float* __restrict__ scale_out, int frag) {
  const float s = f16_scale(*amax);
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
  const int64_t kn = (int64_t)K * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    wsplit_one<int64_t>(W, sb, sk, sn, K, N, kn, i, s, planes, frag);
}

// The same preparation of a SMALL weight (<= WPREP_SMALL elements: the layers of the bundled graphs, whose train_steps
// are launch-bound) in ONE launch: every block reduces max |W| over the whole weight itself (<= 512 KB, from L2), so
// there is no zeroed accumulator, no atomic and no second kernel; then it splits its own share.  Same max, same scale,
// same planes as the two-launch form, bit for bit.
constexpr int64_t WPREP_SMALL = 131072;
constexpr int WPREP_THREADS = 1024, WPREP_PER_BLOCK = 4096;
__global__ __launch_bounds__(WPREP_THREADS) void wprep_small_kernel(const float* __restrict__ W, int64_t sb, int64_t sk, int64_t sn,
                                                                    int K, int N, int total, int dense, float* __restrict__ amax_scale,
                                                                    uint16_t* __restrict__ planes, int frag) {
  __shared__ uint32_t wave_max[WPREP_THREADS / 64];
  uint32_t m = 0u;
  const int kn = K * N;
  auto upd = [&](float v) { m = max(m, __float_as_uint(fabsf(v))); };
  if (dense > 0 && (dense & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
    // the view covers `dense` consecutive floats from W (any order of its dimensions; broadcast dimensions counted once):
    // the maximum does not care about the order.  8 independent 16-byte loads per thread and batch - the loop is latency,
    // not bandwidth (one batch covers 32 768 floats)
    const int n4 = dense >> 2;
    for (int base = threadIdx.x; base < n4; base += 8 * WPREP_THREADS) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * WPREP_THREADS;
        v[j] = idx < n4 ? ld4(W + (int64_t)idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { upd(v[j].x); upd(v[j].y); upd(v[j].z); upd(v[j].w); }
    }
  } else if (dense > 0) {
    for (int base = threadIdx.x; base < dense; base += 8 * WPREP_THREADS) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * WPREP_THREADS;
        v[j] = idx < dense ? W[idx] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) upd(v[j]);
    }
  } else {
    const int scan = sb == 0 ? kn : total;          // a weight broadcast over the batch (one MLP for every head): one copy
    for (int base = threadIdx.x; base < scan; base += 4 * WPREP_THREADS) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = base + j * WPREP_THREADS;
        const int bb = i / kn, r = i - bb * kn;
        const int k = r / N, n = r - k * N;
        v[j] = i < scan ? W[bb * sb + k * sk + n * sn] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) upd(v[j]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  m = 0u;
#pragma unroll
  for (int w = 0; w < WPREP_THREADS / 64; ++w) m = max(m, wave_max[w]);
  const float s = f16_scale(__uint_as_float(m));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    amax_scale[0] = __uint_as_float(m);
    amax_scale[1] = s;
  }
  const int lo = blockIdx.x * WPREP_PER_BLOCK, hi = min(total, lo + WPREP_PER_BLOCK);
#pragma unroll 4
  for (int i = lo + threadIdx.x; i < hi; i += WPREP_THREADS) wsplit_one<int>(W, sb, sk, sn, K, N, kn, i, s, planes, frag);
}

// Bound on |x W| for the f16x3 scale of a GEMM output: max over (batch, column) of sum_k |W[b][k][n]| - the largest column
// abs-sum - times the bound on |x| and a safety factor, floored (|elu(v)| <= max(|v|, 1)).  The ATen chain (abs, sum, max, mul,
// mul, clamp) was six launches per layer of a launch-bound step; here one block: a thread per column, 8 rows in flight.
// out[0] = the largest column abs-sum; out[1] = max(floor, in_bound * out[0] * scale) (in_bound == NULL: not written).
constexpr int WBOUND_THREADS = 1024;
__global__ __launch_bounds__(WBOUND_THREADS) void weight_bound_kernel(const float* __restrict__ W, int64_t sb, int64_t sk, int64_t sn,
                                                                      int K, int N, int batch, const float* __restrict__ in_bound,
                                                                      float scale, float floor_, float* __restrict__ out) {
  __shared__ float wave_max[WBOUND_THREADS / 64];
  float best = 0.f;
  const int cols = batch * N;
  for (int c = threadIdx.x; c < cols; c += WBOUND_THREADS) {
    const int b = c / N, n = c - b * N;
    const float* p = W + b * sb + n * sn;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(k + j) * sk];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += fabsf(v[j]);
    }
    for (; k < K; ++k) s += fabsf(p[(int64_t)k * sk]);
    best = fmaxf(best, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int w = 0; w < WBOUND_THREADS / 64; ++w) m = fmaxf(m, wave_max[w]);
    out[0] = m;
    if (in_bound != nullptr) out[1] = fmaxf(floor_, in_bound[0] * m * scale);
  }
}

// max |A| over a (batched, strided) fp32 operand into *out (a device float the caller zeroed): the bit pattern
// of a non-negative float orders like an unsigned integer, so the reduction is one atomicMax per wave.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ A, int64_t lda, int64_t a_bs, int64_t rows,
                                                   int M, int K4, uint32_t* __restrict__ out) {
  uint32_t m = 0u;
  const int64_t total = rows * K4;
  auto upd = [&](const f32x4 v) {
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  };
  if (lda == (int64_t)K4 * 4 && (rows == M || a_bs == (int64_t)M * lda)) {      // dense: one flat sweep, 4 loads in flight
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {
      const f32x4 v0 = ld4(A + i * 4), v1 = ld4(A + (i + stride) * 4), v2 = ld4(A + (i + 2 * stride) * 4), v3 = ld4(A + (i + 3 * stride) * 4);
      upd(v0); upd(v1); upd(v2); upd(v3);
    }
    for (; i < total; i += stride) upd(ld4(A + i * 4));
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / K4;
      const int c = (int)(i - r * K4) * 4;
      const int64_t bz = r / M, rm = r - bz * M;
      upd(ld4(A + bz * a_bs + rm * lda + c));
    }
  }
  block_atomic_max(m, out);
}

}  // namespace disgat

// ------------------------------------------------------------------------------------------
#include <utility>
#include "disgat_api.h"

namespace disgat {
// Backward of the fused epilogue activation from the saved OUTPUT: gin = g * act'(pre-activation), with
// ELU' = (out > 0 ? 1 : out + 1) (out + 1 = exp(v) for v <= 0) and leaky' = (out > 0 ? 1 : slope).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                      float* __restrict__ gin, int64_t n4, int act, float slope,
                                                      uint32_t* __restrict__ amax_out) {
  uint32_t m = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 gv = ld4(g + i * 4), o = ld4(out + i * 4);
    f32x4 r;
    r.x = gv.x * (o.x > 0.f ? 1.f : (act == 1 ? o.x + 1.f : slope));
    r.y = gv.y * (o.y > 0.f ? 1.f : (act == 1 ? o.y + 1.f : slope));
    r.z = gv.z * (o.z > 0.f ? 1.f : (act == 1 ? o.z + 1.f : slope));
    r.w = gv.w * (o.w > 0.f ? 1.f : (act == 1 ? o.w + 1.f : slope));
    st4(gin + i * 4, r);
    m = max(max(m, __float_as_uint(fabsf(r.x))), max(__float_as_uint(fabsf(r.y)), max(__float_as_uint(fabsf(r.z)), __float_as_uint(fabsf(r.w)))));
  }
  if (amax_out != nullptr) block_atomic_max(m, amax_out);   // max |gin| for the GEMMs that consume it (saves them a pass over gin)
}
}  // namespace disgat

extern "C" int disgat_act_bwd(const float* g, const float* out, float* gin, int64_t n, int act, float slope,
                              float* amax_out, disgat_stream_t stream) {
  using namespace disgat;
  if (amax_out != nullptr) {
    const hipError_t e = hipMemsetAsync(amax_out, 0, sizeof(float), reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail((int)e, "act_bwd: memset failed: %s", hipGetErrorString(e));
  }
  if (n == 0) return 0;
  DISGAT_REQUIRE(g && out && gin && n > 0 && n % 4 == 0, "act_bwd: null pointer or n %% 4 != 0");
  DISGAT_REQUIRE(act == 1 || act == 2, "act_bwd: act=%d (1 = ELU, 2 = leaky ReLU)", act);
  DISGAT_REQUIRE(aligned16(g) && aligned16(out) && aligned16(gin), "act_bwd: buffers must be 16-byte aligned");
  const int64_t n4 = n / 4;
  const int64_t want = n4 / (256 * 4) + 1;             // >= 4 float4 per thread, at most 4096 blocks
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, out, gin, n4,
                     act, slope, reinterpret_cast<uint32_t*>(amax_out));
  return check_launch("act_bwd_kernel");
}

extern "C" int disgat_gemm_split(const float* A, int64_t lda, int64_t a_batch_stride, const uint16_t* Bt_planes,
                                 const float* bias, const float* init, int64_t ldi, int64_t init_batch_stride, float* C,
                                 int64_t ldc, int64_t c_batch_stride, int M, int N, int K, int batch, int act,
                                 float slope, int terms, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0) return 0;
  DISGAT_REQUIRE(A && Bt_planes && C && M > 0 && batch > 0, "gemm_split: null pointer / bad sizes");
  DISGAT_REQUIRE(N > 0 && N % GBN == 0 && K > 0 && K % GBK == 0, "gemm_split: N=%d must be a multiple of %d and K=%d of %d", N, GBN, K, GBK);
  DISGAT_REQUIRE(lda % 4 == 0 && a_batch_stride % 4 == 0 && aligned16(A) && aligned16(Bt_planes),
                 "gemm_split: A rows must be 16-byte aligned (lda, batch stride multiples of 4)");
  DISGAT_REQUIRE(terms == 3 || terms == 6, "gemm_split: terms must be 3 or 6");
  DISGAT_REQUIRE(act >= 0 && act <= 2, "gemm_split: act must be 0 (none), 1 (elu) or 2 (leaky relu)");
  GemmArgs G{A, lda, a_batch_stride, Bt_planes, bias, init, ldi, init_batch_stride, C, ldc, c_batch_stride,
             M, N, K, batch, act, slope, terms, (M + GBM - 1) / GBM, N / GBN};
  const int64_t blocks = (int64_t)((G.mt + 7) / 8) * 8 * G.nt;
  DISGAT_REQUIRE(blocks < ((int64_t)1 << 31) && batch < 65536, "gemm_split: grid too large");
  if (terms == 6)
    hipLaunchKernelGGL(gemm_split_kernel<6>, dim3((unsigned)blocks, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G);
  else
    hipLaunchKernelGGL(gemm_split_kernel<3>, dim3((unsigned)blocks, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G);
  return check_launch("gemm_split_kernel");
}

namespace disgat {
static int split_f16_impl(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                          uint16_t* planes, float* amax_scale, int frag, hipStream_t st) {
  if ((int64_t)batch * K * N <= WPREP_SMALL) {
    const int64_t tot = (int64_t)batch * K * N;
    // does the view cover one run of consecutive floats starting at W?  (dimensions in stride order, each stride = the
    // extent below it; a broadcast dimension - stride 0 - or one of size 1 adds nothing)
    int64_t dims[3][2] = {{stride_b, batch}, {stride_k, K}, {stride_n, N}};
    for (int i = 0; i < 3; ++i)
      for (int j = i + 1; j < 3; ++j)
        if (dims[j][0] < dims[i][0]) { std::swap(dims[i][0], dims[j][0]); std::swap(dims[i][1], dims[j][1]); }
    int64_t extent = 1;
    bool dense = true;
    for (int i = 0; i < 3; ++i) {
      if (dims[i][1] == 1 || dims[i][0] == 0) continue;
      if (dims[i][0] != extent) { dense = false; break; }
      extent *= dims[i][1];
    }
    hipLaunchKernelGGL(wprep_small_kernel, dim3((unsigned)((tot + WPREP_PER_BLOCK - 1) / WPREP_PER_BLOCK)), dim3(WPREP_THREADS), 0, st,
                       W, stride_b, stride_k, stride_n, K, N, (int)tot, dense ? (int)extent : 0, amax_scale, planes, frag);
    return check_launch("wprep_small_kernel");
  }
  const hipError_t e = hipMemsetAsync(amax_scale, 0, 2 * sizeof(float), st);
  if (e != hipSuccess) return fail((int)e, "split_f16: memset failed: %s", hipGetErrorString(e));
  const int64_t total = (int64_t)batch * K * N;
  const int grid = (int)(total / 256 + 1 < 1024 ? total / 256 + 1 : 1024);
  hipLaunchKernelGGL(wamax_kernel, dim3(grid), dim3(256), 0, st, W, stride_b, stride_k, stride_n, K, N, total,
                     reinterpret_cast<uint32_t*>(amax_scale));
  hipLaunchKernelGGL(wsplit_kernel, dim3(grid), dim3(256), 0, st, W, stride_b, stride_k, stride_n, K, N, total, amax_scale,
                     planes, amax_scale + 1, frag);
  return check_launch("wsplit_kernel");
}
}  // namespace disgat

extern "C" int disgat_weight_bound(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                                   const float* in_bound, float scale, float floor_value, float* out, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(W && out && K > 0 && N > 0 && batch > 0, "weight_bound: null pointer / bad sizes");
  DISGAT_REQUIRE((int64_t)batch * K * N <= WPREP_SMALL, "weight_bound: one block serves weights of up to %lld elements", (long long)WPREP_SMALL);
  hipLaunchKernelGGL(weight_bound_kernel, dim3(1), dim3(WBOUND_THREADS), 0, reinterpret_cast<hipStream_t>(stream), W, stride_b, stride_k,
                     stride_n, K, N, batch, in_bound, scale, floor_value, out);
  return check_launch("weight_bound_kernel");
}

extern "C" int disgat_split_f16(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                                uint16_t* planes, float* amax_scale, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(W && planes && amax_scale && K > 0 && N > 0 && batch > 0, "split_f16: null pointer / bad sizes");
  // layout rule shared with disgat_gemm_f16x3: K <= 256 selects its A-stationary kernel, which reads fragment-major planes
  const int frag = (K <= 256 && K % 32 == 0 && N % 16 == 0) ? 1 : 0;
  return split_f16_impl(W, stride_b, stride_k, stride_n, K, N, batch, planes, amax_scale, frag, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int disgat_split_f16_rm(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                                   uint16_t* planes, float* amax_scale, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(W && planes && amax_scale && K > 0 && N > 0 && batch > 0, "split_f16_rm: null pointer / bad sizes");
  DISGAT_REQUIRE(N % 256 == 0 && K % 32 == 0, "split_f16_rm: N=%d must be a multiple of 256 and K=%d of 32", N, K);
  return split_f16_impl(W, stride_b, stride_k, stride_n, K, N, batch, planes, amax_scale, 2, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int disgat_amax(const float* A, int64_t lda, int64_t a_batch_stride, int M, int K, int batch, float* out,
                           disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(out != nullptr, "amax: null output");
  {
    const hipError_t e = hipMemsetAsync(out, 0, sizeof(float), reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail((int)e, "amax: memset failed: %s", hipGetErrorString(e));
  }
  if (M == 0 || batch == 0 || K == 0) return 0;
  DISGAT_REQUIRE(A && out && M > 0 && K > 0 && batch > 0, "amax: null pointer / bad sizes");
  DISGAT_REQUIRE(K % 4 == 0 && lda % 4 == 0 && a_batch_stride % 4 == 0 && aligned16(A), "amax: K, lda and the batch stride must be multiples of 4, A 16-byte aligned");
  const int64_t rows = (int64_t)M * batch;
  const int64_t work = rows * (K / 4);
  const int64_t want = work / (256 * 8) + 1;           // >= 8 float4 per thread, at most 2048 blocks (8 per CU)
  const int grid = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(amax_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A, lda, a_batch_stride,
                     rows, M, K / 4, reinterpret_cast<uint32_t*>(out));
  return check_launch("amax_kernel");
}

extern "C" int disgat_gemm_f16x3(const float* A, int64_t lda, int64_t a_batch_stride, const uint16_t* Bt_planes,
                                 const float* a_amax, const float* b_scale, const float* bias, const float* init,
                                 int64_t ldi, int64_t init_batch_stride, float* C, int64_t ldc, int64_t c_batch_stride,
                                 int M, int N, int K, int batch, int act, float slope, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0) return 0;
  DISGAT_REQUIRE(A && Bt_planes && C && a_amax && b_scale && M > 0 && batch > 0, "gemm_f16x3: null pointer / bad sizes");
  DISGAT_REQUIRE(N > 0 && K > 0 && K % GBK == 0 && (N % GBN == 0 || gemm_rs_takes(N, K)),
                 "gemm_f16x3: N=%d must be a multiple of %d (of 32 for K = 64 / 128 / 256) and K=%d of %d", N, GBN, K, GBK);
  DISGAT_REQUIRE(lda % 4 == 0 && a_batch_stride % 4 == 0 && aligned16(A) && aligned16(Bt_planes),
                 "gemm_f16x3: A rows must be 16-byte aligned (lda, batch stride multiples of 4)");
  DISGAT_REQUIRE(act >= 0 && act <= 2, "gemm_f16x3: act must be 0 (none), 1 (elu) or 2 (leaky relu)");
  GemmHArgs G{A, lda, a_batch_stride, Bt_planes, a_amax, b_scale, bias, init, ldi, init_batch_stride, C, ldc,
              c_batch_stride, M, N, K, batch, act, slope, (M + GBM - 1) / GBM, N / GBN};
  const int64_t blocks = (int64_t)((G.mt + 7) / 8) * 8 * G.nt;
  DISGAT_REQUIRE(blocks < ((int64_t)1 << 31) && batch < 65536, "gemm_f16x3: grid too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // K = 64 / 128 / 256: A register-stationary, weights through an LDS ring (gemm_rs.hip); DISGAT_GEMM_AS=1 keeps the
  // A-in-LDS kernel below for those shapes too (same-box A/B)
  static const bool force_as = getenv("DISGAT_GEMM_AS") && atoi(getenv("DISGAT_GEMM_AS")) != 0;
  if ((!force_as || N % GBN != 0) && gemm_rs_takes(N, K)) return launch_gemm_f16x3_rs(G, st);
  if (K <= 256) {                       // A-stationary: the whole 128 x K A tile lives in LDS (dynamic, > 64 KB)
    const int lds_bytes = 2 * AS_BM * (K + AS_PAD) * (int)sizeof(uint16_t);
    static int lds_set = 0;
    if (lds_set < lds_bytes) {
      for (const void* f : {reinterpret_cast<const void*>(gemm_f16x3_as_kernel<0>), reinterpret_cast<const void*>(gemm_f16x3_as_kernel<1>),
                            reinterpret_cast<const void*>(gemm_f16x3_as_kernel<2>)}) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail((int)e, "gemm_f16x3: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(e));
      }
      lds_set = lds_bytes;
    }
    const dim3 grid((unsigned)G.mt, batch);
    if (act == 1) hipLaunchKernelGGL(gemm_f16x3_as_kernel<1>, grid, dim3(512), lds_bytes, st, G);
    else if (act == 2) hipLaunchKernelGGL(gemm_f16x3_as_kernel<2>, grid, dim3(512), lds_bytes, st, G);
    else hipLaunchKernelGGL(gemm_f16x3_as_kernel<0>, grid, dim3(512), lds_bytes, st, G);
    return check_launch("gemm_f16x3_as_kernel");
  }
  hipLaunchKernelGGL(gemm_f16x3_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, st, G);
  return check_launch("gemm_f16x3_kernel");
}

extern "C" int disgat_gemm_f16x3_tn(const float* A, int64_t lda, int64_t a_batch_stride, const float* G, int64_t ldg,
                                    int64_t g_batch_stride, const float* a_amax, const float* g_amax, float* partials,
                                    int M, int Ka, int N, int batch, int splits, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0) return 0;
  DISGAT_REQUIRE(A && G && a_amax && g_amax && partials && M > 0 && batch > 0 && splits > 0, "gemm_f16x3_tn: null pointer / bad sizes");
  DISGAT_REQUIRE(Ka > 0 && Ka % 128 == 0 && N > 0 && N % 128 == 0, "gemm_f16x3_tn: Ka=%d and N=%d must be multiples of 128", Ka, N);
  DISGAT_REQUIRE(lda % 4 == 0 && ldg % 4 == 0 && a_batch_stride % 4 == 0 && g_batch_stride % 4 == 0 && aligned16(A) && aligned16(G),
                 "gemm_f16x3_tn: operand rows must be 16-byte aligned");
  const int rows = ((M + splits - 1) / splits + 31) / 32 * 32;
  GemmTNArgs T{A, lda, a_batch_stride, G, ldg, g_batch_stride, a_amax, g_amax, partials, M, Ka, N, batch, splits, rows};
  DISGAT_REQUIRE(splits < 65536 && batch < 65536, "gemm_f16x3_tn: grid too large");
  hipLaunchKernelGGL(gemm_f16x3_tn_kernel, dim3((unsigned)((Ka / 128) * (N / 128)), splits, batch), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), T);
  return check_launch("gemm_f16x3_tn_kernel");
}
