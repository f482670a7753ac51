// Y[M][N] = X[M][K] W^T + bias for a handful of output columns (N <= 16): the last layer of the DifHead classifier
// (models.py:523-543 via pretrainer.py:819-832: hidden -> nhead logits on N_nodes * nhead rows, 8M x 256 -> 8 at the bench
// size).  A GEMV-shaped, HBM-bound operation - 1 KB read per 32 B written - that hipBLASLt's fp32 GEMM ran at 1.2 TB/s
// (6.6 ms for 8 GB).  One wave per row batch: every lane keeps its K/64-column slice of all N weight rows in registers,
// reads a float4 of the row (one coalesced KB per wave), forms N partial dot products and the transposed butterfly
// (disgat_common.h: multi_reduce) leaves output column h in lane group h.  fp32 FMA throughout.
#include "disgat_common.h"
#include "disgat_api.h"

namespace disgat {

struct SkinnyArgs {
  const float* X;
  int64_t ldx, M;
  const float* W;      // [N][ldw] (nn.Linear weight layout)
  int64_t ldw;
  const float* bias;   // [N] or null
  float* Y;
  int64_t ldy;
  int N;               // valid columns (<= 1 << NL)
};

template <int NL, int KC>        // 2^NL padded output columns, KC chunks of 256 input columns
__global__ __launch_bounds__(DISGAT_BLOCK) void linear_skinny_kernel(const SkinnyArgs A) {
  constexpr int NP = 1 << NL;
  constexpr int G = 64 >> NL;                       // lanes per output column after the reduction
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * DISGAT_WAVES_PER_BLOCK;
  f32x4 w[NP][KC];
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int q = 0; q < KC; ++q)
      w[c][q] = c < A.N ? ld4(A.W + (int64_t)c * A.ldw + q * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int col = lane >> (6 - NL);
  const float b = (A.bias != nullptr && col < A.N) ? A.bias[col] : 0.f;
  constexpr int R = 4;                              // rows in flight per wave
  for (int64_t r0 = wave * R; r0 < A.M; r0 += n_waves * R) {
    f32x4 x[R][KC];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i < A.M ? r0 + i : A.M - 1;
#pragma unroll
      for (int q = 0; q < KC; ++q) x[i][q] = ld4(A.X + r * A.ldx + q * 256 + lane * 4);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float part[NP];
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < KC; ++q) acc = dot4(w[c][q], x[i][q], acc);
        part[c] = acc;
      }
      const float tot = multi_reduce<NL>(part);
      if (r0 + i < A.M && (lane & (G - 1)) == 0 && col < A.N) A.Y[(r0 + i) * A.ldy + col] = tot + b;
    }
  }
}

// Weight gradient of the same layer: dW[N][K] = G^T X with G [M][N] (ldg), reducing over the M rows - hipBLASLt's fp32 GEMM
// took 6.6 ms for the 8 GB of X at the bench size.  Every wave sums its rows into N x K/64 float4 registers per lane and
// stores one partial [N][K]; the host adds the partials (a few thousand, in a fixed order).
struct SkinnyWgradArgs {
  const float* X;
  int64_t ldx, M;
  const float* G;
  int64_t ldg;
  float* part;         // [n_waves][N][K]
  int N, K;
};

template <int NP, int KC>
__global__ __launch_bounds__(DISGAT_BLOCK) void linear_skinny_wgrad_kernel(const SkinnyWgradArgs A) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * DISGAT_WAVES_PER_BLOCK;
  f32x4 acc[NP][KC];
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int q = 0; q < KC; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int R = 4;
  for (int64_t r0 = wave * R; r0 < A.M; r0 += n_waves * R) {
    f32x4 x[R][KC];
    float g[R][NP];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const bool ok = r0 + i < A.M;
      const int64_t r = ok ? r0 + i : A.M - 1;
#pragma unroll
      for (int q = 0; q < KC; ++q) x[i][q] = ld4(A.X + r * A.ldx + q * 256 + lane * 4);
#pragma unroll
      for (int c = 0; c < NP; ++c) g[i][c] = (ok && c < A.N) ? A.G[r * A.ldg + c] : 0.f;    // wave-uniform addresses
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int q = 0; q < KC; ++q) acc[c][q] += g[i][c] * x[i][q];
  }
  float* p = A.part + wave * (int64_t)A.N * A.K + lane * 4;
#pragma unroll
  for (int c = 0; c < NP; ++c)
    if (c < A.N)
#pragma unroll
      for (int q = 0; q < KC; ++q) st4(p + (int64_t)c * A.K + q * 256, acc[c][q]);
}

template <int NL>
static int launch_skinny(const SkinnyArgs& A, int kc, hipStream_t s) {
  const int64_t want = (A.M + 15) / 16;             // >= 4 row batches per wave
  const int grid = (int)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
  switch (kc) {
    case 1: hipLaunchKernelGGL((linear_skinny_kernel<NL, 1>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A); break;
    case 2: hipLaunchKernelGGL((linear_skinny_kernel<NL, 2>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A); break;
    default: return fail(-2, "linear_skinny: K must be 256 or 512");
  }
  return check_launch("linear_skinny_kernel");
}

}  // namespace disgat

extern "C" int disgat_linear_skinny(const float* X, int64_t ldx, int64_t M, int K, const float* W, int64_t ldw,
                                    const float* bias, int N, float* Y, int64_t ldy, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0) return 0;
  DISGAT_REQUIRE(X && W && Y && M > 0 && N > 0 && N <= 16, "linear_skinny: null pointer or N=%d outside 1..16", N);
  DISGAT_REQUIRE(K == 256 || K == 512, "linear_skinny: K=%d must be 256 or 512", K);
  DISGAT_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K && ldy >= N && aligned16(X) && aligned16(W),
                 "linear_skinny: X / W rows must be 16-byte aligned and at least K long");
  SkinnyArgs A{X, ldx, M, W, ldw, bias, Y, ldy, N};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int kc = K / 256;
  if (N <= 2) return launch_skinny<1>(A, kc, s);
  if (N <= 4) return launch_skinny<2>(A, kc, s);
  if (N <= 8) return launch_skinny<3>(A, kc, s);
  return launch_skinny<4>(A, kc, s);
}

extern "C" int disgat_linear_skinny_wgrad(const float* X, int64_t ldx, int64_t M, int K, const float* G, int64_t ldg, int N,
                                          float* partials, int n_waves, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(X && G && partials && M > 0 && N > 0 && N <= 16, "linear_skinny_wgrad: null pointer or N=%d outside 1..16", N);
  DISGAT_REQUIRE(K == 256 || K == 512, "linear_skinny_wgrad: K=%d must be 256 or 512", K);
  DISGAT_REQUIRE(ldx % 4 == 0 && ldx >= K && ldg >= N && aligned16(X) && aligned16(partials),
                 "linear_skinny_wgrad: X rows / partials must be 16-byte aligned");
  DISGAT_REQUIRE(n_waves > 0 && n_waves % DISGAT_WAVES_PER_BLOCK == 0, "linear_skinny_wgrad: n_waves must be a positive multiple of %d",
                 DISGAT_WAVES_PER_BLOCK);
  SkinnyWgradArgs A{X, ldx, M, G, ldg, partials, N, K};
  const dim3 grid(n_waves / DISGAT_WAVES_PER_BLOCK), block(DISGAT_BLOCK);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DISGAT_SW(NP_)                                                                                  \
  do {                                                                                                  \
    if (K == 256) hipLaunchKernelGGL((linear_skinny_wgrad_kernel<NP_, 1>), grid, block, 0, s, A);       \
    else hipLaunchKernelGGL((linear_skinny_wgrad_kernel<NP_, 2>), grid, block, 0, s, A);                \
  } while (0)
  if (N <= 2) DISGAT_SW(2);
  else if (N <= 4) DISGAT_SW(4);
  else if (N <= 8) DISGAT_SW(8);
  else DISGAT_SW(16);
#undef DISGAT_SW
  return check_launch("linear_skinny_wgrad_kernel");
}
