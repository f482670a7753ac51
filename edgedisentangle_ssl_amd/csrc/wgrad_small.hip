// Weight gradients with a SMALL result: out[b] = A[b]^T G[b], A [M, K], G [M, N], K x N a few tiles, M the node count.
//
// Reference: the matmuls autograd runs for `torch.mm(h, W)` / `nn.Linear` backward (layers.py:350, 363, 376, 398, 110;
// models.py:538) when nhid = 64 - the layers of the bundled graphs.  The reduction runs over all node rows while the result is
// 64 x 64 ... 64 x 512: a library GEMM puts a handful of workgroups on it (25-30 us at 2 700 rows, 77-430 us at 19 793 - most
// of what a captured epoch on those graphs still spent in one kernel class), and the K >= 128, M >= 4 096 split-K MFMA kernel
// (disgat_gemm_f16x3_tn) does not tile it.  Here: plain fp32 FMAs, 64 x 64 output tiles, the rows cut into `splits` ranges so
// that tiles x splits fills the chip; a second launch adds the ranges' partial results in range order (deterministic; skipped
// when there is one range).
#include "disgat_api.h"
#include "disgat_common.h"

namespace {
using namespace disgat;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WG_T = 64;      // output tile edge
constexpr int WG_R = 16;      // rows per LDS step

struct WgArgs {
  const float* a;
  int64_t lda, a_bs;
  const float* g;
  int64_t ldg, g_bs;
  float* out;          // splits == 1: the result [batch][K][N]; else the partial records [batch][splits][K][N]
  int M, K, N, splits, rows_per, kt, nt;
};

__global__ __launch_bounds__(256) void wgrad_small_kernel(const WgArgs A) {
  __shared__ __attribute__((aligned(16))) float As[2][WG_R][WG_T + 4];
  __shared__ __attribute__((aligned(16))) float Gs[2][WG_R][WG_T + 4];
  const int tid = threadIdx.x;
  const int bz = blockIdx.y;
  const int tile = blockIdx.x / A.splits, sp = blockIdx.x - tile * A.splits;
  const int k0 = (tile / A.nt) * WG_T, n0 = (tile % A.nt) * WG_T;
  const int m_lo = sp * A.rows_per, m_hi = min(A.M, m_lo + A.rows_per);
  const float* a = A.a + (int64_t)bz * A.a_bs;
  const float* g = A.g + (int64_t)bz * A.g_bs;
  // loader role: row lr of the step, float4 column lc of either tile
  const int lr = tid >> 4, lc = (tid & 15) * 4;
  const bool a_ok = k0 + lc < A.K, g_ok = n0 + lc < A.N;       // K, N are multiples of 4: a float4 is inside or outside
  const int ty = tid >> 4, tx = tid & 15;                       // compute role: k = k0 + 4 ty .. +3, n = n0 + 4 tx .. +3
  f32x2 acc[4][2];              // pairs of neighbouring n: one v_pk_fma_f32 per pair (the plain form ran at half the FMA rate)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x2{0.f, 0.f};
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int m, f32x4& va, f32x4& vg) {
    const bool in = m + lr < m_hi;
    va = (in && a_ok) ? ld4(a + (int64_t)(m + lr) * A.lda + k0 + lc) : zero;
    vg = (in && g_ok) ? ld4(g + (int64_t)(m + lr) * A.ldg + n0 + lc) : zero;
  };
  f32x4 va, vg;
  if (m_lo < m_hi) fetch(m_lo, va, vg);
  int buf = 0;
  for (int m = m_lo; m < m_hi; m += WG_R, buf ^= 1) {
    *reinterpret_cast<f32x4*>(&As[buf][lr][lc]) = va;
    *reinterpret_cast<f32x4*>(&Gs[buf][lr][lc]) = vg;
    __syncthreads();                       // (two buffers: the next step's stores cannot overtake this step's reads)
    if (m + WG_R < m_hi) fetch(m + WG_R, va, vg);
#pragma unroll
    for (int r = 0; r < WG_R; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(&As[buf][r][ty * 4]);
      const f32x4 y = *reinterpret_cast<const f32x4*>(&Gs[buf][r][tx * 4]);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      const f32x2 y01{y.x, y.y}, y23{y.z, y.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] += f32x2{xs[i], xs[i]} * y01;
        acc[i][1] += f32x2{xs[i], xs[i]} * y23;
      }
    }
  }
  float* out = A.out + (((int64_t)bz * A.splits + sp) * A.K) * A.N;
  if (n0 + tx * 4 < A.N) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + ty * 4 + i;
      if (k < A.K) st4(out + (int64_t)k * A.N + n0 + tx * 4, f32x4{acc[i][0].x, acc[i][0].y, acc[i][1].x, acc[i][1].y});
    }
  }
}

// out[b][i] = sum_s part[b][s][i], s ascending; i counts float4s of the K x N result
__global__ __launch_bounds__(256) void wgrad_small_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int splits,
                                                              int64_t kn4, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int64_t b = i / kn4, r = i - b * kn4;
  const float* p = part + (b * splits * kn4 + r) * 4;
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld4(p + (int64_t)(s + j) * kn4 * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; s < splits; ++s) acc += ld4(p + (int64_t)s * kn4 * 4);
  st4(out + i * 4, acc);
}

}  // namespace

extern "C" int disgat_wgrad_small(const float* A, int64_t lda, int64_t a_batch_stride, const float* G, int64_t ldg,
                                  int64_t g_batch_stride, int M, int K, int N, int batch, int splits, float* partials,
                                  float* out, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(out && K > 0 && N > 0 && batch > 0 && M >= 0, "wgrad_small: null output / bad sizes");
  DISGAT_REQUIRE(K % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldg % 4 == 0 && a_batch_stride % 4 == 0 && g_batch_stride % 4 == 0,
                 "wgrad_small: K, N and every stride must be multiples of 4 floats");
  DISGAT_REQUIRE(splits >= 1 && splits <= 1024 && (splits == 1 || partials != nullptr), "wgrad_small: splits=%d needs the partials scratch", splits);
  DISGAT_REQUIRE(M == 0 || (A && G && aligned16(A) && aligned16(G)), "wgrad_small: operands must be non-null and 16-byte aligned");
  DISGAT_REQUIRE(aligned16(out) && (partials == nullptr || aligned16(partials)), "wgrad_small: outputs must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int kt = (K + WG_T - 1) / WG_T, nt = (N + WG_T - 1) / WG_T;
  int rows_per = (M + splits - 1) / splits;
  rows_per = (rows_per + WG_R - 1) / WG_R * WG_R;
  if (rows_per == 0) rows_per = WG_R;
  const int64_t blocks = (int64_t)kt * nt * splits;
  DISGAT_REQUIRE(blocks < ((int64_t)1 << 31) && batch < 65536, "wgrad_small: grid too large");
  WgArgs W{A, lda, a_batch_stride, G, ldg, g_batch_stride, splits == 1 ? out : partials, M, K, N, splits, rows_per, kt, nt};
  hipLaunchKernelGGL(wgrad_small_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, st, W);
  if (splits > 1) {
    const int64_t kn4 = (int64_t)K * N / 4, total4 = kn4 * batch;
    hipLaunchKernelGGL(wgrad_small_sum_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, partials, out, splits, kn4, total4);
  }
  return check_launch("wgrad_small_kernel");
}
