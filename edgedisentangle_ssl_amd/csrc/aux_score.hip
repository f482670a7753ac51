// Auxiliary node-pair scoring + pair-loss partial sums for gfx950.
//
// Replaces the aux branch of DisGALayer.forward_sparse (layers.py:355-360, 368-372, 381-389)
// for all heads at once: the same score formula as the edge pass, evaluated on an arbitrary
// int64 (2,M) pair list (pairs need not be edges; the reference's samplers emit them in
// row-major order, pretrainer.py:703).  One wave64 per 64 consecutive pairs; the row-side
// operand stays in registers while consecutive pairs share their row (sorted lists: ~M/N
// pairs per row), the column-side operand is gathered per pair with a 2-deep pipeline.
// Only heads [h_lo, h_hi) are evaluated: DisEdge supervises the first H/2 heads on the homo
// list and the last H/2 on the hetero list (pretrainer.py:619-620), so half the gather is
// skipped there.  Output layout [H][M] (each head contiguous = the reference's [M,1] tensors).
#include "disgat_common.h"

namespace disgat {

struct AuxArgs {
  const int64_t* pr;
  const int64_t* pc;
  int64_t M;
  int N, F_in;
  int h_lo, h_hi;
  const float* x;
  int ldx;
  const float* rowop;
  int ld_row;
  const float* colop;
  int ld_col;
  const float* a;
  float* out;
  uint32_t* sign;   // att 3, optional: [M][64] sign words for the backward pass (disgat_common.h)
  int batches;      // att 3 / 4: 64-pair batches per wave (1 .. AUX3_BATCHES)
};

// 64-pair batches one wave walks (its `a` vector and a row operand spanning batches are loaded once): same-box T_iter
// at C4 with 1 / 2 / 4 / 8: 6363 / 6370 / 6397 / 6400 GB/s.  Lists too short to give every SIMD a few waves that way walk
// fewer (AuxArgs::batches: a 45 K-pair Cora list is 176 waves of 256 pairs, one fifth of the chip's SIMDs, or 707 of 64)
#ifndef AUX3_BATCHES
#define AUX3_BATCHES 4
#endif
// att 3: lane = (head, g) exactly as in edge_fwd_kernel<3,...>.  DOT = att 4 (att 2 over the per-head projected
// operands, layers.py:362-365): e = <P[row][h][:], Q[col][h][:]>, same lane map, no `a`, no nonlinearity.
template <int HL, int QN, bool SIGN, bool DOT = false>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void aux_att3_kernel(const AuxArgs A) {
  constexpr int GL = 6 - HL;
  constexpr int G = 1 << GL;
  constexpr int FQ = QN * G * 4;
  const int lane = threadIdx.x & 63;
  const int64_t mw = ((int64_t)blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6)) * (64 * A.batches);
  if (mw >= A.M) return;
  const int myh = lane >> GL;
  const bool active = (myh >= A.h_lo) && (myh < A.h_hi);
  const int qoff = myh * FQ + (lane & (G - 1)) * 4;
  int64_t m0 = mw;                 // the wave's current batch of 64 pairs (AUX3_BATCHES consecutive batches per wave:
  int cnt = 0, rv = 0, cv = 0;     // `a` and a row operand that spans batches are loaded once)

  f32x4 a_r[QN], p_r[QN], qA[QN], qB[QN];
#pragma unroll
  for (int j = 0; j < QN; ++j) {
    a_r[j] = (active && !DOT) ? ld4(A.a + qoff + j * G * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    p_r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int cur_r = -1;

  auto load_q = [&](f32x4(&q)[QN], int c) {
    if (active) {
      const float* qp = A.colop + (size_t)c * A.ld_col + qoff;
#pragma unroll
      for (int j = 0; j < QN; ++j) q[j] = ld4g(qp + j * G * 4);
    }
  };
  // scores of the chunk are parked in registers (lane g of a head group keeps the pairs with
  // i % G == g) and written once per chunk as G-float (32-B for H=8) runs instead of 64 scattered
  // 4-byte stores per head: 8x fewer store instructions and no 4-byte partial-line writes
  constexpr int KEEP = 64 / G;
  float keep[KEEP];
#pragma unroll
  for (int t = 0; t < KEEP; ++t) keep[t] = 0.f;
  const int g = lane & (G - 1);
  // What the sign-recording variant costs over the plain one at C4 (52-54 against 44-45 ms per list; round 5, same-box builds,
  // profiles/r05/sign_record.md): NOT its vector instructions (a hand-laid v_pk_add / v_pk_mul form with 29 % fewer of them:
  // same time), not its lower occupancy (2 waves per SIMD against 3: a third gather buffer, 6 rows in flight per SIMD as in
  // the plain variant: same time), not the cache policy of the record store (non-temporal: same time) and not the static
  // wait counts (the row loads are conditional, so the compiler's vmcnt for the current pair also waits for the prefetch of
  // the next one; with unconditional loads the counts run one pair ahead: 0.5 ms of 263 per step).  It IS the store: with
  // the record computed but not written (-DDISGAT_SIGN_NOSTORE) the kernel takes 45.9 ms - 17 GB of writes into a read stream
  // that already saturates HBM are served at about a third of the read rate.
  auto compute = [&](const f32x4(&q)[QN], int i) {
    const int r = __builtin_amdgcn_readlane(rv, i);
    if (r != cur_r) {  // wave-uniform
      cur_r = r;
      if (active) {
        const float* pp = A.rowop + (size_t)r * A.ld_row + qoff;
#pragma unroll
        for (int j = 0; j < QN; ++j) p_r[j] = ld4(pp + j * G * 4);
      }
    }
    float acc = 0.f;
    if constexpr (DOT) {
#pragma unroll
      for (int j = 0; j < QN; ++j) acc = dot4(p_r[j], q[j], acc);
    } else if constexpr (SIGN) {
      SignAcc sg;
#pragma unroll
      for (int j = 0; j < QN; ++j) acc = dot4_lrelu_sign(a_r[j], p_r[j], q[j], acc, sg, j);
      // every lane stores (whole 64-word rows; the words of unscored heads are never read): a store under
      // `active` also splits the block and costs a second v_max per feature (z no longer known canonical)
#if defined(DISGAT_SIGN_NOSTORE)     // timing ablation (tools/prof_sign.sh): the record is computed and kept alive, not stored
      asm volatile("" ::"v"(sg.word()));
#else
      A.sign[(m0 + i) * 64 + lane] = sg.word();
#endif
    } else {
#pragma unroll
      for (int j = 0; j < QN; ++j) acc = dot4_lrelu(a_r[j], p_r[j], q[j], acc);
    }
    acc = group_sum<GL>(acc);
    const bool mine = (i & (G - 1)) == g;
    const int slot = i >> GL;
#pragma unroll
    for (int t = 0; t < KEEP; ++t) keep[t] = (mine && slot == t) ? acc : keep[t];
  };

  for (int b = 0; b < A.batches; ++b) {
    m0 = mw + (int64_t)b * 64;
    if (m0 >= A.M) break;
    cnt = (int)min((int64_t)64, A.M - m0);
    rv = (lane < cnt) ? (int)A.pr[m0 + lane] : 0;
    cv = (lane < cnt) ? (int)A.pc[m0 + lane] : 0;
    load_q(qA, __builtin_amdgcn_readlane(cv, 0));
    int i = 0;
    for (; i + 1 < cnt; i += 2) {
      load_q(qB, __builtin_amdgcn_readlane(cv, i + 1));
      compute(qA, i);
      if (i + 2 < cnt) load_q(qA, __builtin_amdgcn_readlane(cv, i + 2));
      compute(qB, i + 1);
    }
    if (i < cnt) compute(qA, i);
    if (active) {
      float* op = A.out + (int64_t)myh * A.M + m0 + g;
#pragma unroll
      for (int t = 0; t < KEEP; ++t)
        if (t * G + g < cnt) op[t * G] = keep[t];
    }
  }
}

// att 2: e = <P[row][h][:], x[col][:]>, coalesced x mapping (lane*4 floats); the H dot products of a
// pair are reduced together (multi_reduce) into head groups of G lanes, scores parked like att 3.
template <int HL, int XN>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void aux_att2_kernel(const AuxArgs A) {
  constexpr int H = 1 << HL;
  constexpr int GL = 6 - HL;
  constexpr int G = 1 << GL;
  const int lane = threadIdx.x & 63;
  const int64_t m0 = ((int64_t)blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 64;
  if (m0 >= A.M) return;
  const int cnt = (int)min((int64_t)64, A.M - m0);
  const int myh = lane >> GL;
  const int g = lane & (G - 1);
  const bool active = (myh >= A.h_lo) && (myh < A.h_hi);
  const int xoff = lane * 4;
  const int rv = (lane < cnt) ? (int)A.pr[m0 + lane] : 0;
  const int cv = (lane < cnt) ? (int)A.pc[m0 + lane] : 0;
  f32x4 p_r[H * XN], xA[XN], xB[XN];
#pragma unroll
  for (int j = 0; j < H * XN; ++j) p_r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur_r = -1;
  constexpr int KEEP = 64 / G;
  float keep[KEEP];
#pragma unroll
  for (int t = 0; t < KEEP; ++t) keep[t] = 0.f;

  auto load_x = [&](f32x4(&xv)[XN], int c) {
    const float* xp = A.x + (size_t)c * A.ldx + xoff;
#pragma unroll
    for (int t = 0; t < XN; ++t) xv[t] = (t * 256 + xoff < A.F_in) ? ld4(xp + t * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto compute = [&](const f32x4(&xv)[XN], int i) {
    const int r = __builtin_amdgcn_readlane(rv, i);
    if (r != cur_r) {
      cur_r = r;
      const float* pp = A.rowop + (size_t)r * A.ld_row;
#pragma unroll
      for (int hh = 0; hh < H; ++hh)
        if (hh >= A.h_lo && hh < A.h_hi) {
#pragma unroll
          for (int t = 0; t < XN; ++t) {
            const int o = t * 256 + xoff;
            p_r[hh * XN + t] = (o < A.F_in) ? ld4(pp + hh * A.F_in + o) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
    }
    float part[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < XN; ++t) acc = dot4(p_r[hh * XN + t], xv[t], acc);   // unscored heads keep p_r = 0
      part[hh] = acc;
    }
    const float e = multi_reduce<HL>(part);
    const bool mine = (i & (G - 1)) == g;
    const int slot = i >> GL;
#pragma unroll
    for (int t = 0; t < KEEP; ++t) keep[t] = (mine && slot == t) ? e : keep[t];
  };

  // 2-deep gather pipeline.  Measured on the same box: a 4-deep one (as in the att-1 / att-2 edge pass) 5.09 vs 5.14 TB/s,
  // a v_pk_fma_f32 form of the dot products 4.77 vs 5.09 - the kernel is bound by the H dot products + butterfly, not
  // by latency
  load_x(xA, __builtin_amdgcn_readlane(cv, 0));
  int i = 0;
  for (; i + 1 < cnt; i += 2) {
    load_x(xB, __builtin_amdgcn_readlane(cv, i + 1));
    compute(xA, i);
    if (i + 2 < cnt) load_x(xA, __builtin_amdgcn_readlane(cv, i + 2));
    compute(xB, i + 1);
  }
  if (i < cnt) compute(xA, i);
  if (active) {
    float* op = A.out + (int64_t)myh * A.M + m0 + g;
#pragma unroll
    for (int t = 0; t < KEEP; ++t)
      if (t * G + g < cnt) op[t * G] = keep[t];
  }
}

// att 1: e = s1[row][h] + s2[col][h]; one thread per pair, heads in float4 groups (writes coalesced
// per head across the threads of a wave).  VEC4: h_lo, h_hi, ld_row, ld_col multiples of 4.
template <bool VEC4>
__global__ __launch_bounds__(256) void aux_att1_kernel(const AuxArgs A) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < A.M; m += stride) {
    const float* s1 = A.rowop + (size_t)A.pr[m] * A.ld_row;
    const float* s2 = A.colop + (size_t)A.pc[m] * A.ld_col;
    if constexpr (VEC4) {
      for (int h = A.h_lo; h < A.h_hi; h += 4) {
        const f32x4 v = ld4(s1 + h) + ld4(s2 + h);
        A.out[(int64_t)(h + 0) * A.M + m] = v.x;
        A.out[(int64_t)(h + 1) * A.M + m] = v.y;
        A.out[(int64_t)(h + 2) * A.M + m] = v.z;
        A.out[(int64_t)(h + 3) * A.M + m] = v.w;
      }
    } else {
      for (int h = A.h_lo; h < A.h_hi; ++h) A.out[(int64_t)h * A.M + m] = s1[h] + s2[h];
    }
  }
}

// pred = sigmoid(sum_{h in [h_lo,h_hi)} aux[h][m]); squared error against 0/1 labels, split by
// label so the host can apply adj_mse_loss's class weights (utils.py:287-298):
//   acc[0] += sum_{t!=0} (pred-t)^2, acc[1] += sum_{t==0} (pred-t)^2, acc[2] += #{t!=0}
// Entries with a negative label are padding (fixed-capacity lists of captured steps): skipped here, zero gradient below.
__global__ __launch_bounds__(256) void pair_loss_kernel(const float* __restrict__ aux, int64_t M, int h_lo, int h_hi,
                                                        const float* __restrict__ labels, double* __restrict__ part) {
  double sp = 0.0, sn = 0.0, np_ = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
    float s = 0.f;
    for (int h = h_lo; h < h_hi; ++h) s += aux[(int64_t)h * M + m];
    const float t = labels[m];
    if (t < 0.f) continue;                     // padding of a fixed-capacity list: in no sum
    const float d = sigmoidf_(s) - t;
    if (t != 0.f) {
      sp += (double)(d * d);
      np_ += 1.0;
    } else {
      sn += (double)(d * d);
    }
  }
  __shared__ double red[3][DISGAT_WAVES_PER_BLOCK];
  double v[3] = {sp, sn, np_};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_down(v[q], off, 64);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < DISGAT_WAVES_PER_BLOCK; ++w) t += red[threadIdx.x][w];
    part[(size_t)blockIdx.x * 3 + threadIdx.x] = t;         // one partial per block: the final sum runs in block order
  }
}

// acc[q] += sum over blocks of part[b][q] in a FIXED order (lane l adds blocks l, l+64, ... in sequence, then a fixed
// shuffle tree): deterministic, no floating-point atomics anywhere in the loss.  One wave per quantity.
// value (optional) = {loss, neg_w, m}: utils.adj_mse_loss on the whole list (utils.py:287-298: positives weigh 1, zeros
// n_pos / (m^2 - n_pos), mean over the m valid pairs) - the scalar arithmetic the host formulation spent ~10 launches on.
__global__ __launch_bounds__(192) void pair_loss_finish_kernel(const double* __restrict__ part, int n_blocks, double* __restrict__ acc,
                                                               const double* __restrict__ count, double m_host,
                                                               double* __restrict__ value, float* __restrict__ loss32) {
  __shared__ double tot[3];
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double t = 0.0;
  for (int b = lane; b < n_blocks; b += 64) t += part[(size_t)b * 3 + q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if (lane == 0) {
    acc[q] = t;
    tot[q] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0 && value != nullptr) {
    const double m = count != nullptr ? *count : m_host;
    const double neg_w = tot[2] / (m * m - tot[2]);
    const double loss = (tot[0] + neg_w * tot[1]) / m;
    value[0] = loss;
    value[1] = neg_w;
    value[2] = m;
    if (loss32 != nullptr) *loss32 = (float)loss;
  }
}

// Backward of the same loss: d loss / d aux[h][m] = coef[t_m != 0 ? 0 : 1] * 2 (p - t) p (1 - p) for h in [h_lo, h_hi), 0 for
// the other rows (coef = upstream gradient x class weight / M, prepared by the host on the device).
__global__ __launch_bounds__(256) void pair_loss_bwd_kernel(const float* __restrict__ aux, int64_t M, int H, int h_lo, int h_hi,
                                                            const float* __restrict__ labels, const float* __restrict__ coef,
                                                            const double* __restrict__ value, const float* __restrict__ gout,
                                                            float* __restrict__ g, int transposed) {
  float c_pos, c_neg;
  if (coef != nullptr) {
    c_pos = coef[0];
    c_neg = coef[1];
  } else {                       // from the forward's {loss, neg_w, m} and the upstream gradient
    const double per = (double)gout[0] / value[2];
    c_pos = (float)per;
    c_neg = (float)(value[1] * per);
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
    float s = 0.f;
    for (int h = h_lo; h < h_hi; ++h) s += aux[(int64_t)h * M + m];
    const float t = labels[m];
    const float p = sigmoidf_(s);
    const float gv = t < 0.f ? 0.f : (t != 0.f ? c_pos : c_neg) * 2.0f * (p - t) * p * (1.0f - p);   // t < 0: padding
    if (transposed) {        // [M][H]: the H gradients of a pair side by side (what the column-side segment pass gathers)
      for (int h = 0; h < H; ++h) g[m * H + h] = (h >= h_lo && h < h_hi) ? gv : 0.f;
    } else {
      for (int h = 0; h < H; ++h) g[(int64_t)h * M + m] = (h >= h_lo && h < h_hi) ? gv : 0.f;
    }
  }
}

}  // namespace disgat

// ------------------------------------------------------------------------------------------
// C ABI
#include "disgat_api.h"

namespace disgat {

template <int HL, bool SIGN, bool DOT = false>
static int launch_aux3(int qn, const AuxArgs& A, int grid, hipStream_t s) {
  switch (qn) {
    case 1: hipLaunchKernelGGL((aux_att3_kernel<HL, 1, SIGN, DOT>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A); break;
    case 2: hipLaunchKernelGGL((aux_att3_kernel<HL, 2, SIGN, DOT>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A); break;
    case 4: hipLaunchKernelGGL((aux_att3_kernel<HL, 4, SIGN, DOT>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A); break;
    case 8: hipLaunchKernelGGL((aux_att3_kernel<HL, 8, SIGN, DOT>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A); break;
    default: return fail(-2, "aux_score att=3: F_out must be QN*(64/H)*4 with QN in {1,2,4,8}");
  }
  return check_launch("aux_att3_kernel");
}

template <int HL>
static int launch_aux2(int xn, const AuxArgs& A, int grid, hipStream_t s) {
  if (xn == 1) {
    hipLaunchKernelGGL((aux_att2_kernel<HL, 1>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A);
  } else if (xn == 2 && HL <= 3) {
    if constexpr (HL <= 3) hipLaunchKernelGGL((aux_att2_kernel<HL, 2>), dim3(grid), dim3(DISGAT_BLOCK), 0, s, A);
  } else {
    return fail(-2, "aux_score att=2: F_in=%d too wide for H=%d", A.F_in, 1 << HL);
  }
  return check_launch("aux_att2_kernel");
}

}  // namespace disgat

extern "C" int disgat_aux_score(int att, const int64_t* pair_rows, const int64_t* pair_cols, int64_t M, int N, int H,
                                int F_in, int F_out, int h_lo, int h_hi, const float* x, int ldx, const float* rowop,
                                int ld_row, const float* colop, int ld_col, const float* a, float* out,
                                uint32_t* sign_bits, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(att >= 1 && att <= 4, "aux_score: att=%d not in 1..3 (4 = att 2 over projected operands)", att);
  DISGAT_REQUIRE(M >= 0 && N > 0, "aux_score: bad sizes");
  if (M == 0 || h_lo >= h_hi) return 0;
  const int hl = ilog2_exact(H);
  DISGAT_REQUIRE(hl >= 1 && hl <= 4, "aux_score: H=%d must be a power of two in [2,16]", H);
  DISGAT_REQUIRE(h_lo >= 0 && h_hi <= H, "aux_score: head range [%d,%d) outside [0,%d)", h_lo, h_hi, H);
  DISGAT_REQUIRE(pair_rows && pair_cols && rowop && out, "aux_score: null pointer");
  AuxArgs A{pair_rows, pair_cols, M, N, F_in, h_lo, h_hi, x, ldx, rowop, ld_row, colop, ld_col, a, out,
            att == 3 ? sign_bits : nullptr, AUX3_BATCHES};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  while (A.batches > 1 && M < (int64_t)64 * A.batches * 8192) --A.batches;      // short lists: more, shorter waves
  const int64_t per_wave = (att == 3 || att == 4) ? 64 * A.batches : 64;
  const int64_t waves = (M + per_wave - 1) / per_wave;
  const int64_t grid64 = (waves + DISGAT_WAVES_PER_BLOCK - 1) / DISGAT_WAVES_PER_BLOCK;
  DISGAT_REQUIRE(grid64 < (int64_t)1 << 31, "aux_score: M too large for one launch");
  const int grid = (int)grid64;
  if (att == 1) {
    DISGAT_REQUIRE(colop && ld_row >= H && ld_col >= H, "aux_score att=1: bad s1/s2");
    const int g1 = (int)min((int64_t)16384, (M + 255) / 256);
    const bool vec = (h_lo % 4 == 0) && (h_hi % 4 == 0) && (ld_row % 4 == 0) && (ld_col % 4 == 0) && aligned16(rowop) && aligned16(colop);
    if (vec) hipLaunchKernelGGL(aux_att1_kernel<true>, dim3(g1), dim3(256), 0, s, A);
    else hipLaunchKernelGGL(aux_att1_kernel<false>, dim3(g1), dim3(256), 0, s, A);
    return check_launch("aux_att1_kernel");
  }
  if (att == 2) {
    DISGAT_REQUIRE(x && F_in > 0 && F_in % 4 == 0 && ldx % 4 == 0 && ld_row % 4 == 0 && ld_row >= H * F_in &&
                       aligned16(x) && aligned16(rowop),
                   "aux_score att=2: bad x/P strides or alignment");
    const int xn = (F_in + 255) / 256;
    switch (hl) {
      case 1: return launch_aux2<1>(xn, A, grid, s);
      case 2: return launch_aux2<2>(xn, A, grid, s);
      case 3: return launch_aux2<3>(xn, A, grid, s);
      default: return launch_aux2<4>(xn, A, grid, s);
    }
  }
  const int g4 = (64 >> hl) * 4;
  DISGAT_REQUIRE(F_out > 0 && F_out % g4 == 0, "aux_score att=3: F_out=%d must be a multiple of %d", F_out, g4);
  DISGAT_REQUIRE(colop && (a || att == 4) && ld_row % 4 == 0 && ld_col % 4 == 0 && ld_row >= H * F_out && ld_col >= H * F_out &&
                     aligned16(rowop) && aligned16(colop) && (att == 4 || aligned16(a)),
                 "aux_score att=3: bad P/Q strides or alignment");
  const int qn = F_out / g4;
  if (att == 4) {
    switch (hl) {
      case 1: return launch_aux3<1, false, true>(qn, A, grid, s);
      case 2: return launch_aux3<2, false, true>(qn, A, grid, s);
      case 3: return launch_aux3<3, false, true>(qn, A, grid, s);
      default: return launch_aux3<4, false, true>(qn, A, grid, s);
    }
  }
  if (A.sign != nullptr) {
    switch (hl) {
      case 1: return launch_aux3<1, true>(qn, A, grid, s);
      case 2: return launch_aux3<2, true>(qn, A, grid, s);
      case 3: return launch_aux3<3, true>(qn, A, grid, s);
      default: return launch_aux3<4, true>(qn, A, grid, s);
    }
  }
  switch (hl) {
    case 1: return launch_aux3<1, false>(qn, A, grid, s);
    case 2: return launch_aux3<2, false>(qn, A, grid, s);
    case 3: return launch_aux3<3, false>(qn, A, grid, s);
    default: return launch_aux3<4, false>(qn, A, grid, s);
  }
}

extern "C" int disgat_pair_loss(const float* aux, int64_t M, int h_lo, int h_hi, const float* labels, const double* count,
                                double* acc, double* block_partials, double* value, float* loss32, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(acc && block_partials && M >= 0 && h_lo >= 0 && h_hi >= h_lo, "pair_loss: bad arguments");
  DISGAT_REQUIRE(M == 0 || (aux && labels), "pair_loss: null scores / labels");
  DISGAT_REQUIRE(loss32 == nullptr || value != nullptr, "pair_loss: loss32 needs value");
  const int grid = (int)min((int64_t)DISGAT_PAIR_LOSS_MAX_BLOCKS, (M + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (grid > 0) {
    hipLaunchKernelGGL(pair_loss_kernel, dim3(grid), dim3(256), 0, st, aux, M, h_lo, h_hi, labels, block_partials);
    if (int rc = check_launch("pair_loss_kernel")) return rc;
  }
  hipLaunchKernelGGL(pair_loss_finish_kernel, dim3(1), dim3(192), 0, st, block_partials, grid, acc, count, (double)M, value, loss32);
  return check_launch("pair_loss_finish_kernel");
}

extern "C" int disgat_pair_loss_bwd(const float* aux, int64_t M, int H, int h_lo, int h_hi, const float* labels,
                                    const float* coef, const double* value, const float* gout, float* g, int g_transposed,
                                    disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(aux && labels && g && M >= 0 && H > 0 && h_lo >= 0 && h_hi >= h_lo && h_hi <= H, "pair_loss_bwd: bad arguments");
  DISGAT_REQUIRE(coef != nullptr || (value != nullptr && gout != nullptr), "pair_loss_bwd: either coef or (value, gout)");
  if (M == 0) return 0;
  const int grid = (int)min((int64_t)4096, (M + 255) / 256);
  hipLaunchKernelGGL(pair_loss_bwd_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), aux, M, H, h_lo,
                     h_hi, labels, coef, value, gout, g, g_transposed);
  return check_launch("pair_loss_bwd_kernel");
}
