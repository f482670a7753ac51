// Fused DISGAT edge pass for gfx950: per-edge score -> sigmoid -> row softmax -> weighted
// neighbour aggregation, all H heads of one layer in ONE pass over the CSR edges.
//
// Replaces, for all heads at once, the reference's DisGALayer.forward_sparse
// (layers.py:340-416) + utils.sp_softmax / utils.sp_matmul (utils.py:192-207).
//
// Dataflow (why it is one pass): the softmax logits are sigmoid(e) in (0,1), so
// exp() needs no max-shift and numerator sum_k w_k x[col_k] and denominator sum_k w_k
// accumulate together.  Aggregation commutes with the per-head output projection
// (sum_k a_k (x_k W) = (sum_k a_k x_k) W), so the neighbour row x[col] is gathered ONCE per
// edge for all heads and the kernel emits Z[N][H][F_in]; the projection is a dense GEMM.
//
// Mapping: one wave64 per work item (a CSR row, or a <=chunk slice of a hub row), 4
// independent waves per block, no LDS, no barriers.  Register-resident per wave: the row-side
// score operand, the H x F_in accumulators and a 2-deep software pipeline of gathered
// neighbour operands (the next edge's loads are in flight while the current one is scored).
//   att 3: lane = (head h = lane / G, g = lane % G), G = 64/H.  Lane (h,g) owns floats
//          h*F_out + (j*G+g)*4 .. +3 (j < QN) of the P/Q/a rows: every wave load touches H
//          full 128-B segments (G=8), and the per-head reduction is log2(G) DPP adds.
//   att 2: x[col] and P[row][h][:] coalesced (lane*4 floats); the H per-head dot products are
//          reduced together by one transposed butterfly (multi_reduce) into the same head groups.
//   att 1: score head of a lane = lane % H, s2[col][h] gathered directly.
//   att 4: att 2 in the reference's own formulation (layers.py:362-365): e = <h[row][h][:], h[col][h][:]> with
//          h = x W per head - the dot product runs over F_out instead of F_in, so the layer input may be arbitrarily
//          wide (raw bag-of-words features, --origin_feat); lane map, operand layout and padding of att 3
//          (rowop = colop = h [N][H*F_out]), no `a`, no nonlinearity.
#include "gemm_common.h"

namespace disgat {

struct EdgeFwdArgs {
  const int4* items;
  int n_items;
  const int32_t* col;
  int64_t E;
  int N, F_in;          // F_in: valid floats per x row (multiple of 4)
  const float* x;
  int ldx;
  const float* rowop;
  int ld_row;
  const float* colop;
  int ld_col;
  const float* a;
  float* Z;
  float* edge_e;
  float* den;
  float* part_z;
  float* part_den;
  int sage_div;
  DropCfg drop;
  uint32_t* sign;       // att 3, optional: [E][64] sign words (disgat_common.h) for the backward pass
  const float* e_in;    // optional [H][E]: added to the score before the sigmoid (partial scores of a head whose
                        // features are spread over several launches: heads wider than one launch's 1024 features)
  uint16_t* Zh;         // optional: Z as the two fp16 planes of the f16x3 GEMM that consumes it (gemm_planes.hip), same
  uint16_t* Zl;         // [N][H][F_in] order, instead of fp32 Z - same bytes, and the consumer splits nothing
  const float* z_bound; // device scalar >= max |Z| (an analytic bound: Z rows are convex combinations of x rows)
};

template <int ATT, int HL, int QN, int XN>
struct ColBuf {
  f32x4 q[(ATT == 3 || ATT == 4) ? QN : 1];
  f32x4 xv[XN];
  float s2;
};

template <int ATT, int HL, int QN, int XN, bool SIGN>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void edge_fwd_kernel(const EdgeFwdArgs A) {
  constexpr int H = 1 << HL;
  constexpr int GL = 6 - HL;
  constexpr int G = 1 << GL;
  constexpr int FQ = QN * G * 4;  // att 3: floats per head in the P/Q/a rows
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= A.n_items) return;
  const int4 it = A.items[item];
  const int row = rfl(it.x), kb = rfl(it.y), ke = rfl(it.z), slot = rfl(it.w);

  // which head this lane scores, and which lane to read head hh's weight from
  // att 2/3: head group = lane / G (the layout the multi-value reduction / the DPP group sums leave
  // the scores in); att 1: head = lane % H (scores are gathered directly per lane)
  constexpr bool PQ = (ATT == 3 || ATT == 4);       // per-head projected operands on both sides (att-3 lane map)
  const int myh = (ATT != 1) ? (lane >> GL) : (lane & (H - 1));
  auto head_lane = [](int hh) { return (ATT != 1) ? (hh << GL) : hh; };
  const int xoff = lane * 4;
  const int qoff = PQ ? (myh * FQ + (lane & (G - 1)) * 4) : 0;

  // ---- row-side operands (once per work item)
  f32x4 a_r[ATT == 3 ? QN : 1];
  f32x4 p_r[PQ ? QN : (ATT == 2 ? H * XN : 1)];
  float s1r = 0.f;
  if constexpr (PQ) {
    const float* pp = A.rowop + (size_t)row * A.ld_row + qoff;
#pragma unroll
    for (int j = 0; j < QN; ++j) {
      if constexpr (ATT == 3) a_r[j] = ld4(A.a + qoff + j * G * 4);
      p_r[j] = ld4(pp + j * G * 4);
    }
  } else if constexpr (ATT == 2) {
    const float* pp = A.rowop + (size_t)row * A.ld_row;
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int o = i * 256 + xoff;
        p_r[hh * XN + i] = (o < A.F_in) ? ld4(pp + hh * A.F_in + o) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  } else {
    s1r = A.rowop[(size_t)row * A.ld_row + myh];
  }

  f32x4 zacc[H * XN];
#pragma unroll
  for (int i = 0; i < H * XN; ++i) zacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float den = 0.f;   // softmax denominator: every edge
  float dsum = 0.f;  // sum of the dropped-out numerator weights (== den without dropout)

  using Buf = ColBuf<ATT, HL, QN, XN>;

  auto load_edge = [&](Buf& b, int c) {
    if constexpr (PQ) {
      const float* qp = A.colop + (size_t)c * A.ld_col + qoff;
#pragma unroll
      for (int j = 0; j < QN; ++j) b.q[j] = ld4g(qp + j * G * 4);
    } else if constexpr (ATT == 1) {
      b.s2 = A.colop[(size_t)c * A.ld_col + myh];
    }
    const float* xp = A.x + (size_t)c * A.ldx + xoff;
#pragma unroll
    for (int i = 0; i < XN; ++i)
      b.xv[i] = (i * 256 + xoff < A.F_in) ? ld4(xp + i * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // Raw scores are parked in registers per 64-edge batch (a lane keeps the batch positions i with
  // i % GE == ge for its head) and flushed as GE-float runs: 8x fewer store instructions and no
  // 4-byte partial-line writes (those cost the aux scorer 22 % before the same change).
  constexpr int GE = 64 >> HL;                       // lanes per head
  const int ge = (ATT != 1) ? (lane & (G - 1)) : (lane >> HL);
  float keep[H];
  auto compute = [&](const Buf& b, int64_t k, int i) {
    float e;
    if constexpr (ATT == 3 && SIGN) {
      float acc = 0.f;
      SignAcc sg;
#pragma unroll
      for (int j = 0; j < QN; ++j) acc = dot4_lrelu_sign(a_r[j], p_r[j], b.q[j], acc, sg, j);
      A.sign[k * 64 + lane] = sg.word();                         // one 64-word (256-B) row per edge, coalesced
      e = group_sum<GL>(acc);
    } else if constexpr (ATT == 3) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < QN; ++j) acc = dot4_lrelu(a_r[j], p_r[j], b.q[j], acc);
      e = group_sum<GL>(acc);
    } else if constexpr (ATT == 4) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < QN; ++j) acc = dot4(p_r[j], b.q[j], acc);
      e = group_sum<GL>(acc);
    } else if constexpr (ATT == 2) {
      float part[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        float acc = 0.f;     // scalar FMAs: a v_pk_fma_f32 form of these dot products measured 3 % slower (4.15 vs 4.28 TB/s)
#pragma unroll
        for (int t = 0; t < XN; ++t) acc = dot4(p_r[hh * XN + t], b.xv[t], acc);
        part[hh] = acc;
      }
      e = multi_reduce<HL>(part);
    } else {
      e = s1r + b.s2;
    }
    if (A.e_in != nullptr) e += A.e_in[(int64_t)myh * A.E + k];      // one address per head group: a broadcast load
    {
      const bool mine = (i & (GE - 1)) == ge;
      const int slot = i / GE;
#pragma unroll
      for (int t = 0; t < H; ++t) keep[t] = (mine && slot == t) ? e : keep[t];
    }
    const float w = softmax_num(e);
    den += w;
    const float wd = w * drop_mult(A.drop, k, myh, H);
    dsum += wd;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float wh = readlane_f(wd, head_lane(hh));
#pragma unroll
      for (int i = 0; i < XN; ++i) zacc[hh * XN + i] += wh * b.xv[i];
    }
  };

  // ---- edge loop: 64 column indices per coalesced load; gathered operands run ahead of the arithmetic in a register
  // pipeline - 2 deep for att 3 / 4 (8-9 KB per edge: two edges per wave already keep ~18 KB in flight), 4 deep for
  // att 1 / 2, whose edges bring only the 1 KB x row (a 2-deep pipeline left the wave waiting on HBM latency: 4.0-4.2
  // TB/s against 5.8 for att 3; a buffer is 4-5 registers there)
  auto flush = [&](int kbase, int cnt) {
    float* ep = A.edge_e + (int64_t)myh * A.E + kbase + ge;
#pragma unroll
    for (int t = 0; t < H; ++t)
      if (t * GE + ge < cnt) ep[t * GE] = keep[t];
  };
  if constexpr (PQ) {
    Buf bufA, bufB;
    for (int kbase = kb; kbase < ke; kbase += 64) {
      const int cnt = min(64, ke - kbase);
      const int cv = (lane < cnt) ? A.col[kbase + lane] : 0;
      load_edge(bufA, __builtin_amdgcn_readlane(cv, 0));
      int i = 0;
      for (; i + 1 < cnt; i += 2) {
        load_edge(bufB, __builtin_amdgcn_readlane(cv, i + 1));
        compute(bufA, (int64_t)kbase + i, i);
        if (i + 2 < cnt) load_edge(bufA, __builtin_amdgcn_readlane(cv, i + 2));
        compute(bufB, (int64_t)kbase + i + 1, i + 1);
      }
      if (i < cnt) compute(bufA, (int64_t)kbase + i, i);
      flush(kbase, cnt);
    }
  } else {
    Buf b0, b1, b2, b3;
    for (int kbase = kb; kbase < ke; kbase += 64) {
      const int cnt = min(64, ke - kbase);
      const int cv = (lane < cnt) ? A.col[kbase + lane] : 0;
      // lanes >= cnt hold column 0 (valid memory): loads past the batch end are harmless and never consumed
      load_edge(b0, __builtin_amdgcn_readlane(cv, 0));
      load_edge(b1, __builtin_amdgcn_readlane(cv, 1));
      load_edge(b2, __builtin_amdgcn_readlane(cv, 2));
      for (int i = 0; i < cnt; i += 4) {
        load_edge(b3, __builtin_amdgcn_readlane(cv, (i + 3) & 63));
        compute(b0, (int64_t)kbase + i, i);
        if (i + 1 < cnt) {
          load_edge(b0, __builtin_amdgcn_readlane(cv, (i + 4) & 63));
          compute(b1, (int64_t)kbase + i + 1, i + 1);
        }
        if (i + 2 < cnt) {
          load_edge(b1, __builtin_amdgcn_readlane(cv, (i + 5) & 63));
          compute(b2, (int64_t)kbase + i + 2, i + 2);
        }
        if (i + 3 < cnt) {
          load_edge(b2, __builtin_amdgcn_readlane(cv, (i + 6) & 63));
          compute(b3, (int64_t)kbase + i + 3, i + 3);
        }
      }
      flush(kbase, cnt);
    }
  }

  // ---- epilogue
  if (slot < 0) {
    float inv = (den > 0.f) ? 1.0f / den : 0.f;
    // SageConv divides the aggregate by rowsum(attention)+1 (layers.py:103), attention taken AFTER
    // dropout (layers.py:394, 402): rowsum = dsum*inv
    if (A.sage_div) inv = inv / (dsum * inv + 1.0f);
    if (A.Zh != nullptr) {
      const float sz = f16_scale(*A.z_bound);
      const size_t zo = (size_t)row * (H * A.F_in) + xoff;
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        const float sc = readlane_f(inv, head_lane(hh)) * sz;
#pragma unroll
        for (int i = 0; i < XN; ++i)
          if (i * 256 + xoff < A.F_in) {
            u32x2 h, l;
            split4h(zacc[hh * XN + i] * sc, h, l);
            *reinterpret_cast<u32x2*>(A.Zh + zo + hh * A.F_in + i * 256) = h;
            *reinterpret_cast<u32x2*>(A.Zl + zo + hh * A.F_in + i * 256) = l;
          }
      }
    } else {
      float* zp = A.Z + (size_t)row * (H * A.F_in) + xoff;
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        const float sc = readlane_f(inv, head_lane(hh));
#pragma unroll
        for (int i = 0; i < XN; ++i)
          if (i * 256 + xoff < A.F_in) st4(zp + hh * A.F_in + i * 256, zacc[hh * XN + i] * sc);
      }
    }
    if (A.den != nullptr && ((ATT != 1) ? ((lane & (G - 1)) == 0) : (lane < H))) {
      A.den[(size_t)row * (2 * H) + myh] = den;
      A.den[(size_t)row * (2 * H) + H + myh] = dsum;
    }
  } else {
    float* zp = A.part_z + (size_t)slot * (H * A.F_in) + xoff;
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
      for (int i = 0; i < XN; ++i)
        if (i * 256 + xoff < A.F_in) st4(zp + hh * A.F_in + i * 256, zacc[hh * XN + i]);
    if ((ATT != 1) ? ((lane & (G - 1)) == 0) : (lane < H)) {
      A.part_den[(size_t)slot * (2 * H) + myh] = den;
      A.part_den[(size_t)slot * (2 * H) + H + myh] = dsum;
    }
  }
}

// Split rows: sum the chunk partials in chunk order, then normalise like the epilogue above.
__global__ __launch_bounds__(256) void edge_combine_kernel(const int32_t* __restrict__ split_rows,
                                                           const int32_t* __restrict__ split_ptr, int H, int F_in,
                                                           const float* __restrict__ part_z,
                                                           const float* __restrict__ part_den, float* __restrict__ Z,
                                                           float* __restrict__ den_out, int sage_div,
                                                           uint16_t* __restrict__ Zh, uint16_t* __restrict__ Zl,
                                                           const float* __restrict__ z_bound) {
  const int s = blockIdx.x;
  const int row = split_rows[s];
  const int s0 = split_ptr[s], s1 = split_ptr[s + 1];
  const int HF = H * F_in;
  for (int idx = threadIdx.x * 4; idx < HF; idx += 256 * 4) {
    const int h = idx / F_in;
    float den = 0.f, dsum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sl = s0; sl < s1; sl += 8) {          // 8 independent loads in flight (a hub row has hundreds of slices),
      f32x4 pz[8];                                  // summed in slice order: deterministic
      float pd[8], ps[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = min(sl + u, s1 - 1);
        pz[u] = ld4(part_z + (size_t)q * HF + idx);
        pd[u] = part_den[(size_t)q * (2 * H) + h];
        ps[u] = part_den[(size_t)q * (2 * H) + H + h];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (sl + u < s1) {
          acc += pz[u];
          den += pd[u];
          dsum += ps[u];
        }
    }
    float inv = (den > 0.f) ? 1.0f / den : 0.f;
    if (sage_div) inv = inv / (dsum * inv + 1.0f);
    if (Zh != nullptr) {
      u32x2 h, l;
      split4h(acc * (inv * f16_scale(*z_bound)), h, l);
      *reinterpret_cast<u32x2*>(Zh + (size_t)row * HF + idx) = h;
      *reinterpret_cast<u32x2*>(Zl + (size_t)row * HF + idx) = l;
    } else {
      st4(Z + (size_t)row * HF + idx, acc * inv);
    }
    if (den_out != nullptr && idx % F_in == 0) {
      den_out[(size_t)row * (2 * H) + h] = den;
      den_out[(size_t)row * (2 * H) + H + h] = dsum;
    }
  }
}

}  // namespace disgat

// ------------------------------------------------------------------------------------------
// C ABI
#include "disgat_api.h"

namespace disgat {

template <int ATT, int HL, int QN, int XN>
static int launch_edge(const EdgeFwdArgs& args, hipStream_t stream) {
  const int grid = (args.n_items + DISGAT_WAVES_PER_BLOCK - 1) / DISGAT_WAVES_PER_BLOCK;
  if constexpr (ATT == 3) {
    if (args.sign != nullptr) {
      hipLaunchKernelGGL((edge_fwd_kernel<ATT, HL, QN, XN, true>), dim3(grid), dim3(DISGAT_BLOCK), 0, stream, args);
      return check_launch("edge_fwd_kernel");
    }
  }
  hipLaunchKernelGGL((edge_fwd_kernel<ATT, HL, QN, XN, false>), dim3(grid), dim3(DISGAT_BLOCK), 0, stream, args);
  return check_launch("edge_fwd_kernel");
}

template <int ATT, int HL, int QN>
static int launch_edge_x(int xn, const EdgeFwdArgs& args, hipStream_t stream) {
  if (xn == 1) return launch_edge<ATT, HL, QN, 1>(args, stream);
  if constexpr (HL <= 3) {
    if (xn == 2) return launch_edge<ATT, HL, QN, 2>(args, stream);
  }
  return fail(-2, "edge_fwd: F_in=%d too wide for H=%d (H*ceil(F_in/256) must be <= 16)", args.F_in, 1 << HL);
}

template <int ATT, int HL>
static int launch_edge_q(int qn, int xn, const EdgeFwdArgs& args, hipStream_t stream) {
  if constexpr (ATT == 3 || ATT == 4) {
    switch (qn) {
      case 1: return launch_edge_x<ATT, HL, 1>(xn, args, stream);
      case 2: return launch_edge_x<ATT, HL, 2>(xn, args, stream);
      case 4: return launch_edge_x<ATT, HL, 4>(xn, args, stream);
      case 8: return launch_edge_x<ATT, HL, 8>(xn, args, stream);
      default: return fail(-2, "edge_fwd att=3/4: F_out must be QN*(64/H)*4 with QN in {1,2,4,8}");
    }
  } else {
    return launch_edge_x<ATT, HL, 1>(xn, args, stream);
  }
}

template <int ATT>
static int launch_edge_h(int hl, int qn, int xn, const EdgeFwdArgs& args, hipStream_t stream) {
  switch (hl) {
    case 1: return launch_edge_q<ATT, 1>(qn, xn, args, stream);
    case 2: return launch_edge_q<ATT, 2>(qn, xn, args, stream);
    case 3: return launch_edge_q<ATT, 3>(qn, xn, args, stream);
    case 4: return launch_edge_q<ATT, 4>(qn, xn, args, stream);
    default: return fail(-2, "edge_fwd: H must be 2, 4, 8 or 16 (pad heads on the host)");
  }
}

}  // namespace disgat

extern "C" int disgat_edge_fwd(int att, const int32_t* items, int n_items, const int32_t* col, int64_t E, int N, int H,
                               int F_in, int F_out, const float* x, int ldx, const float* rowop, int ld_row,
                               const float* colop, int ld_col, const float* a, float* Z, float* edge_e, float* den,
                               float* part_z, float* part_den, int sage_div, float drop_p, uint64_t drop_seed,
                               const uint64_t* drop_seed_dev, uint32_t* sign_bits, const float* e_in, uint16_t* Z_hi, uint16_t* Z_lo, const float* z_bound,
                               disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(att >= 1 && att <= 4, "edge_fwd: att=%d not in 1..3 (4 = att 2 over projected operands)", att);
  DISGAT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "edge_fwd: dropout p=%f outside [0,1)", (double)drop_p);
  DISGAT_REQUIRE(n_items >= 0 && N > 0 && E >= 0, "edge_fwd: bad sizes n_items=%d N=%d E=%lld", n_items, N, (long long)E);
  if (n_items == 0) return 0;
  const int hl = ilog2_exact(H);
  DISGAT_REQUIRE(hl >= 1 && hl <= 4, "edge_fwd: H=%d must be a power of two in [2,16]", H);
  DISGAT_REQUIRE(F_in > 0 && F_in % 4 == 0 && ldx % 4 == 0 && ldx >= F_in, "edge_fwd: F_in=%d ldx=%d must be multiples of 4", F_in, ldx);
  DISGAT_REQUIRE(items && x && rowop && (Z || Z_hi) && (E == 0 || (col && edge_e)), "edge_fwd: null pointer");   // E == 0: rows are only zero-filled
  DISGAT_REQUIRE(aligned16(items) && aligned16(x) && aligned16(Z), "edge_fwd: items/x/Z must be 16-byte aligned");
  DISGAT_REQUIRE((Z_hi == nullptr) == (Z_lo == nullptr) && (Z_hi == nullptr || (z_bound != nullptr && aligned16(Z_hi) && aligned16(Z_lo))),
                 "edge_fwd: Z_hi, Z_lo (16-byte aligned) and z_bound go together");
  const int xn = (F_in + 255) / 256;
  int qn = 1;
  if (att == 3 || att == 4) {
    const int g4 = (64 >> hl) * 4;
    DISGAT_REQUIRE(F_out > 0 && F_out % g4 == 0, "edge_fwd att=3: F_out=%d must be a multiple of %d", F_out, g4);
    qn = F_out / g4;
    DISGAT_REQUIRE(colop && (a || att == 4) && ld_row % 4 == 0 && ld_col % 4 == 0 && ld_row >= H * F_out && ld_col >= H * F_out,
                   "edge_fwd att=3: bad P/Q strides");
    DISGAT_REQUIRE(aligned16(rowop) && aligned16(colop) && (att == 4 || aligned16(a)), "edge_fwd att=3: P/Q/a must be 16-byte aligned");
  } else if (att == 2) {
    DISGAT_REQUIRE(ld_row % 4 == 0 && ld_row >= H * F_in && aligned16(rowop), "edge_fwd att=2: bad P stride/alignment");
  } else {
    DISGAT_REQUIRE(colop && ld_row >= H && ld_col >= H, "edge_fwd att=1: bad s1/s2");
  }
  EdgeFwdArgs args{reinterpret_cast<const int4*>(items), n_items, col, E, N, F_in, x, ldx, rowop, ld_row, colop, ld_col,
                   a, Z, edge_e, den, part_z, part_den, sage_div,
                   DropCfg{drop_seed, (uint32_t)((double)drop_p * 4294967296.0), 1.0f / (1.0f - drop_p), drop_seed_dev},
                   att == 3 ? sign_bits : nullptr, e_in, Z_hi, Z_lo, z_bound};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (att) {
    case 1: return launch_edge_h<1>(hl, qn, xn, args, s);
    case 2: return launch_edge_h<2>(hl, qn, xn, args, s);
    case 4: return launch_edge_h<4>(hl, qn, xn, args, s);
    default: return launch_edge_h<3>(hl, qn, xn, args, s);
  }
}

extern "C" int disgat_edge_combine(const int32_t* split_rows, const int32_t* split_ptr, int n_split, int H, int F_in,
                                   const float* part_z, const float* part_den, float* Z, float* den, int sage_div,
                                   uint16_t* Z_hi, uint16_t* Z_lo, const float* z_bound, disgat_stream_t stream) {
  using namespace disgat;
  if (n_split == 0) return 0;
  DISGAT_REQUIRE(n_split > 0 && H > 0 && F_in > 0 && F_in % 4 == 0, "edge_combine: bad sizes");
  DISGAT_REQUIRE(split_rows && split_ptr && part_z && part_den && (Z || (Z_hi && Z_lo && z_bound)), "edge_combine: null pointer");
  hipLaunchKernelGGL(edge_combine_kernel, dim3(n_split), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), split_rows,
                     split_ptr, H, F_in, part_z, part_den, Z, den, sage_div, Z_hi, Z_lo, z_bound);
  return check_launch("edge_combine_kernel");
}
