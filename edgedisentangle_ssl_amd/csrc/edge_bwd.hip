// Backward of the fused DISGAT edge pass and of the aux-pair scorer for gfx950.
//
// The reference gets these gradients from ATen autograd through index / cat / mm /
// scatter_add_ (loss.backward() at pretrainer.py:752, 631, 836; trainer.py:200).  Here they are
// gather-only segment passes over work items (no float atomics except for the few split hub
// segments), reusing the forward's lane maps:
//
//  bwd_alpha_kernel     per CSR row r, head h, with s = sigmoid(e), w = exp(s), alpha = w/den:
//                         galpha_k = sc * <gZ[r,h,:], x[col_k,:]>          (Z = sc * sum_k alpha_k x[col_k])
//                         t        = sum_k alpha_k galpha_k = <gZ[r,h,:], Z[r,h,:]>   (no second sweep)
//                         ge_k     = ge_in_k + alpha_k (galpha_k - t) * s_k (1 - s_k)
//                       writes ge[H][E] and beta[H][E] = alpha*sc (coefficients of the transposed pass).
//  seg_grad_att3_kernel e = sum_f a_f lrelu(P[key] + Q[other]):  gkey[key] = sum_m g_m a lrelu'(z_m),
//                       ga += sum_m g_m lrelu(z_m).  Symmetric in P and Q, so the same kernel gives gP
//                       (segments = CSR rows / row-sorted aux pairs) and gQ (segments = CSC columns /
//                       column-sorted aux pairs, g read through a permutation).
//  seg_grad_hx_row      gkey[key,h,:] = sum_m coef[h,m] X[other_m,:]        (att 2: gP)
//  seg_grad_hx_col      gkey[key,:] (+)= sum_m sum_h coef[h,m] Mx[other_m,h,:]   (gx from beta*gZ; att 2: ge*P)
//
// Segments longer than the chunk are split over several work items (slot >= 0); those add into
// rows the host has zeroed, everything else is a plain store by the single owner of the row.
#include "disgat_common.h"

namespace disgat {

// Where a work item's result goes: a whole key's row of gkey, or - for a key split across several work items (hub
// rows / columns) - that slice's own partial record part[slot] (row layout of gkey), which seg_combine_kernel adds in
// slice order afterwards (deterministic).  part == null keeps the older behaviour: atomic adds into a zeroed row.
__device__ __forceinline__ float* seg_out_row(float* gkey, float* part, int ld, int key, int slot) {
  return (slot >= 0 && part != nullptr) ? part + (size_t)slot * ld : gkey + (size_t)key * ld;
}

__device__ __forceinline__ void out4(float* p, f32x4 v, bool atomic, bool accumulate) {
  if (atomic) {
    atomicAdd(p + 0, v.x);
    atomicAdd(p + 1, v.y);
    atomicAdd(p + 2, v.z);
    atomicAdd(p + 3, v.w);
  } else if (accumulate) {
    st4(p, ld4(p) + v);
  } else {
    st4(p, v);
  }
}

// max |v| bookkeeping for the buffers the segment passes produce (the f16x3 GEMMs that consume them need max |.| as their
// scale input; measuring it in the producer saves a pass over the 8 GB buffer).  Bit patterns of non-negative floats order
// like unsigned integers.
__device__ __forceinline__ uint32_t amax4(uint32_t m, const f32x4 v) {
  return max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
}
__device__ __forceinline__ void wave_atomic_max(uint32_t m, uint32_t* out) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

// ------------------------------------------------------------------------------------------
struct BwdAlphaArgs {
  const int4* items;
  int n_items;
  const int32_t* col;
  int64_t E;
  int F_in;
  const float* x;
  int ldx;
  const float* gZ;      // [N][H][F_in]
  const float* Z;       // [N][H][F_in]
  const float* edge_e;  // [H][E]
  const float* den;     // [N][2][H]: softmax denominators, dropped-out numerator sums
  const float* ge_in;   // [H][E] or null
  float* ge_out;        // [H][E], or [E][H] with ge_t (what the sign-record segment passes gather per position)
  int ge_t;
  float* beta;          // [H][E]
  int sage_div;
  DropCfg drop;
};

template <int HL, int XN>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void bwd_alpha_kernel(const BwdAlphaArgs A) {
  constexpr int H = 1 << HL;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= A.n_items) return;
  const int4 it = A.items[item];
  const int row = rfl(it.x), kb = rfl(it.y), ke = rfl(it.z);
  if (kb >= ke) return;
  const int myh = lane & (H - 1);
  const int xoff = lane * 4;

  f32x4 gz[H * XN];
  float tsel = 0.f;
  {
    const float* gp = A.gZ + (size_t)row * (H * A.F_in);
    const float* zp = A.Z + (size_t)row * (H * A.F_in);
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int o = i * 256 + xoff;
        const bool ok = o < A.F_in;
        gz[hh * XN + i] = ok ? ld4(gp + hh * A.F_in + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 zz = ok ? ld4(zp + hh * A.F_in + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc = dot4(gz[hh * XN + i], zz, acc);
      }
      acc = group_sum<6>(acc);
      tsel = (myh == hh) ? acc : tsel;
    }
  }
  const float den = A.den[(size_t)row * (2 * H) + myh];
  const float dsum = A.den[(size_t)row * (2 * H) + H + myh];
  const float inv = (den > 0.f) ? 1.0f / den : 0.f;
  const float scs = A.sage_div ? 1.0f / (dsum * inv + 1.0f) : 1.0f;

  f32x4 xA[XN], xB[XN];
  auto load_x = [&](f32x4(&xv)[XN], int c) {
    const float* xp = A.x + (size_t)c * A.ldx + xoff;
#pragma unroll
    for (int i = 0; i < XN; ++i) xv[i] = (i * 256 + xoff < A.F_in) ? ld4(xp + i * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // Per 64-edge batch the per-(edge, head) scalars are handled in a batch layout - lane (g', h) with
  // g' = lane / H owns batch positions i with i % GE == g' - so that edge_e / ge_in are read and
  // ge_out / beta written as GE-float runs instead of 4-byte scattered accesses.
  constexpr int GE = 64 >> HL;
  const int ge = lane >> HL;
  float galk[H];                                     // galpha of the positions this lane owns
  auto compute = [&](const f32x4(&xv)[XN], int i) {
    // the H dot products <gZ[r,h,:], x[c,:]> are reduced together by the transposed butterfly (head h's total lands
    // in lane group h), then each lane fetches the total of ITS head (lane % H): ~12 cross-lane ops instead of H
    // full 64-lane reductions
    float part[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < XN; ++t) acc = dot4(gz[hh * XN + t], xv[t], acc);
      part[hh] = acc;
    }
    const float gal = __shfl(multi_reduce<HL>(part), myh << (6 - HL), 64);
    const bool mine = (i & (GE - 1)) == ge;
    const int slot = i / GE;
#pragma unroll
    for (int t = 0; t < H; ++t) galk[t] = (mine && slot == t) ? gal : galk[t];
  };
  for (int kbase = kb; kbase < ke; kbase += 64) {
    const int cnt = min(64, ke - kbase);
    const int cv = (lane < cnt) ? A.col[kbase + lane] : 0;
    load_x(xA, __builtin_amdgcn_readlane(cv, 0));
    int i = 0;
    for (; i + 1 < cnt; i += 2) {
      load_x(xB, __builtin_amdgcn_readlane(cv, i + 1));
      compute(xA, i);
      if (i + 2 < cnt) load_x(xA, __builtin_amdgcn_readlane(cv, i + 2));
      compute(xB, i + 1);
    }
    if (i < cnt) compute(xA, i);
#pragma unroll
    for (int t = 0; t < H; ++t) {
      const int idx = t * GE + ge;
      if (idx < cnt) {
        const int64_t k = (int64_t)kbase + idx;
        const int64_t o = (int64_t)myh * A.E + k;
        const float e = A.edge_e[o];
        const float s = sigmoidf_(e);
        const float alpha = expf(s) * inv;
        const float cf = scs * drop_mult(A.drop, k, myh, H);   // Z = sum_k alpha_k * cf_k * x[col_k]
        const float gs = alpha * (cf * galk[t] - tsel);
        float g = gs * s * (1.0f - s);
        if (A.ge_in != nullptr) g += A.ge_in[o];
        A.ge_out[A.ge_t ? k * H + myh : o] = g;      // [E][H]: lane (g', h) -> consecutive floats, one 256-byte run per t
        A.beta[o] = alpha * cf;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
struct SegArgs {
  const int4* items;     // {key, m_begin, m_end, slot}
  int n_items;
  const int32_t* other;  // [M] index of the gathered operand per list position
  const int32_t* perm;   // [M] or null: position in g/coef of list position m
  const float* g;        // [H][g_stride]
  int64_t g_stride;
  int h_lo, h_hi;
  int F;                 // hx kernels: floats per (node, head) row (multiple of 4)
  const float* keyop;
  int ld_key;
  const float* otherop;
  int ld_other;
  const float* a;        // [H*FQ], or null: gkey receives u itself instead of a (.) u
  float* gkey;
  int ld_gkey;
  float* ga_part;        // [n_waves][H*FQ] or null
  int accumulate;
  float* part;           // [n_slots][ld_gkey] partial records of split keys, or null (atomics)
};

// DOT (att 4: e = <keyop[key], otherop[other]> per head, no `a`): gkey[key] = sum_m g_m * otherop[other_m].
template <int HL, int QN, bool DOT = false>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void seg_grad_att3_kernel(const SegArgs A) {
  constexpr int GL = 6 - HL;
  constexpr int G = 1 << GL;
  constexpr int FQ = QN * G * 4;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * DISGAT_WAVES_PER_BLOCK;
  const int myh = lane >> GL;
  const bool active = (myh >= A.h_lo) && (myh < A.h_hi);
  const int qoff = myh * FQ + (lane & (G - 1)) * 4;

  f32x4 a_r[QN], ga[QN];
#pragma unroll
  for (int j = 0; j < QN; ++j) {
    a_r[j] = DOT ? f32x4{0.f, 0.f, 0.f, 0.f} : ld4(A.a + qoff + j * G * 4);
    ga[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int item = wave; item < A.n_items; item += n_waves) {   // persistent: ga stays in registers
    const int4 it = A.items[item];
    const int key = rfl(it.x), mb = rfl(it.y), me = rfl(it.z), slot = rfl(it.w);
    if (key < 0) continue;          // padding item of a fixed-capacity item table (captured steps)
    f32x4 p_r[QN], gk[QN], qA[QN], qB[QN];
    {
      const float* pp = A.keyop + (size_t)key * A.ld_key + qoff;
#pragma unroll
      for (int j = 0; j < QN; ++j) {
        p_r[j] = active ? ld4(pp + j * G * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        gk[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    auto load_q = [&](f32x4(&q)[QN], int o) {
      if (active) {
        const float* qp = A.otherop + (size_t)o * A.ld_other + qoff;
#pragma unroll
        for (int j = 0; j < QN; ++j) q[j] = ld4(qp + j * G * 4);
      }
    };
    auto compute = [&](const f32x4(&q)[QN], int gpos) {
      if (active) {
        const float gv = A.g[(int64_t)myh * A.g_stride + gpos];
        if constexpr (DOT) {
#pragma unroll
          for (int j = 0; j < QN; ++j) gk[j] += gv * q[j];
          return;
        }
#pragma unroll
        for (int j = 0; j < QN; ++j) {
          const f32x4 z = p_r[j] + q[j];
          f32x4 l, d;
          l.x = lrelu001(z.x); l.y = lrelu001(z.y); l.z = lrelu001(z.z); l.w = lrelu001(z.w);
          d.x = z.x > 0.f ? 1.f : 0.01f; d.y = z.y > 0.f ? 1.f : 0.01f;
          d.z = z.z > 0.f ? 1.f : 0.01f; d.w = z.w > 0.f ? 1.f : 0.01f;
          ga[j] += gv * l;
          gk[j] += (gv * a_r[j]) * d;
        }
      }
    };
    for (int mbase = mb; mbase < me; mbase += 64) {
      const int cnt = min(64, me - mbase);
      const int ov = (lane < cnt) ? A.other[mbase + lane] : 0;
      const int pv = (lane < cnt) ? (A.perm ? A.perm[mbase + lane] : mbase + lane) : 0;
      load_q(qA, __builtin_amdgcn_readlane(ov, 0));
      int i = 0;
      for (; i + 1 < cnt; i += 2) {
        load_q(qB, __builtin_amdgcn_readlane(ov, i + 1));
        compute(qA, __builtin_amdgcn_readlane(pv, i));
        if (i + 2 < cnt) load_q(qA, __builtin_amdgcn_readlane(ov, i + 2));
        compute(qB, __builtin_amdgcn_readlane(pv, i + 1));
      }
      if (i < cnt) compute(qA, __builtin_amdgcn_readlane(pv, i));
    }
    float* op = seg_out_row(A.gkey, A.part, A.ld_gkey, key, slot) + qoff;
    const bool to_part = slot >= 0 && A.part != nullptr;
#pragma unroll
    for (int j = 0; j < QN; ++j) out4(op + j * G * 4, gk[j], slot >= 0 && !to_part, A.accumulate != 0 && !to_part);
  }
  if (A.ga_part != nullptr) {
    float* gp = A.ga_part + (size_t)wave * (FQ << HL) + qoff;
#pragma unroll
    for (int j = 0; j < QN; ++j) st4(gp + j * G * 4, ga[j]);
  }
}

// ------------------------------------------------------------------------------------------
// Gather-free att-3 score backward.  leaky_relu' takes two values, so with the forward's sign record
// (one bit per (edge | pair, feature): z > 0) the gradient of e = sum_f a_f lrelu(P[key] + Q[other]) is
//     u[key][f]  = sum_m g_m (z_mf > 0 ? 1 : 0.01) = 0.01 * sum_m g_m + 0.99 * sum_{m: bit} g_m
//     gkey[key]  = a (.) u[key]
//     ga         = sum_m g_m lrelu(z_m) = sum_keys keyop[key] (.) u[key]  +  the same sum of the other side's pass
// (lrelu(z) = lrelu'(z) z and z = P + Q, so the a-gradient splits into two node-level sums).
// Per list position this reads one 64-word sign row and H upstream gradients instead of the other side's
// 4*H*F_out-byte operand row: 288 B instead of 8 KB at H=8, F_out=256.  Lane map as edge_fwd_kernel<3>.
struct SignArgs {
  const int4* items;     // {key, m_begin, m_end, slot} over a list sorted by key
  int n_items;
  const int32_t* perm;   // [M] or null: row of g / sign belonging to list position m
  const float* g;        // upstream gradient of the raw scores: head h, position p at g[h * g_stride + p * g_pstride]
  int64_t g_stride, g_pstride;
  int h_lo, h_hi;
  const uint32_t* sign;  // [M][64] sign words (disgat_common.h)
  const float* keyop;    // P (row pass) or Q (column pass); read only when ga_part != null
  int ld_key;
  const float* a;        // [H*FQ], or null: gkey receives u itself instead of a (.) u
  float* gkey;
  int ld_gkey;
  float* ga_part;        // [n_waves][H*FQ] or null
  int accumulate;        // 1: add into gkey (heads outside [h_lo, h_hi) are left untouched) instead of storing
  float* part;           // [n_slots][ld_gkey] partial records of split keys, or null (atomics)
  uint32_t* amax_out;    // or null: raised to max |value stored into gkey| (whole keys; split keys: disgat_seg_combine)
};

template <int HL, int QN>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void seg_grad_sign_kernel(const SignArgs A) {
  constexpr int GL = 6 - HL;
  constexpr int G = 1 << GL;
  constexpr int FQ = QN * G * 4;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * DISGAT_WAVES_PER_BLOCK;
  const int myh = lane >> GL;
  const bool active = (myh >= A.h_lo) && (myh < A.h_hi);
  const int qoff = myh * FQ + (lane & (G - 1)) * 4;
  const uint32_t* sg = A.sign + lane;
  const float* gh = A.g + (int64_t)myh * A.g_stride;
  const bool want_ga = A.ga_part != nullptr;

  f32x4 ga[QN];
#pragma unroll
  for (int j = 0; j < QN; ++j) ga[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t mx = 0u;
  const float k99 = 0.99f / sign_unit();       // the record bits arrive as 0 / sign_unit() (a power of two): undone here

  for (int item = wave; item < A.n_items; item += n_waves) {   // persistent: ga stays in registers
    const int4 it = A.items[item];
    const int key = rfl(it.x), mb = rfl(it.y), me = rfl(it.z), slot = rfl(it.w);
    if (key < 0) continue;          // padding item of a fixed-capacity item table (captured steps)
    f32x4 up[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) up[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gall = 0.f;
    for (int mbase = mb; mbase < me; mbase += 64) {
      const int cnt = min(64, me - mbase);
      const int pv = (lane < cnt) ? (A.perm ? A.perm[mbase + lane] : mbase + lane) : 0;
      // groups of GS positions, the next group's loads in flight while the current one is accumulated (2-deep);
      // lanes >= cnt hold position 0 (valid memory), their gradient is masked.  Narrow heads (QN <= 2: the nhid = 64 layers
      // of the bundled graphs, whose 32-entry items are nothing but a chain of memory round trips) have the registers for
      // groups of 8 - 16 positions in flight; the wide heads of the large workloads stay at 4
      constexpr int GS = QN <= 2 ? 8 : 4;
      auto fetch = [&](uint32_t(&w)[GS], float(&gv)[GS], int i) {
#pragma unroll
        for (int t = 0; t < GS; ++t) {
          const int64_t pos = __builtin_amdgcn_readlane(pv, (i + t) & 63);
          w[t] = sg[pos * 64];
          gv[t] = (active && i + t < cnt) ? gh[pos * A.g_pstride] : 0.f;
        }
      };
      auto accum = [&](const uint32_t(&w)[GS], const float(&gv)[GS]) {
#pragma unroll
        for (int t = 0; t < GS; ++t) {
          __builtin_amdgcn_sched_barrier(0);      // one position at a time: keeps the 32 bit->float temporaries from piling up
          gall += gv[t];
#pragma unroll
          for (int p = 0; p < (QN >= 2 ? QN / 2 : 1); ++p) {      // per group pair: shift, and, 4 packed fp4 converts, 4 v_pk_fma_f32
            f32x4 ev, od;
            sign_floats2<QN>(w[t], p, ev, od);
            up[2 * p] += gv[t] * ev;
            if (QN >= 2) up[2 * p + 1] += gv[t] * od;
          }
        }
      };
      uint32_t wA[GS], wB[GS];
      float gA[GS], gB[GS];
      fetch(wA, gA, 0);
      for (int i = 0; i < cnt; i += 2 * GS) {
        if (i + GS < cnt) fetch(wB, gB, i + GS);
        accum(wA, gA);
        if (i + 2 * GS < cnt) fetch(wA, gA, i + 2 * GS);
        if (i + GS < cnt) accum(wB, gB);
      }
    }
    float* op = seg_out_row(A.gkey, A.part, A.ld_gkey, key, slot) + qoff;
    const bool to_part = slot >= 0 && A.part != nullptr;      // a partial record is always stored whole (zeros for idle heads)
    const float base = 0.01f * gall;
    const float* pp = A.keyop + (size_t)key * A.ld_key + qoff;
#pragma unroll
    for (int j = 0; j < QN; ++j) {           // a (8 KB, cache-resident) and the key's operand row are read here, once per item
      const f32x4 u = k99 * up[j] + base;
      if (want_ga) ga[j] += ld4(pp + j * G * 4) * u;
      if (to_part || !A.accumulate || active) {
        const f32x4 val = A.a ? ld4(A.a + qoff + j * G * 4) * u : u;
        if (A.amax_out != nullptr && slot < 0) {       // whole key: the value that ends up in gkey is known here
          float* q = op + j * G * 4;
          const f32x4 fin = A.accumulate ? ld4(q) + val : val;
          st4(q, fin);
          mx = amax4(mx, fin);
        } else {
          out4(op + j * G * 4, val, slot >= 0 && !to_part, A.accumulate != 0 && !to_part);
        }
      }
    }
  }
  if (A.amax_out != nullptr) wave_atomic_max(mx, A.amax_out);     // once per wave of a persistent launch
  if (want_ga) {
    float* gp = A.ga_part + (size_t)wave * (FQ << HL) + qoff;
#pragma unroll
    for (int j = 0; j < QN; ++j) st4(gp + j * G * 4, ga[j]);
  }
}

// gkey[key][h][:] = sum_m coef[h][pos(m)] * X[other_m][:]
template <int HL, int XN>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void seg_grad_hx_row_kernel(const SegArgs A) {
  constexpr int H = 1 << HL;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= A.n_items) return;
  const int4 it = A.items[item];
  const int key = rfl(it.x), mb = rfl(it.y), me = rfl(it.z), slot = rfl(it.w);
  if (key < 0) return;              // padding item of a fixed-capacity item table (captured steps)
  const int xoff = lane * 4;
  const int myh = lane & (H - 1);
  const bool hact = (myh >= A.h_lo) && (myh < A.h_hi);
  f32x4 acc[H * XN], xA[XN], xB[XN];
  float cA, cB;      // the position's coefficients travel with its operand row through the 2-deep pipeline (loaded in
                     // compute() their latency was exposed once per position)
#pragma unroll
  for (int i = 0; i < H * XN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](f32x4(&xv)[XN], float& cf, int o, int gpos) {
    const float* xp = A.otherop + (size_t)o * A.ld_other + xoff;
#pragma unroll
    for (int i = 0; i < XN; ++i) xv[i] = (i * 256 + xoff < A.F) ? ld4(xp + i * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
    cf = (lane < H && hact) ? A.g[(int64_t)myh * A.g_stride + gpos] : 0.f;
  };
  auto compute = [&](const f32x4(&xv)[XN], const float cf) {
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float c = readlane_f(cf, hh);
#pragma unroll
      for (int i = 0; i < XN; ++i) acc[hh * XN + i] += c * xv[i];
    }
  };
  for (int mbase = mb; mbase < me; mbase += 64) {
    const int cnt = min(64, me - mbase);
    const int ov = (lane < cnt) ? A.other[mbase + lane] : 0;
    const int pv = (lane < cnt) ? (A.perm ? A.perm[mbase + lane] : mbase + lane) : 0;
    load_x(xA, cA, __builtin_amdgcn_readlane(ov, 0), __builtin_amdgcn_readlane(pv, 0));
    int i = 0;
    for (; i + 1 < cnt; i += 2) {
      load_x(xB, cB, __builtin_amdgcn_readlane(ov, i + 1), __builtin_amdgcn_readlane(pv, i + 1));
      compute(xA, cA);
      if (i + 2 < cnt) load_x(xA, cA, __builtin_amdgcn_readlane(ov, i + 2), __builtin_amdgcn_readlane(pv, i + 2));
      compute(xB, cB);
    }
    if (i < cnt) compute(xA, cA);
  }
  float* op = seg_out_row(A.gkey, A.part, A.ld_gkey, key, slot) + xoff;
  const bool to_part = slot >= 0 && A.part != nullptr;
#pragma unroll
  for (int hh = 0; hh < H; ++hh)
#pragma unroll
    for (int i = 0; i < XN; ++i)
      if (i * 256 + xoff < A.F) out4(op + hh * A.F + i * 256, acc[hh * XN + i], slot >= 0 && !to_part, A.accumulate != 0 && !to_part);
}

// gkey[key][:] (+)= sum_m sum_h coef[h][pos(m)] * Mx[other_m][h][:]
template <int HL, int XN>
__global__ __launch_bounds__(DISGAT_BLOCK, 2) void seg_grad_hx_col_kernel(const SegArgs A) {
  constexpr int H = 1 << HL;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= A.n_items) return;
  const int4 it = A.items[item];
  const int key = rfl(it.x), mb = rfl(it.y), me = rfl(it.z), slot = rfl(it.w);
  if (key < 0) return;              // padding item of a fixed-capacity item table (captured steps)
  const int xoff = lane * 4;
  const int myh = lane & (H - 1);
  const bool hact = (myh >= A.h_lo) && (myh < A.h_hi);
  f32x4 acc[XN], mA[H * XN], mB[H * XN];
  float cA, cB;      // coefficients prefetched with the operand row (see seg_grad_hx_row_kernel)
#pragma unroll
  for (int i = 0; i < XN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_m = [&](f32x4(&mv)[H * XN], float& cf, int o, int gpos) {
    cf = (lane < H && hact) ? A.g[(int64_t)myh * A.g_stride + gpos] : 0.f;
    const float* mp = A.otherop + (size_t)o * A.ld_other + xoff;
    // every head's slice is loaded (heads outside [h_lo, h_hi) get a zero coefficient): testing the range per head put
    // a branch and a full vmcnt(0) wait around each load
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
      for (int i = 0; i < XN; ++i)
        mv[hh * XN + i] = (i * 256 + xoff < A.F) ? ld4(mp + hh * A.F + i * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto compute = [&](const f32x4(&mv)[H * XN], const float cf) {
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float c = readlane_f(cf, hh);
#pragma unroll
      for (int i = 0; i < XN; ++i) acc[i] += c * mv[hh * XN + i];
    }
  };
  for (int mbase = mb; mbase < me; mbase += 64) {
    const int cnt = min(64, me - mbase);
    const int ov = (lane < cnt) ? A.other[mbase + lane] : 0;
    const int pv = (lane < cnt) ? (A.perm ? A.perm[mbase + lane] : mbase + lane) : 0;
    load_m(mA, cA, __builtin_amdgcn_readlane(ov, 0), __builtin_amdgcn_readlane(pv, 0));
    int i = 0;
    for (; i + 1 < cnt; i += 2) {
      load_m(mB, cB, __builtin_amdgcn_readlane(ov, i + 1), __builtin_amdgcn_readlane(pv, i + 1));
      compute(mA, cA);
      if (i + 2 < cnt) load_m(mA, cA, __builtin_amdgcn_readlane(ov, i + 2), __builtin_amdgcn_readlane(pv, i + 2));
      compute(mB, cB);
    }
    if (i < cnt) compute(mA, cA);
  }
  float* op = seg_out_row(A.gkey, A.part, A.ld_gkey, key, slot) + xoff;
  const bool to_part = slot >= 0 && A.part != nullptr;
#pragma unroll
  for (int i = 0; i < XN; ++i)
    if (i * 256 + xoff < A.F) out4(op + i * 256, acc[i], slot >= 0 && !to_part, A.accumulate != 0 && !to_part);
}

// att 1 (e = s1[row] + s2[col], layers.py:349-353): the score operands' gradients are plain segment sums of the score
// gradients, gkey[key][h] = sum over the key's list positions of g[h][perm(m)].  One wave per item, lane = (position slot,
// head); every lane adds its positions in list order, the slots of a head are combined by a symmetric butterfly: a fixed
// summation order, no float atomics (the reference's index_add_-style autograd is neither).
struct SumArgs {
  const int4* items;
  int n_items;
  const int32_t* perm;
  const float* g;
  int64_t g_stride, g_pos_stride;
  int h_lo, h_hi, H;
  float* gkey;
  int ld;
  int accumulate;
  float* part;
};
__global__ __launch_bounds__(DISGAT_BLOCK) void seg_sum_kernel(const SumArgs A) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * DISGAT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= A.n_items) return;
  const int4 it = A.items[item];
  const int key = rfl(it.x), mb = rfl(it.y), me = rfl(it.z), slot = rfl(it.w);
  if (key < 0) return;              // padding item of a fixed-capacity item table (captured steps)
  const int h = lane & (A.H - 1), p = lane / A.H, P = 64 / A.H;
  const bool hact = h >= A.h_lo && h < A.h_hi;
  float acc = 0.f;
  if (hact)
    for (int m = mb + p; m < me; m += P) {
      const int64_t pos = A.perm ? A.perm[m] : m;
      acc += A.g[(int64_t)h * A.g_stride + pos * A.g_pos_stride];
    }
  for (int o = 32; o >= A.H; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane < A.ld) {                // lanes H .. ld-1: the zero padding of rows narrower than 4 floats
    float* out = seg_out_row(A.gkey, A.part, A.ld, key, slot) + lane;
    const float v = lane < A.H ? acc : 0.f;
    *out = (A.accumulate && !(slot >= 0 && A.part != nullptr)) ? *out + v : v;
  }
}

// gkey[key][0:width] (+)= sum of the key's partial records, in slice order: one block per split key.
__global__ __launch_bounds__(256) void seg_combine_kernel(const int32_t* __restrict__ split_keys, const int32_t* __restrict__ split_ptr,
                                                          int width, const float* __restrict__ part, int ld, float* __restrict__ gkey,
                                                          int accumulate, uint32_t* __restrict__ amax_out) {
  const int key = split_keys[blockIdx.x];
  if (key < 0) return;              // padding entry of a fixed-capacity split table
  const int s0 = split_ptr[blockIdx.x], s1 = split_ptr[blockIdx.x + 1];
  float* out = gkey + (size_t)key * ld;
  uint32_t mx = 0u;
  for (int c = threadIdx.x * 4; c < width; c += 256 * 4) {
    f32x4 acc = accumulate ? ld4(out + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    // eight records in flight, added in slice order (a hub of a small graph has tens of slices: one load at a time the
    // loop was a chain of memory round trips - 13 us per launch on chameleon)
    int sl = s0;
    for (; sl + 8 <= s1; sl += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ld4(part + (size_t)(sl + j) * ld + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; sl < s1; ++sl) acc += ld4(part + (size_t)sl * ld + c);
    st4(out + c, acc);
    mx = amax4(mx, acc);
  }
  if (amax_out != nullptr) wave_atomic_max(mx, amax_out);
}

}  // namespace disgat

// ------------------------------------------------------------------------------------------
// C ABI
#include "disgat_api.h"

extern "C" int disgat_bwd_alpha(const int32_t* items, int n_items, const int32_t* col, int64_t E, int H, int F_in,
                                const float* x, int ldx, const float* gZ, const float* Z, const float* edge_e,
                                const float* den, const float* ge_in, float* ge_out, int ge_transposed, float* beta,
                                int sage_div, float drop_p, uint64_t drop_seed, const uint64_t* drop_seed_dev,
                                disgat_stream_t stream) {
  using namespace disgat;
  if (n_items == 0 || E == 0) return 0;
  const int hl = ilog2_exact(H);
  DISGAT_REQUIRE(hl >= 1 && hl <= 4, "bwd_alpha: H=%d must be a power of two in [2,16]", H);
  DISGAT_REQUIRE(F_in > 0 && F_in % 4 == 0 && ldx % 4 == 0, "bwd_alpha: F_in/ldx must be multiples of 4");
  DISGAT_REQUIRE(items && col && x && gZ && Z && edge_e && den && ge_out && beta, "bwd_alpha: null pointer");
  const int xn = (F_in + 255) / 256;
  DISGAT_REQUIRE(xn == 1 || (xn == 2 && hl <= 3), "bwd_alpha: F_in=%d too wide for H=%d", F_in, H);
  BwdAlphaArgs A{reinterpret_cast<const int4*>(items), n_items, col, E, F_in, x, ldx, gZ, Z, edge_e, den, ge_in,
                 ge_out, ge_transposed != 0, beta, sage_div,
                 DropCfg{drop_seed, (uint32_t)((double)drop_p * 4294967296.0), 1.0f / (1.0f - drop_p), drop_seed_dev}};
  const dim3 grid((n_items + DISGAT_WAVES_PER_BLOCK - 1) / DISGAT_WAVES_PER_BLOCK), block(DISGAT_BLOCK);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DISGAT_BA(HL_, XN_) hipLaunchKernelGGL((bwd_alpha_kernel<HL_, XN_>), grid, block, 0, s, A)
  if (xn == 1) {
    switch (hl) {
      case 1: DISGAT_BA(1, 1); break;
      case 2: DISGAT_BA(2, 1); break;
      case 3: DISGAT_BA(3, 1); break;
      default: DISGAT_BA(4, 1); break;
    }
  } else {
    switch (hl) {
      case 1: DISGAT_BA(1, 2); break;
      case 2: DISGAT_BA(2, 2); break;
      default: DISGAT_BA(3, 2); break;
    }
  }
#undef DISGAT_BA
  return check_launch("bwd_alpha_kernel");
}

static int seg_common_checks(const char* who, const int32_t* items, int n_items, const int32_t* other, const float* g,
                             int H, int h_lo, int h_hi, const float* otherop, float* gkey) {
  using namespace disgat;
  const int hl = ilog2_exact(H);
  DISGAT_REQUIRE(hl >= 1 && hl <= 4, "%s: H=%d must be a power of two in [2,16]", who, H);
  DISGAT_REQUIRE(h_lo >= 0 && h_hi <= H && h_lo < h_hi, "%s: bad head range [%d,%d)", who, h_lo, h_hi);
  DISGAT_REQUIRE(items && other && g && otherop && gkey && n_items > 0, "%s: null pointer", who);
  return 0;
}

extern "C" int disgat_seg_grad_att3(const int32_t* items, int n_items, const int32_t* other, const int32_t* perm,
                                    const float* g, int64_t g_stride, int h_lo, int h_hi, int H, int F_out,
                                    const float* keyop, int ld_key, const float* otherop, int ld_other, const float* a,
                                    float* gkey, int ld_gkey, float* ga_part, int n_waves, float* part,
                                    disgat_stream_t stream) {
  using namespace disgat;
  if (n_items == 0) return 0;
  if (int rc = seg_common_checks("seg_grad_att3", items, n_items, other, g, H, h_lo, h_hi, otherop, gkey)) return rc;
  const int hl = ilog2_exact(H);
  const int g4 = (64 >> hl) * 4;
  DISGAT_REQUIRE(F_out > 0 && F_out % g4 == 0 && keyop, "seg_grad_att3: bad F_out=%d", F_out);
  DISGAT_REQUIRE(a != nullptr || ga_part == nullptr, "seg_grad_att3: a == NULL selects the plain dot-product score (att 4), which has no `a` gradient");
  DISGAT_REQUIRE(n_waves > 0 && n_waves % DISGAT_WAVES_PER_BLOCK == 0, "seg_grad_att3: n_waves must be a positive multiple of %d", DISGAT_WAVES_PER_BLOCK);
  const int qn = F_out / g4;
  SegArgs A{reinterpret_cast<const int4*>(items), n_items, other, perm, g, g_stride, h_lo, h_hi, F_out, keyop, ld_key,
            otherop, ld_other, a, gkey, ld_gkey, ga_part, 0, part};
  const dim3 grid(n_waves / DISGAT_WAVES_PER_BLOCK), block(DISGAT_BLOCK);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DISGAT_SG(HL_, QN_)                                                                     \
  do {                                                                                          \
    if (a) hipLaunchKernelGGL((seg_grad_att3_kernel<HL_, QN_, false>), grid, block, 0, s, A);  \
    else hipLaunchKernelGGL((seg_grad_att3_kernel<HL_, QN_, true>), grid, block, 0, s, A);     \
  } while (0)
#define DISGAT_SGQ(HL_)                                                                       \
  switch (qn) {                                                                               \
    case 1: DISGAT_SG(HL_, 1); break;                                                         \
    case 2: DISGAT_SG(HL_, 2); break;                                                         \
    case 4: DISGAT_SG(HL_, 4); break;                                                         \
    case 8: DISGAT_SG(HL_, 8); break;                                                         \
    default: return fail(-2, "seg_grad_att3: F_out must be QN*(64/H)*4 with QN in {1,2,4,8}"); \
  }
  switch (hl) {
    case 1: DISGAT_SGQ(1); break;
    case 2: DISGAT_SGQ(2); break;
    case 3: DISGAT_SGQ(3); break;
    default: DISGAT_SGQ(4); break;
  }
#undef DISGAT_SGQ
#undef DISGAT_SG
  return check_launch("seg_grad_att3_kernel");
}

extern "C" int disgat_seg_grad_sign(const int32_t* items, int n_items, const int32_t* perm, const float* g,
                                    int64_t g_stride, int64_t g_pos_stride, int h_lo, int h_hi, int H, int F_out,
                                    const uint32_t* sign_bits,
                                    const float* keyop, int ld_key, const float* a, float* gkey, int ld_gkey,
                                    float* ga_part, int n_waves, int accumulate, float* part, float* amax_out,
                                    disgat_stream_t stream) {
  using namespace disgat;
  if (n_items == 0) return 0;
  const int hl = ilog2_exact(H);
  DISGAT_REQUIRE(hl >= 1 && hl <= 4, "seg_grad_sign: H=%d must be a power of two in [2,16]", H);
  DISGAT_REQUIRE(h_lo >= 0 && h_hi <= H && h_lo < h_hi, "seg_grad_sign: bad head range [%d,%d)", h_lo, h_hi);
  DISGAT_REQUIRE(items && g && sign_bits && gkey && n_items > 0, "seg_grad_sign: null pointer");
  DISGAT_REQUIRE(ga_part == nullptr || keyop != nullptr, "seg_grad_sign: ga_part needs keyop");
  const int g4 = (64 >> hl) * 4;
  DISGAT_REQUIRE(F_out > 0 && F_out % g4 == 0, "seg_grad_sign: bad F_out=%d", F_out);
  DISGAT_REQUIRE(ld_gkey % 4 == 0 && ld_key % 4 == 0 && (a == nullptr || aligned16(a)) && aligned16(gkey) && (keyop == nullptr || aligned16(keyop)),
                 "seg_grad_sign: strides must be multiples of 4 floats and bases 16-byte aligned");
  DISGAT_REQUIRE(n_waves > 0 && n_waves % DISGAT_WAVES_PER_BLOCK == 0, "seg_grad_sign: n_waves must be a positive multiple of %d", DISGAT_WAVES_PER_BLOCK);
  const int qn = F_out / g4;
  SignArgs A{reinterpret_cast<const int4*>(items), n_items, perm, g, g_stride, g_pos_stride, h_lo, h_hi, sign_bits, keyop, ld_key, a,
             gkey, ld_gkey, ga_part, accumulate, part, reinterpret_cast<uint32_t*>(amax_out)};
  const dim3 grid(n_waves / DISGAT_WAVES_PER_BLOCK), block(DISGAT_BLOCK);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DISGAT_SS(HL_, QN_) hipLaunchKernelGGL((seg_grad_sign_kernel<HL_, QN_>), grid, block, 0, s, A)
#define DISGAT_SSQ(HL_)                                                                        \
  switch (qn) {                                                                                \
    case 1: DISGAT_SS(HL_, 1); break;                                                          \
    case 2: DISGAT_SS(HL_, 2); break;                                                          \
    case 4: DISGAT_SS(HL_, 4); break;                                                          \
    case 8: DISGAT_SS(HL_, 8); break;                                                          \
    default: return fail(-2, "seg_grad_sign: F_out must be QN*(64/H)*4 with QN in {1,2,4,8}"); \
  }
  switch (hl) {
    case 1: DISGAT_SSQ(1); break;
    case 2: DISGAT_SSQ(2); break;
    case 3: DISGAT_SSQ(3); break;
    default: DISGAT_SSQ(4); break;
  }
#undef DISGAT_SSQ
#undef DISGAT_SS
  return check_launch("seg_grad_sign_kernel");
}

template <bool COL>
static int launch_hx(const disgat::SegArgs& A, int hl, int xn, hipStream_t s) {
  using namespace disgat;
  const dim3 grid((A.n_items + DISGAT_WAVES_PER_BLOCK - 1) / DISGAT_WAVES_PER_BLOCK), block(DISGAT_BLOCK);
#define DISGAT_HX(HL_, XN_)                                                                   \
  do {                                                                                        \
    if (COL) hipLaunchKernelGGL((seg_grad_hx_col_kernel<HL_, XN_>), grid, block, 0, s, A);    \
    else hipLaunchKernelGGL((seg_grad_hx_row_kernel<HL_, XN_>), grid, block, 0, s, A);        \
  } while (0)
  if (xn == 1) {
    switch (hl) {
      case 1: DISGAT_HX(1, 1); break;
      case 2: DISGAT_HX(2, 1); break;
      case 3: DISGAT_HX(3, 1); break;
      default: DISGAT_HX(4, 1); break;
    }
  } else if (xn == 2 && hl <= 3) {
    switch (hl) {
      case 1: DISGAT_HX(1, 2); break;
      case 2: DISGAT_HX(2, 2); break;
      default: DISGAT_HX(3, 2); break;
    }
  } else {
    return fail(-2, "seg_grad_hx: F=%d too wide for H=%d", A.F, 1 << hl);
  }
#undef DISGAT_HX
  return check_launch("seg_grad_hx_kernel");
}

extern "C" int disgat_seg_grad_hx(int col_mode, const int32_t* items, int n_items, const int32_t* other,
                                  const int32_t* perm, const float* coef, int64_t coef_stride, int h_lo, int h_hi, int H,
                                  int F, const float* otherop, int ld_other, float* gkey, int ld_gkey, int accumulate,
                                  float* part, disgat_stream_t stream) {
  using namespace disgat;
  if (n_items == 0) return 0;
  if (int rc = seg_common_checks("seg_grad_hx", items, n_items, other, coef, H, h_lo, h_hi, otherop, gkey)) return rc;
  DISGAT_REQUIRE(F > 0 && F % 4 == 0 && ld_other % 4 == 0 && ld_gkey % 4 == 0, "seg_grad_hx: F/strides must be multiples of 4");
  SegArgs A{reinterpret_cast<const int4*>(items), n_items, other, perm, coef, coef_stride, h_lo, h_hi, F, nullptr, 0,
            otherop, ld_other, nullptr, gkey, ld_gkey, nullptr, accumulate, part};
  const int hl = ilog2_exact(H);
  const int xn = (F + 255) / 256;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return col_mode ? launch_hx<true>(A, hl, xn, s) : launch_hx<false>(A, hl, xn, s);
}

extern "C" int disgat_seg_sum(const int32_t* items, int n_items, const int32_t* perm, const float* g, int64_t g_stride,
                              int64_t g_pos_stride, int h_lo, int h_hi, int H, float* gkey, int ld_gkey, int accumulate,
                              float* part, disgat_stream_t stream) {
  using namespace disgat;
  if (n_items == 0) return 0;
  const int hl = ilog2_exact(H);
  DISGAT_REQUIRE(hl >= 1 && hl <= 4, "seg_sum: H=%d must be a power of two in [2,16]", H);
  DISGAT_REQUIRE(items && g && gkey && h_lo >= 0 && h_hi <= H, "seg_sum: null pointer / head range");
  DISGAT_REQUIRE(ld_gkey >= H && ld_gkey <= 64 && ld_gkey % 4 == 0, "seg_sum: row stride %d (H <= stride <= 64, a multiple of 4)", ld_gkey);
  SumArgs A{reinterpret_cast<const int4*>(items), n_items, perm, g, g_stride, g_pos_stride, h_lo, h_hi, H, gkey, ld_gkey, accumulate, part};
  const dim3 grid((n_items + DISGAT_WAVES_PER_BLOCK - 1) / DISGAT_WAVES_PER_BLOCK), block(DISGAT_BLOCK);
  hipLaunchKernelGGL(seg_sum_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), A);
  return check_launch("seg_sum_kernel");
}

extern "C" int disgat_seg_combine(const int32_t* split_keys, const int32_t* split_ptr, int n_split, int width,
                                  const float* part, float* gkey, int ld_gkey, int accumulate, float* amax_out,
                                  disgat_stream_t stream) {
  using namespace disgat;
  if (n_split == 0) return 0;
  DISGAT_REQUIRE(split_keys && split_ptr && part && gkey, "seg_combine: null pointer");
  DISGAT_REQUIRE(width > 0 && width % 4 == 0 && width <= ld_gkey && ld_gkey % 4 == 0 && aligned16(part) && aligned16(gkey),
                 "seg_combine: width / stride must be multiples of 4 floats (width <= stride), bases 16-byte aligned");
  hipLaunchKernelGGL(seg_combine_kernel, dim3(n_split), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), split_keys,
                     split_ptr, width, part, ld_gkey, gkey, accumulate, reinterpret_cast<uint32_t*>(amax_out));
  return check_launch("seg_combine_kernel");
}
