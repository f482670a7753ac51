// f16x3 GEMM whose A operand arrives ALREADY SPLIT (two fp16 planes written by the kernel that produced it), streamed
// through an LDS ring by LDS-DMA.  gfx950 only.
//
//   C[b] = act( A[b] (M x K) * B[b] (K x N) + bias[b] + init[b] )
//   A[b]: planes Ah, Al [M][K] fp16 (row-major, k contiguous, row stride lda halfs) = hi, lo of A * s_A
//   B[b]: planes [2][N][K] fp16 (row-major) = hi, lo of B^T * s_B                  (disgat_split_f16_rm)
//   C   : fp32 and / or planes Ch, Cl (hi, lo of C * s_C) for the next GEMM of the chain
//
// Why (round 3): the consumers of the two big dense operands of a DISGAT layer - the aggregated neighbourhoods
// Z[N,H,F_in] (edge pass -> per-head projection, layers.py:397-399) and the concatenated heads [N,H*nhid]
// (projection -> FuseLayer, layers.py:905) - used to split fp32 -> (hi, lo) on the fly: 2.6-3.8 VALU instructions per
// MFMA, one block per CU, load -> split -> barrier -> MFMA -> store in series (20 us per 128-row tile of which 5 us
// MFMA).  Here the producers store the planes (same bytes as fp32) and this kernel does no operand arithmetic at all:
//   * persistent 512-thread blocks, one per CU, each walking a contiguous range of (head, row tile, column step) units
//     with the k-loop running on across units - loads of the next unit are in flight during a unit's epilogue;
//   * both operands reach LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write), 1 KB = 16 rows x 64 B per
//     wave instruction, XOR-swizzled through the per-lane SOURCE address (the LDS image of a DMA is lane-linear);
//   * waves 0-3 stream the A tiles (HBM, NSA ring slots: NSA-1 k-steps = 16 KB each in flight), waves 4-7 the weight
//     tiles (L2, NSB slots) - vmcnt counts a wave's loads, stores and DMAs in issue order, so a wave that had both
//     kinds in flight could not wait for the near weight tile without draining the far A tiles;
//   * counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per k-step; the epilogue's stores are counted too.
// 8 waves as 2 (M) x 4 (N), wave tile 64 x 64 = 4 x 4 MFMA 16x16x32 tiles, two fp32 accumulators per tile (hi*hi and
// the 2^-11-weighted cross terms), product accumulated transposed (weight fragment as the MFMA's A operand) and the
// weight rows of a 32-column group permuted on the way into LDS so that a lane ends up with 8 consecutive output
// columns of one row: one 16-byte store per plane (or two for fp32) per 16 x 32 block.
// Round 5, LG variant (disgat_gemm_planes_logits; DifHead's classifier): a second, skinny f16x3 product in the epilogue - the
// unit's activated [128 x 256] tile, still in the accumulators, is split and becomes the operand of `. W2` (256 -> n_out <= 16,
// one 16-wide MFMA tile); the four 64-column strips of a row meet in 32 KB of LDS behind the ring and are summed in strip order
// after the next k-step's barrier.  The [M x batch, 256] hidden layer is never written.
#include "gemm_common.h"
#include "disgat_api.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace disgat {

struct GemmPArgs {
  const uint16_t* Ah;
  const uint16_t* Al;
  int64_t lda, a_bs;       // halfs
  const uint16_t* Bt;      // [batch][2][N][K]
  const float* a_bound;    // device scalar the A planes were scaled by: s_A = f16_scale(*a_bound)
  const float* b_scale;    // device scalar s_B
  const float* bias;       // [batch][N] or null
  const float* init;       // or null
  int64_t ldi, i_bs;
  float* C;                // fp32 output or null
  int64_t ldc, c_bs;
  uint16_t* Ch;            // plane output or null
  uint16_t* Cl;
  int64_t ldp, p_bs;       // halfs
  const float* c_bound;    // device scalar: bound of |C| the output planes are scaled by
  int M, N, K, batch;
  float slope;
  int nsteps, units;       // column steps of 256; units = row tiles x nsteps x batch
  int dbg;                 // switches (DISGAT_PL_DEBUG): 64 = round 3's unit map; with DISGAT_PL_DIAG: 1 no stores, 2 no MFMA, 4 A rows from a cache-resident range
  // LG variant (disgat_gemm_planes_logits): a second, skinny product in the epilogue - L = act(.) W2 + b2, N = 256 -> n_out <= 16
  const uint16_t* W2f;     // fragment image of W2 * s_W2: [8 groups of 32 rows][2 planes][64 lanes][8 halfs]
  const float* w2_scale;   // device scalar s_W2
  const float* bias2;      // [16] (zero past n_out) or null
  float* L;                // [M * batch][n_out]: row (m, b) at (m * batch + b) * n_out
  int n_out;
};

constexpr int PL_BM = 128, PL_BN = 256;
constexpr int PL_A_SLOT = 2 * PL_BM * 64;     // bytes: 2 planes x 128 rows x 32 halfs
constexpr int PL_B_SLOT = 2 * PL_BN * 64;

#ifndef DISGAT_PL_DIAG
#define DISGAT_PL_DIAG 0      // 1: ablation switches (GemmPArgs::dbg) and s_memtime phase stamps compiled in (tools/gp_*.py)
#endif
__device__ unsigned long long pl_stamps[16];     // diagnostic (DISGAT_PL_DEBUG & 32): cycles per loop phase, [role][phase]

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const uint16_t* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
}
// LDS row rho (0..255) of a weight tile holds weight row n0 + (rho & ~31) + 8 ((rho & 15) >> 2) + 4 ((rho >> 4) & 1) + (rho & 3):
// within a group of 32 columns, {0-3, 8-11, 16-19, 24-27} come first, then {4-7, ...}; MFMA tiles 2g / 2g+1 then give a lane
// columns 8q..8q+3 / 8q+4..8q+7 of the group.  The permutation is applied when the weight planes are written
// (gemm_split.hip: wsplit_kernel, frag == 2), together with the chunk swizzle.

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int PL_LG_LDS = 4 * PL_BM * 16 * 4;     // LG: per column strip (wn) and row 16 partial outputs, fp32

template <int ACT, bool F32, bool PL, bool LG = false>
__global__ __launch_bounds__(512, 1) void gemm_planes_kernel(const GemmPArgs G) {
  static_assert(!LG || (!F32 && !PL), "the logits variant has no other output");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_pl[];
  constexpr int NSA = 3, NSB = 2;                           // ring slots: A two k-steps ahead, weights one
  constexpr int MC = 5, ML = 3;                             // 16-row tiles of a compute wave / of a loader wave
  constexpr int S_L = ML * ((F32 ? 4 : 0) + (PL ? 4 : 0));  // store instructions per epilogue of a loader wave
  static_assert(4 + S_L <= 63, "vmcnt is a 6-bit counter");
  const int lane = threadIdx.x & 63;
  const int wave = rfl(threadIdx.x >> 6);
  const int wn = wave & 3;
  const bool ld_wave = wave >= 4;
  const int wrow = ld_wave ? 16 * MC : 0;                   // first tile row of this wave's 64-column strip
  const int KT = G.K >> 5;

  // XCD-aware unit map.  Units are ordered HEAD-major and the blocks of one XCD (blockIdx.x % 8 under the round-robin
  // dispatch) take a contiguous eighth of them: with 8 heads every XCD works on ONE head's weight planes at a time (256 KB
  // in its 4 MB L2) instead of cycling through all 2 MB of them per row tile beside the streamed A tiles, which evicted
  // the weights about once in four (projection: 12.3 GB read for 8.2 GB of operand, profiles/r03/gemm_f16x3_pmc.md).  The
  // 8 XCDs still walk the same rows at the same time, so a row's [H][F_in] segments are read together.
  const bool xmap = gridDim.x % 8 == 0 && !(G.dbg & 64);    // DISGAT_PL_DEBUG=64: round 3's map (same-box A/B)
  const int vb = xmap ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ub = (int)((int64_t)G.units * vb / gridDim.x);
  const int ue = (int)((int64_t)G.units * (vb + 1) / gridDim.x);
  if (ub >= ue) return;
  const int per_head = G.units / G.batch;

  const float sA = f16_scale(*G.a_bound);
  const float sAB = sA * *G.b_scale;
  const float inv = 1.0f / sAB;
  const float xw = inv * (1.0f / 2048.f);
  float sC = 1.0f;
  if constexpr (PL || LG) sC = f16_scale(*G.c_bound);
  // LG: the unit's activated [128 x 256] tile is the B operand of a second f16x3 product with W2 (transposed form again: a
  // lane's 8 consecutive columns of one row ARE its 8 k-values of a 16x16x32 fragment).  Each wave's 64-column strip gives a
  // partial [rows x 16] result; the four strips meet in LDS and are summed, in strip order, after the next barrier.
  float* const lg_part = reinterpret_cast<float*>(lds_pl + NSA * PL_A_SLOT + NSB * PL_B_SLOT);
  int pend_m0 = -1, pend_bz = 0;
  auto lg_reduce = [&](int m0, int bz) __attribute__((always_inline)) {
    const int row = threadIdx.x >> 2, qq = threadIdx.x & 3;
    f32x4 s = *reinterpret_cast<const f32x4*>(lg_part + (0 * PL_BM + row) * 16 + 4 * qq);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(lg_part + (w * PL_BM + row) * 16 + 4 * qq);
      s = f32x4{s.x + p.x, s.y + p.y, s.z + p.z, s.w + p.w};
    }
    if (G.bias2 != nullptr) {
      const f32x4 b = ld4(G.bias2 + 4 * qq);
      s = f32x4{s.x + b.x, s.y + b.y, s.z + b.z, s.w + b.w};
    }
    if (m0 + row < G.M && 4 * qq < G.n_out) {
      float* lp = G.L + ((int64_t)(m0 + row) * G.batch + bz) * G.n_out + 4 * qq;
      if ((G.n_out & 3) == 0) {
        st4(lp, s);
      } else {
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * qq + r < G.n_out) lp[r] = v[r];
      }
    }
  };

  auto decode = [&](int u, int& m0, int& n0, int& bz) __attribute__((always_inline)) {
    int r;
    if (xmap) {
      bz = u / per_head;
      r = u - bz * per_head;
    } else {
      bz = u % G.batch;
      r = u / G.batch;
    }
    n0 = (r % G.nsteps) * PL_BN;
    m0 = (r / G.nsteps) * PL_BM;
  };

  // ---- loader state (waves 4-7).  Per k-step a loader wave w = wave & 3 issues 8 weight pieces (plane w >> 1, LDS rows
  // 128 (w & 1) + 16 j: 1 KB of consecutive memory each, the planes are stored as LDS images) and 4 A pieces (plane
  // w >> 1, rows 64 (w & 1) + 16 j; lane i fills LDS row i >> 2, 16-byte slot i & 3 from source chunk
  // (i & 3) ^ ((row >> 1) & 3): the read swizzle below).  Only these four waves touch the vector-memory path for
  // loads: with all eight issuing (the A pieces from the compute half) a step's 48 pieces took 1.4-1.6k cycles to
  // issue instead of 0.95k, same-box.
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ ((lrow >> 1) & 3);
  const int lw = wave & 3;
  const uint16_t* pa[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint16_t* pb = nullptr;
  int lu = ub, lt_a = 0, lt_b = 0, lu_b = ub;
  auto set_a_ptrs = [&](int u) __attribute__((always_inline)) {
    int m0, n0, bz;
    decode(u, m0, n0, bz);
    const uint16_t* base = ((lw >> 1) ? G.Al : G.Ah) + (int64_t)bz * G.a_bs + lchunk * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row = min(m0 + 64 * (lw & 1) + 16 * j + lrow, G.M - 1);      // rows past M: valid memory, never stored
      if (DISGAT_PL_DIAG && (G.dbg & 4)) row &= 127;
      pa[j] = base + (int64_t)row * G.lda;
    }
  };
  auto set_b_ptr = [&](int u) __attribute__((always_inline)) {
    int m0, n0, bz;
    decode(u, m0, n0, bz);
    pb = G.Bt + ((((int64_t)bz * 2 + (lw >> 1)) * G.nsteps + (n0 >> 8)) * KT * 256 + 128 * (lw & 1)) * 32 + lane * 8;
  };
  unsigned char* const ldsA = lds_pl;
  unsigned char* const ldsB = lds_pl + NSA * PL_A_SLOT;
  int la = 0, lb = 0;                                  // ring slots the next issues fill
  // the A tile / the weight tile of the next not-yet-requested k-step (their cursors run 2 and 1 steps ahead of the
  // arithmetic); past the block's last unit they keep re-loading it: the wait counts assume every step issues
  auto issue_a = [&]() __attribute__((always_inline)) {
    unsigned char* dst = ldsA + la * PL_A_SLOT + (lw >> 1) * (PL_BM * 64) + (lw & 1) * (64 * 64);
    // (plain cache policy: the non-temporal hint on these once-read A pieces - meant to keep the weight planes in L2 - ran
    // the projection 4.98 vs 4.18 ms and the fuser 3.23 vs 2.90 ms, same box, interleaved rounds)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      glds16(pa[j], dst + j * 1024);
      pa[j] += 32;
    }
    la = (la + 1 == NSA) ? 0 : la + 1;
    if (++lt_a == KT) {
      lt_a = 0;
      if (lu + 1 < ue) ++lu;
      set_a_ptrs(lu);
    }
  };
  auto issue_b = [&]() __attribute__((always_inline)) {
    unsigned char* dst = ldsB + lb * PL_B_SLOT + (lw >> 1) * (PL_BN * 64) + (lw & 1) * (128 * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(pb + j * 512, dst + j * 1024);
    pb += 256 * 32;
    lb = (lb + 1 == NSB) ? 0 : lb + 1;
    if (++lt_b == KT) {
      lt_b = 0;
      if (lu_b + 1 < ue) ++lu_b;
      set_b_ptr(lu_b);
    }
  };
  // ---- prologue, in the order the loop continues: A(0), B(0), A(1); then every step issues B(s + 1), A(s + 2)
  if (ld_wave) {
    set_a_ptrs(lu);
    set_b_ptr(lu_b);
    issue_a();
    issue_b();
    issue_a();
  }

  // fragment read offset inside a tile of 64-byte rows (tile bases are multiples of 16 rows)
  const int fo = (lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 1) & 3)) << 4);
  int sa = 0, sb = 0;                                  // ring slots of the current k-step
  bool relax = false;                                  // previous unit ended with a full epilogue (exactly S_L stores)
  const bool stamp_on = DISGAT_PL_DIAG && (G.dbg & 32) != 0;
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0;
  auto stamp = [&](int ph) __attribute__((always_inline)) {
    if (!stamp_on) return;
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (ph >= 0) st_acc[ph] += t - st_prev;
    st_prev = t;
  };
  stamp(-1);

  for (int u = ub; u < ue; ++u) {
    int m0, n0, bz;
    decode(u, m0, n0, bz);
    const int q = lane >> 4;
    const int row0 = m0 + wrow + (lane & 15);
    const int col0 = n0 + wn * 64 + 8 * q;

    f32x4v acc[MC][4], acx[MC][4];
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        acx[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
    if (G.bias != nullptr || G.init != nullptr) {
      // bias and the additive matrix seed the hi*hi accumulator (times s_A s_B, a power of two: exact)
      const float* bias = G.bias ? G.bias + (int64_t)bz * G.N : nullptr;
      const float* init = G.init ? G.init + (int64_t)bz * G.i_bs : nullptr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cj = col0 + 32 * (j >> 1) + 4 * (j & 1);
        const f32x4 bv = bias ? ld4(bias + cj) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MC; ++i) {
          const int row = row0 + 16 * i;
          const bool mine = i < ML || !ld_wave;
          const f32x4 iv = (init && mine && row < G.M) ? ld4(init + (int64_t)row * G.ldi + cj) : f32x4{0.f, 0.f, 0.f, 0.f};
          acc[i][j] = f32x4v{(bv.x + iv.x) * sAB, (bv.y + iv.y) * sAB, (bv.z + iv.z) * sAB, (bv.w + iv.w) * sAB};
        }
      }
    }

    for (int t = 0; t < KT; ++t) {
      // (1) this wave's LDS reads of the previous step have returned; (2) a loader's DMA pieces of THIS step have landed
      // (queue, oldest first: A(s), B(s), A(s + 1) - and for the first step after an epilogue that epilogue's S_L
      // stores: the 4 A(s + 1) pieces and those stores may stay in flight); (3) barrier: everybody's have.  Then the
      // slots read one step ago may be refilled.  A compute wave has no load of its own to wait for - its stores
      // (5/8 of the block's) never sit in front of anything.
      stamp(t == 0 ? 5 : 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ld_wave) {
        if (relax && t == 0) wait_vm<4 + S_L>();
        else wait_vm<4>();
      }
      stamp(0);
      __builtin_amdgcn_s_barrier();
      stamp(1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LG) {
        // the previous unit's partial results are all in LDS (its epilogue's ds_writes were waited for above, the barrier
        // says: by every wave); this unit's epilogue writes them again only after KT more barriers
        if (t == 0 && pend_m0 >= 0) lg_reduce(pend_m0, pend_bz);
      }
      if (ld_wave) {
        issue_b();
        issue_a();
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* Ab = ldsA + sa * PL_A_SLOT + wrow * 64 + fo;
      const unsigned char* Bb = ldsB + sb * PL_B_SLOT + wn * (64 * 64) + fo;
      sa = (sa + 1 == NSA) ? 0 : sa + 1;
      sb = (sb + 1 == NSB) ? 0 : sb + 1;
      f16x8 bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(Bb + j * 1024);
        bl[j] = *reinterpret_cast<const f16x8*>(Bb + PL_BN * 64 + j * 1024);
      }
      if (stamp_on) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(2);
      }
      // per 16-row tile: hi*hi, (weight lo) x (A hi), (weight hi) x (A lo) - 12 MFMAs on one pair of A fragments, the next
      // tile's pair requested before them (left to the compiler each tile read its fragments, waited, then computed:
      // 24 cycles per MFMA instead of 16)
      f16x8 fa[2][2];
      fa[0][0] = *reinterpret_cast<const f16x8*>(Ab);
      fa[0][1] = *reinterpret_cast<const f16x8*>(Ab + PL_BM * 64);
      auto rows = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {
        static_for<decltype(hi_c)::value - decltype(lo_c)::value>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(lo_c)::value + decltype(ic)::value;
          constexpr int cur = i & 1, nxt = cur ^ 1;
          if constexpr (i + 1 < MC) {      // (a loader's last prefetch reads a tile it does not use: valid LDS, harmless)
            fa[nxt][0] = *reinterpret_cast<const f16x8*>(Ab + (i + 1) * 1024);
            fa[nxt][1] = *reinterpret_cast<const f16x8*>(Ab + PL_BM * 64 + (i + 1) * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of this tile's MFMAs
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], fa[cur][0], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], fa[cur][0], acx[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], fa[cur][1], acx[i][j], 0, 0, 0);
        });
      };
      __builtin_amdgcn_s_setprio(1);
      if (!(DISGAT_PL_DIAG && (G.dbg & 2))) {
        rows(std::integral_constant<int, 0>{}, std::integral_constant<int, ML>{});
        if (!ld_wave) rows(std::integral_constant<int, ML>{}, std::integral_constant<int, MC>{});
      } else {
        asm volatile("" ::"v"(fa[0][0]), "v"(fa[0][1]), "v"(bh[0]), "v"(bh[1]), "v"(bh[2]), "v"(bh[3]), "v"(bl[0]), "v"(bl[1]), "v"(bl[2]), "v"(bl[3]));
      }
      __builtin_amdgcn_s_setprio(0);
    }

    stamp(3);
    // ---- epilogue: acc[i][j][r] = C[row0 + 16 i][col0 + 32 (j >> 1) + 4 (j & 1) + r]
    const bool full = m0 + PL_BM <= G.M;
    if (DISGAT_PL_DIAG && (G.dbg & 1)) {
#pragma unroll
      for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]), "v"(acx[i][j]));
      relax = false;
      continue;
    }
    // LG: this wave's two fragments of W2 (rows wn * 64 + 32 g + 8 q .. + 7, output lane & 15), both planes - 16 KB image in
    // L2, fetched per unit so that nothing of it is live across the k-loop (the kernel sits at the 256-register limit)
    f16x8 w2h[2], w2l[2];
    float inv2 = 0.f, xw2 = 0.f;
    if constexpr (LG) {
      const uint16_t* wf = G.W2f + ((size_t)(wn * 2) * 2 * 64 + lane) * 8;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        w2h[g] = *reinterpret_cast<const f16x8*>(wf + (size_t)(g * 2 + 0) * 64 * 8);
        w2l[g] = *reinterpret_cast<const f16x8*>(wf + (size_t)(g * 2 + 1) * 64 * 8);
      }
      inv2 = 1.0f / (sC * *G.w2_scale);
      xw2 = inv2 * (1.0f / 2048.f);
    }
    auto epi = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {
#pragma unroll
      for (int i = decltype(lo_c)::value; i < decltype(hi_c)::value; ++i) {
        const int row = row0 + 16 * i;
        const bool ok = (full || row < G.M) && !(DISGAT_PL_DIAG && (G.dbg & 8));
        f32x4v lacc = f32x4v{0.f, 0.f, 0.f, 0.f}, lacx = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = act_ct<ACT>(fmaf(acx[i][2 * g][r], xw, acc[i][2 * g][r] * inv), G.slope);
            v[4 + r] = act_ct<ACT>(fmaf(acx[i][2 * g + 1][r], xw, acc[i][2 * g + 1][r] * inv), G.slope);
          }
          if constexpr (LG) {
            u32x2 h0, l0, h1, l1;
            split4h(f32x4{v[0], v[1], v[2], v[3]} * sC, h0, l0);
            split4h(f32x4{v[4], v[5], v[6], v[7]} * sC, h1, l1);
            const f16x8 th = __builtin_bit_cast(f16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
            const f16x8 tl = __builtin_bit_cast(f16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
            lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[g], th, lacc, 0, 0, 0);
            lacx = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[g], th, lacx, 0, 0, 0);
            lacx = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[g], tl, lacx, 0, 0, 0);
          }
          if constexpr (F32) {
            float* cp = G.C + (int64_t)bz * G.c_bs + (int64_t)row * G.ldc + col0 + 32 * g;
            if (ok) {
              st4(cp, f32x4{v[0], v[1], v[2], v[3]});
              st4(cp + 4, f32x4{v[4], v[5], v[6], v[7]});
            }
          }
          if constexpr (PL) {
            u32x2 h0, l0, h1, l1;
            split4h(f32x4{v[0], v[1], v[2], v[3]} * sC, h0, l0);
            split4h(f32x4{v[4], v[5], v[6], v[7]} * sC, h1, l1);
            const int64_t o = (int64_t)bz * G.p_bs + (int64_t)row * G.ldp + col0 + 32 * g;
            if (ok) {
              *reinterpret_cast<u32x4*>(G.Ch + o) = u32x4{h0.x, h0.y, h1.x, h1.y};
              *reinterpret_cast<u32x4*>(G.Cl + o) = u32x4{l0.x, l0.y, l1.x, l1.y};
            }
          }
        }
        if constexpr (LG) {
          // lane (n = lane & 15, q): outputs 4 q .. 4 q + 3 of row wrow + 16 i + n, this strip's share
          const f32x4 pv = {fmaf(lacx[0], xw2, lacc[0] * inv2), fmaf(lacx[1], xw2, lacc[1] * inv2),
                            fmaf(lacx[2], xw2, lacc[2] * inv2), fmaf(lacx[3], xw2, lacc[3] * inv2)};
          *reinterpret_cast<f32x4*>(lg_part + (wn * PL_BM + wrow + 16 * i + (lane & 15)) * 16 + 4 * q) = pv;
        }
      }
    };
    epi(std::integral_constant<int, 0>{}, std::integral_constant<int, ML>{});
    if (!ld_wave) epi(std::integral_constant<int, ML>{}, std::integral_constant<int, MC>{});
    relax = full;       // a ragged tile may have skipped store instructions: the next unit counts strictly
    if constexpr (LG) {
      pend_m0 = m0;
      pend_bz = bz;
    }
    stamp(4);
  }
  if constexpr (LG) {
    if (pend_m0 >= 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      lg_reduce(pend_m0, pend_bz);
    }
  }
  // the ring still holds DMAs in flight (re-loads of the last unit): they must land before the block's LDS is released
  wait_vm<0>();
  if (stamp_on && lane == 0 && (wave == 0 || wave == 4)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) atomicAdd(&pl_stamps[(wave >> 2) * 8 + k], st_acc[k]);
  }
}

// fp32 [batch][M][K] (row stride ldx, batch stride x_bs) -> planes Ph, Pl (row stride ldp halfs, batch stride p_bs)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, int64_t ldx, int64_t x_bs, int M,
                                                           int K4, int batch, const float* __restrict__ bound,
                                                           uint16_t* __restrict__ Ph, uint16_t* __restrict__ Pl,
                                                           int64_t ldp, int64_t p_bs) {
  const float s = f16_scale(*bound);
  const int64_t total = (int64_t)batch * M * K4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / K4;
    const int c = (int)(i - r * K4) * 4;
    const int64_t bz = r / M, m = r - bz * M;
    u32x2 h, l;
    split4h(ld4(X + bz * x_bs + m * ldx + c) * s, h, l);
    *reinterpret_cast<u32x2*>(Ph + bz * p_bs + m * ldp + c) = h;
    *reinterpret_cast<u32x2*>(Pl + bz * p_bs + m * ldp + c) = l;
  }
}

// planes -> fp32: v = (hi + lo * 2^-11) / s
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const uint16_t* __restrict__ Ph, const uint16_t* __restrict__ Pl,
                                                            int64_t ldp, int64_t p_bs, int M, int K4, int batch,
                                                            const float* __restrict__ bound, float* __restrict__ X,
                                                            int64_t ldx, int64_t x_bs) {
  const float inv = 1.0f / f16_scale(*bound);
  const float invl = inv * (1.0f / 2048.f);
  const int64_t total = (int64_t)batch * M * K4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / K4;
    const int c = (int)(i - r * K4) * 4;
    const int64_t bz = r / M, m = r - bz * M;
    const u32x2 h = *reinterpret_cast<const u32x2*>(Ph + bz * p_bs + m * ldp + c);
    const u32x2 l = *reinterpret_cast<const u32x2*>(Pl + bz * p_bs + m * ldp + c);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 hh = __builtin_bit_cast(f16x4, h), ll = __builtin_bit_cast(f16x4, l);
    f32x4 v;
    v.x = fmaf((float)ll.x, invl, (float)hh.x * inv);
    v.y = fmaf((float)ll.y, invl, (float)hh.y * inv);
    v.z = fmaf((float)ll.z, invl, (float)hh.z * inv);
    v.w = fmaf((float)ll.w, invl, (float)hh.w * inv);
    st4(X + bz * x_bs + m * ldx + c, v);
  }
}

static int n_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

template <int ACT, bool F32, bool PL, bool LG = false>
static int launch_planes_ring(int, const GemmPArgs& G, hipStream_t st) {
  constexpr int lds_bytes = 3 * PL_A_SLOT + 2 * PL_B_SLOT + (LG ? PL_LG_LDS : 0);
  static bool set = false;
  auto fn = gemm_planes_kernel<ACT, F32, PL, LG>;
  if (!set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return fail((int)e, "gemm_planes: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(e));
    set = true;
  }
  const int grid = G.units < n_cus() ? G.units : n_cus();
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds_bytes, st, G);
  return check_launch("gemm_planes_kernel");
}

template <int ACT>
static int launch_planes_out(bool f32, bool pl, int nsa, const GemmPArgs& G, hipStream_t st) {
  if (f32 && pl) return launch_planes_ring<ACT, true, true>(nsa, G, st);
  if (pl) return launch_planes_ring<ACT, false, true>(nsa, G, st);
  return launch_planes_ring<ACT, true, false>(nsa, G, st);
}

}  // namespace disgat

extern "C" int disgat_gemm_planes(const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, int64_t a_batch_stride,
                                  const uint16_t* Bt_planes, const float* a_bound, const float* b_scale, const float* bias,
                                  const float* init, int64_t ldi, int64_t init_batch_stride, float* C, int64_t ldc,
                                  int64_t c_batch_stride, uint16_t* C_hi, uint16_t* C_lo, int64_t ldp,
                                  int64_t p_batch_stride, const float* c_bound, int M, int N, int K, int batch, int act,
                                  float slope, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0) return 0;
  DISGAT_REQUIRE(A_hi && A_lo && Bt_planes && a_bound && b_scale && M > 0 && batch > 0, "gemm_planes: null pointer / bad sizes");
  DISGAT_REQUIRE(C || (C_hi && C_lo && c_bound), "gemm_planes: no output (C, or C_hi + C_lo + c_bound)");
  DISGAT_REQUIRE((C_hi == nullptr) == (C_lo == nullptr), "gemm_planes: C_hi and C_lo go together");
  DISGAT_REQUIRE(N > 0 && N % PL_BN == 0 && K >= 64 && K % 32 == 0, "gemm_planes: N=%d must be a multiple of %d, K=%d of 32 and >= 64", N, PL_BN, K);
  DISGAT_REQUIRE(lda % 8 == 0 && a_batch_stride % 8 == 0 && aligned16(A_hi) && aligned16(A_lo) && aligned16(Bt_planes),
                 "gemm_planes: plane rows must be 16-byte aligned (lda, batch stride multiples of 8 halfs)");
  DISGAT_REQUIRE(!C || (ldc % 4 == 0 && c_batch_stride % 4 == 0 && aligned16(C)), "gemm_planes: C rows must be 16-byte aligned");
  DISGAT_REQUIRE(!C_hi || (ldp % 8 == 0 && p_batch_stride % 8 == 0 && aligned16(C_hi) && aligned16(C_lo)),
                 "gemm_planes: output plane rows must be 16-byte aligned");
  DISGAT_REQUIRE(!init || (ldi % 4 == 0 && init_batch_stride % 4 == 0 && aligned16(init)), "gemm_planes: init rows must be 16-byte aligned");
  DISGAT_REQUIRE(!bias || aligned16(bias), "gemm_planes: bias must be 16-byte aligned");
  DISGAT_REQUIRE(act >= 0 && act <= 2, "gemm_planes: act must be 0 (none), 1 (elu) or 2 (leaky relu)");
  const int mt = (M + PL_BM - 1) / PL_BM, nsteps = N / PL_BN;
  const int64_t units = (int64_t)mt * nsteps * batch;
  DISGAT_REQUIRE(units < ((int64_t)1 << 31), "gemm_planes: too many tiles");
  GemmPArgs G{A_hi, A_lo, lda, a_batch_stride, Bt_planes, a_bound, b_scale, bias, init, ldi, init_batch_stride,
              C, ldc, c_batch_stride, C_hi, C_lo, ldp, p_batch_stride, c_bound, M, N, K, batch, slope, nsteps, (int)units,
              getenv("DISGAT_PL_DEBUG") ? atoi(getenv("DISGAT_PL_DEBUG")) : 0};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nsa = (K / 32 >= 4) ? 5 : 3;
  const bool f32 = C != nullptr, pl = C_hi != nullptr;
  if (act == 1) return launch_planes_out<1>(f32, pl, nsa, G, st);
  if (act == 2) return launch_planes_out<2>(f32, pl, nsa, G, st);
  return launch_planes_out<0>(f32, pl, nsa, G, st);
}

// The DifHead classifier on the head planes (pretrainer.py:819-832 over models.py:523-543 with cls_layer == 2):
//   L[(m, b)][:] = act( A[b][m][:] W1[b] + bias + init[m][:] ) W2 + bias2          N = 256 hidden columns, n_out <= 16
// The hidden layer ([M x batch, 256] fp32: 8.2 GB at M = 1e6, batch = 8) is neither written nor read back: its tiles
// become the operand of the second product while they are still in the accumulators (gemm_planes_kernel<.., LG = true>).
extern "C" int disgat_gemm_planes_logits(const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, int64_t a_batch_stride,
                                         const uint16_t* Bt_planes, const float* a_bound, const float* b_scale,
                                         const float* bias, const float* init, int64_t ldi, int64_t init_batch_stride,
                                         const float* mid_bound, const uint16_t* W2_frags, const float* w2_scale,
                                         const float* bias2, float* L, int M, int N, int K, int batch, int n_out, int act,
                                         float slope, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0) return 0;
  DISGAT_REQUIRE(A_hi && A_lo && Bt_planes && a_bound && b_scale && mid_bound && W2_frags && w2_scale && L && M > 0 && batch > 0,
                 "gemm_planes_logits: null pointer / bad sizes");
  DISGAT_REQUIRE(N == PL_BN && K >= 64 && K % 32 == 0, "gemm_planes_logits: N=%d must be %d, K=%d a multiple of 32 and >= 64", N, PL_BN, K);
  DISGAT_REQUIRE(n_out >= 1 && n_out <= 16, "gemm_planes_logits: n_out=%d must be in 1..16", n_out);
  DISGAT_REQUIRE(lda % 8 == 0 && a_batch_stride % 8 == 0 && aligned16(A_hi) && aligned16(A_lo) && aligned16(Bt_planes) && aligned16(W2_frags),
                 "gemm_planes_logits: plane rows must be 16-byte aligned (lda, batch stride multiples of 8 halfs)");
  DISGAT_REQUIRE(!init || (ldi % 4 == 0 && init_batch_stride % 4 == 0 && aligned16(init)), "gemm_planes_logits: init rows must be 16-byte aligned");
  DISGAT_REQUIRE((!bias || aligned16(bias)) && (!bias2 || aligned16(bias2)) && ((n_out & 3) != 0 || aligned16(L)),
                 "gemm_planes_logits: bias, bias2 (16 floats) and L must be 16-byte aligned");
  DISGAT_REQUIRE(act >= 0 && act <= 2, "gemm_planes_logits: act must be 0 (none), 1 (elu) or 2 (leaky relu)");
  const int mt = (M + PL_BM - 1) / PL_BM;
  const int64_t units = (int64_t)mt * batch;
  DISGAT_REQUIRE(units < ((int64_t)1 << 31) && (int64_t)M * batch < ((int64_t)1 << 31), "gemm_planes_logits: too many tiles / rows");
  GemmPArgs G{A_hi, A_lo, lda, a_batch_stride, Bt_planes, a_bound, b_scale, bias, init, ldi, init_batch_stride,
              nullptr, 0, 0, nullptr, nullptr, 0, 0, mid_bound, M, N, K, batch, slope, 1, (int)units,
              getenv("DISGAT_PL_DEBUG") ? atoi(getenv("DISGAT_PL_DEBUG")) : 0, W2_frags, w2_scale, bias2, L, n_out};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (act == 1) return launch_planes_ring<1, false, false, true>(0, G, st);
  if (act == 2) return launch_planes_ring<2, false, false, true>(0, G, st);
  return launch_planes_ring<0, false, false, true>(0, G, st);
}

extern "C" int disgat_debug_stamps(unsigned long long* out16, int reset) {
  using namespace disgat;
  if (hipDeviceSynchronize() != hipSuccess) return fail(-1, "debug_stamps: sync failed");
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(pl_stamps), sizeof(pl_stamps)) != hipSuccess) return fail(-1, "debug_stamps: copy failed");
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pl_stamps), z, sizeof(z)) != hipSuccess) return fail(-1, "debug_stamps: reset failed");
  }
  return 0;
}

extern "C" int disgat_split_planes(const float* X, int64_t ldx, int64_t x_batch_stride, int M, int K, int batch,
                                   const float* bound, uint16_t* P_hi, uint16_t* P_lo, int64_t ldp, int64_t p_batch_stride,
                                   disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0 || K == 0) return 0;
  DISGAT_REQUIRE(X && bound && P_hi && P_lo && M > 0 && K > 0 && batch > 0, "split_planes: null pointer / bad sizes");
  DISGAT_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && x_batch_stride % 4 == 0 && ldp % 4 == 0 && p_batch_stride % 4 == 0 && aligned16(X) &&
                     (reinterpret_cast<uintptr_t>(P_hi) & 7) == 0 && (reinterpret_cast<uintptr_t>(P_lo) & 7) == 0,
                 "split_planes: K and the strides must be multiples of 4, X 16-byte and the planes 8-byte aligned");
  const int64_t work = (int64_t)batch * M * (K / 4);
  const int grid = (int)(work / 256 + 1 < 16384 ? work / 256 + 1 : 16384);
  hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, ldx,
                     x_batch_stride, M, K / 4, batch, bound, P_hi, P_lo, ldp, p_batch_stride);
  return check_launch("split_planes_kernel");
}

extern "C" int disgat_planes_to_f32(const uint16_t* P_hi, const uint16_t* P_lo, int64_t ldp, int64_t p_batch_stride, int M,
                                    int K, int batch, const float* bound, float* X, int64_t ldx, int64_t x_batch_stride,
                                    disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0 || batch == 0 || K == 0) return 0;
  DISGAT_REQUIRE(X && bound && P_hi && P_lo && M > 0 && K > 0 && batch > 0, "planes_to_f32: null pointer / bad sizes");
  DISGAT_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && x_batch_stride % 4 == 0 && ldp % 4 == 0 && p_batch_stride % 4 == 0 && aligned16(X) &&
                     (reinterpret_cast<uintptr_t>(P_hi) & 7) == 0 && (reinterpret_cast<uintptr_t>(P_lo) & 7) == 0,
                 "planes_to_f32: K and the strides must be multiples of 4, X 16-byte and the planes 8-byte aligned");
  const int64_t work = (int64_t)batch * M * (K / 4);
  const int grid = (int)(work / 256 + 1 < 16384 ? work / 256 + 1 : 16384);
  hipLaunchKernelGGL(planes_to_f32_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P_hi, P_lo, ldp,
                     p_batch_stride, M, K / 4, batch, bound, X, ldx, x_batch_stride);
  return check_launch("planes_to_f32_kernel");
}
