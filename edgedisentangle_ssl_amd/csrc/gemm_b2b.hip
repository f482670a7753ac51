// Back-to-back GEMM: the per-head output projection of a DISGAT layer and the FuseLayer that consumes it, in ONE kernel.
//
//   out = act2( cat_h [ elu( Z_h W1_h + b1_h ) ] W2 + b2 )         Z_h [M, K1] fp16 planes, W1_h [K1, N1], W2 [H*N1, N2]
//
// /root/reference/layers.py:397-399 (h_em = x W_em, aggregated: here Z_h W_em,h), :404-407 (GCN: + bias), :508 (F.elu),
// models.py:230-233 (the H head outputs handed to the fuser) and layers.py:896-921 (FuseLayer: cat -> Linear -> leaky_relu).
//
// Why.  As two launches (gemm_planes.hip) the projection writes the concatenated heads as fp16 planes - 8.2 GB at
// N = 1e6, H = 8, nhid = 256 - and the fuser reads them straight back to produce 1 GB; for SupEdge, DisEdge and get_em
// nothing else reads that buffer.  Here a 32-column chunk of elu(Z_h W1_h) never leaves the registers it was accumulated in:
//   * a wave keeps its 32 rows of Z_h as MFMA fragments in registers (2 row tiles x K1/32 k-steps x hi/lo = 128 VGPRs at
//     K1 = 256), loaded from the planes the edge pass wrote - no operand arithmetic;
//   * GEMM 1 is the transposed product (weight fragment as the MFMA's A operand), so a lane ends with 4 consecutive columns
//     of ONE row per 16 x 16 tile: after ELU and the hi / lo split, the two tiles of a chunk ARE the lane's 8 k-values of an
//     operand fragment of GEMM 2's next k-step - provided W2's rows are stored in that lane order (a permutation inside every
//     group of 32 rows: k' = 8 q + e  <->  column 4 q + e (e < 4), 16 + 4 q + e - 4 (e >= 4); done once by the weight
//     preparation, ops_gemm.presplit_b2b).  No LDS round trip, no shuffle;
//   * GEMM 2's [32 rows x N2] result (x 2: the f16x3 scheme's hi*hi and cross-term accumulators) stays in 256 accumulator
//     registers for all H heads: one wave per SIMD with the whole 512-register file, 4 waves = 128 rows per workgroup;
//   * both weight streams come as ONE image per chunk (W1 chunk: 2 planes x 2 n-tiles x K1/32 fragment blocks; W2 chunk:
//     2 planes x N2/16 fragment blocks; 32 + 32 KB at 256 / 256 - the host lays the chunks out in this order) through a
//     2-slot LDS ring by LDS-DMA, one chunk ahead, shared by the 4 waves; one s_barrier per chunk.
// One wave per SIMD means nothing else fills the matrix pipe while this wave issues anything else: a group of 4 fragment
// reads issued back to back idled it for ~50 cycles, 16 DMA pieces at the top of a step for ~1000 (in-kernel stamps,
// tools/b2b_stamps.py).  So the instruction stream is laid out by hand: per group of 12 MFMAs the next group's four
// ds_read_b128 sit one by one in the first MFMA gaps (counted lgkmcnt waits in front of each fragment's first use), one DMA
// piece of the next chunk in the tail; every statement is pinned by sched_barrier.
// The ELU / scale / split arithmetic of a chunk (~150 vector instructions, 800 cycles when they stood alone between the two
// GEMMs) runs UNDER matrix work: GEMM 2 is one chunk behind - step s is [GEMM 1 (s)] [GEMM 2 (s-1) with the arithmetic of
// chunk s, two or three instructions at a time, in its MFMA gaps], the intermediate of chunk s-1 waiting in 16 registers.
// Step 0 runs GEMM 2 on a zero intermediate; GEMM 2 of the last chunk follows the loop.
// Measured and dropped (profiles/r05/b2b_*_stamps.log, tools/b2b_ablate.sh): the arithmetic as 192 single instructions, two per
// MFMA gap - they cost their full ~3.6 cycles each, nothing hides under a 16-cycle MFMA in this one-wave-per-SIMD stream
// (6.03 ms against 5.88); the whole kernel on v_mfma_f32_32x32x16_f16 (one 32 x 32 tile per chunk, 6 MFMAs per group, four
// instructions per 32-cycle gap) - about half of the arithmetic hides there, but the accumulator seed, the dependent MFMA
// pairs on the single tile and the paced Z requests cost more than that buys: 6.88 ms.
// Arithmetic per chunk and wave: 96 + 96 MFMAs (16x16x32 f16), 64 ds_read_b128, 16 DMA pieces, ~150 vector instructions.
// HBM: the Z planes once (8.2 GB) + 1 GB of output; the head buffer's 16.4 GB round trip is gone.
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "disgat_api.h"
#include "gemm_common.h"

namespace disgat {

namespace {

struct B2BArgs {
  const uint16_t* Zh;     // planes of Z * s_Z: row m, head h at Zh + m * ldz + h * z_hs (halfs)
  const uint16_t* Zl;
  int64_t ldz, z_hs;
  const float* z_bound;   // device scalar: s_Z = f16_scale(*z_bound)
  const uint16_t* Wc;     // [H][N1 / 32] chunk images: W1 chunk [2 planes][2 n-tiles][KT] + W2 chunk [2 planes][NT2] blocks of 1 KB
  const float* s1;
  const float* bias1;     // [H * N1] or null
  const float* s2;
  const float* bias2;     // [N2] or null
  const float* c_bound;   // device scalar >= max |elu(.)|: the intermediate's scale s_C = f16_scale(*c_bound)
  float* C;               // [M][ldc] fp32
  int64_t ldc;
  int M, H, N1;
  int act2;               // 0 none, 2 leaky relu
  float slope;
};

// -DBB_DIAG=<mask> builds (tools/b2b_ablate.sh): 1 no ring refills, 2 no GEMM 1 MFMAs, 4 no ELU / split arithmetic, 8 no GEMM 2
// MFMAs, 16 no Z reloads, 32 phase stamps - timing ablations, compile-time so that the instruction stream has no branch of
// theirs; results are then meaningless
#ifndef BB_DIAG
#define BB_DIAG 0
#endif
__device__ unsigned long long bb_stamps[16];    // BB_DIAG, dbg & 32: s_memtime ticks per step phase, summed over wave 0 of every block

constexpr int BB_ROWS = 128;      // rows per workgroup: 4 waves x 2 row tiles of 16

template <int N>
__device__ __forceinline__ void bb_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* bb_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* bb_glb_ptr_t;

__device__ __forceinline__ void bb_glds16(const unsigned char* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((bb_glb_ptr_t)src, (bb_lds_ptr_t)dst, 16, 0, 0);
}

template <int... I, class F>
__device__ __forceinline__ void bb_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void bb_static_for(F&& f) {
  bb_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Everything the hand-laid stream consists of is an asm statement (or pinned by sched_barrier): hipcc neither reorders nor
// counts it (cdna_hip_programming.md 5.7).
//   * fragment reads + counted waits: form (ii) - the destination counts as written at the read; the wait statement in
//     front of the first use names it "+v";
//   * GEMM 1's MFMAs with VGPR accumulators: with a 512-register budget hipcc selects the AGPR form for every MFMA of a
//     kernel; GEMM 2's accumulators fill the 256 accumulator registers exactly, and 32 more for GEMM 1's made the allocator
//     shuttle six of GEMM 2's tiles between the two files around their MFMAs.  hipcc pads no hazards around asm:
//     bb_fence8 carries the wait states (a VALU-written C operand -> MFMA; an MFMA's D -> VALU);
//   * the next head's Z fragments (requested by a head's LAST step into the registers of k-steps already retired) and the
//     counted vmcnt in front of their first use: with a register load and LDS-DMA pieces pending together hipcc assumes
//     out-of-order completion and waits vmcnt(0) - for the DMA pieces of the NEXT chunk, requested a moment earlier.
template <int OFF>
__device__ __forceinline__ void bb_rd(f16x8& d, uint32_t sa) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(sa), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void bb_lw(f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N));
}
__device__ __forceinline__ void bb_mfma_v(f32x4v& c, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void bb_fence8(f32x4v (&x)[2][2], f32x4v (&y)[2][2], bool out) {
  if (out)
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(y[0][0]), "+v"(y[0][1]), "+v"(y[1][0]), "+v"(y[1][1]));
  else
    asm volatile("s_nop 1" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(y[0][0]), "+v"(y[0][1]), "+v"(y[1][0]), "+v"(y[1][1]));
}
template <int OFF>
__device__ __forceinline__ void bb_z_load(f16x8& d, const uint16_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void bb_z_wait(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
#define BB_PIN __builtin_amdgcn_sched_barrier(0)

// KT = K1 / 32 k-steps of GEMM 1; NT2 = N2 / 16 column tiles of GEMM 2
template <int KT, int NT2, bool BIAS1>
__global__ __launch_bounds__(256, 1) void proj_fuse_kernel(const B2BArgs G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_bb[];
  constexpr int W1_PLANE = 2 * KT * 1024;       // bytes of one plane of a W1 chunk: 2 n-tiles x KT fragment blocks of 1 KB
  constexpr int W1_PART = 2 * W1_PLANE;
  constexpr int W2_PLANE = NT2 * 1024;          // one plane of a W2 chunk: NT2 fragment blocks
  constexpr int SLOT = W1_PART + 2 * W2_PLANE;  // one chunk image = one ring slot
  constexpr int G1 = KT, G2 = NT2 / 2;          // groups of 12 MFMAs per chunk: GEMM 1 (one k-step each), GEMM 2 (two column tiles)
  constexpr int NP = SLOT / 1024 / 4;           // DMA pieces (1 KB wave instructions) per chunk and wave
  static_assert(NP == G1 + G2, "one DMA piece per group");
  constexpr int PP = (NP + G1 - 1) / G1;        // pieces per GEMM 1 group: all of them during GEMM 1, so that they have GEMM 2 to land in
  constexpr int ZL = 4 * KT;                    // Z loads per head and wave
  static_assert(NT2 % 2 == 0 && (KT == 2 || KT == 4 || KT == 8) && SLOT <= 65536, "tiling");
  float* const bias_lds = reinterpret_cast<float*>(lds_bb + 2 * SLOT);      // BIAS1: [H * N1] floats

  const int lane = threadIdx.x & 63;
  const int wave = rfl(threadIdx.x >> 6);
  const int m0 = blockIdx.x * BB_ROWS;
  const int nj = G.N1 >> 5;                     // chunks (32 columns of a head = one k-step of GEMM 2) per head
  const int h_rot = (int)(blockIdx.x % (unsigned)G.H);      // workgroups start on different heads: they stream different weights
  auto head_of = [&](int hs) __attribute__((always_inline)) {
    const int h = hs + h_rot;
    return h >= G.H ? h - G.H : h;
  };

  const float sZ = f16_scale(*G.z_bound);
  const float s1 = *G.s1, s2 = *G.s2;
  const float sC = f16_scale(*G.c_bound);
  const float sAB1 = sZ * s1, inv1 = 1.0f / sAB1, xw1 = inv1 * (1.0f / 2048.f);
  const float sAB2 = sC * s2, inv2 = 1.0f / sAB2, xw2 = inv2 * (1.0f / 2048.f);

  // ---- the weight ring.  A chunk image is SLOT contiguous bytes = its LDS image; piece q (1 KB) of a chunk goes from
  // image + q KB to slot + q KB, wave w moves pieces w, w + 4, ...: a wave-uniform base + the lane's 16 bytes
  const unsigned char* const wc = reinterpret_cast<const unsigned char*>(G.Wc) + wave * 1024 + lane * 16;
  unsigned char* const ring_w = lds_bb + wave * 1024;
  auto chunk_src = [&](int h, int j) __attribute__((always_inline)) { return wc + (int64_t)(h * nj + j) * SLOT; };
  {
    // chunk 0: its W1 part into slot 0; its W2 part into slot 1 - what step 0's GEMM 2 multiplies its ZERO intermediate
    // with (any finite image does); step 0 then requests W1 of chunk 1 and W2 of chunk 0 like every step
    const unsigned char* src = chunk_src(head_of(0), 0);
#pragma unroll
    for (int i = 0; i < NP; ++i) bb_glds16(src + i * 4096, ring_w + (i < KT ? 0 : SLOT) + i * 4096);
  }
  if constexpr (BIAS1) {
    // GEMM 1's bias through LDS: a global load inside the loop would sit behind the chunk's DMA pieces in the in-order
    // vector-memory counter, and waiting for it would wait for them (visible to every wave after the first barrier)
    for (int i = threadIdx.x; i < G.H * G.N1; i += 256) bias_lds[i] = G.bias1[i];
  }

  // ---- this wave's 32 rows of Z_h as fragments: lane (r = lane & 15, q = lane >> 4) holds k = 32 t + 8 q .. + 7 of rows
  // 16 rt + r (16 bytes per plane and k-step)
  f16x8 ah[2][KT], al[2][KT];
  const int rowA = m0 + wave * 32 + (lane & 15);          // + 16 rt
  const int64_t zoff[2] = {(int64_t)min(rowA, G.M - 1) * G.ldz + (lane >> 4) * 8,
                           (int64_t)min(rowA + 16, G.M - 1) * G.ldz + (lane >> 4) * 8};      // rows past M: valid memory, never stored
  bb_static_for<KT>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const int64_t ho = (int64_t)head_of(0) * G.z_hs + t * 32;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      ah[rt][t] = *reinterpret_cast<const f16x8*>(G.Zh + zoff[rt] + ho);
      al[rt][t] = *reinterpret_cast<const f16x8*>(G.Zl + zoff[rt] + ho);
    }
  });

  // ---- GEMM 2's accumulators: [row tile][column tile], hi*hi and cross terms; bias seeds hi*hi (times s_C s_2: exact)
  f32x4v acc2[2][NT2], acx2[2][NT2];
  const int colq = 4 * (lane >> 4);
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (G.bias2) bv = ld4(G.bias2 + 16 * nt + colq);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      acc2[rt][nt] = f32x4v{bv.x * sAB2, bv.y * sAB2, bv.z * sAB2, bv.w * sAB2};
      acx2[rt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
  }

  // the intermediate of the PREVIOUS chunk, GEMM 2's operand this step (fragments [row tile]: hi, lo)
  f16x8 mh[2], ml[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    mh[rt] = __builtin_bit_cast(f16x8, u32x4{0u, 0u, 0u, 0u});
    ml[rt] = __builtin_bit_cast(f16x8, u32x4{0u, 0u, 0u, 0u});
  }

  // LDS byte address of slot 0 + the lane's 16 bytes of a fragment block
  const uint32_t sa0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)(lds_bb) + lane * 16;

  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
  constexpr bool stamp_on = (BB_DIAG & 32) != 0;
  auto stamp = [&](int ph) __attribute__((always_inline)) {
    if constexpr (!stamp_on) return;
    unsigned long long tt;
    BB_PIN;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    BB_PIN;
    if (ph >= 0) st_acc[ph] += tt - st_prev;
    st_prev = tt;
  };
  stamp(-1);

  constexpr std::integral_constant<int, 0> c0{};
  constexpr std::integral_constant<int, 1> c1{};
  constexpr std::integral_constant<int, 2> c2{};
  constexpr std::integral_constant<int, 3> c3{};
  // A group: 12 MFMAs on fragments cur[0..3] - MFMA i uses cur[0], cur[0], cur[1], cur[1] (hi x hi), cur[2], cur[2], cur[3],
  // cur[3] (lo x hi), cur[0], cur[0], cur[1], cur[1] (hi x lo) - with the NEXT group's four reads in the first four gaps
  // and the waits counted on what is outstanding in issue order: [c0 c1 c2 c3] at entry, then n0, n1 behind M0, M1, ...
  //   W(c0) M0 R(n0) M1 R(n1) W(c1) M2 R(n2) M3 R(n3) W(c2) M4 g M5 g W(c3) M6 g M7 g M8 [tail] g M9 g M10 g M11 g
  // (g: a gap the caller may fill - GEMM 2 puts the ELU / split arithmetic there)
  auto group = [&](f16x8(&cur)[4], auto has_next_c, auto&& mfma, auto&& rdn, auto&& tail, auto&& gap) __attribute__((always_inline)) {
    constexpr bool HN = decltype(has_next_c)::value;
    bb_lw<3>(cur[0]);
    mfma(c0, cur[0]); BB_PIN;
    if constexpr (HN) { rdn(c0); BB_PIN; }
    mfma(c1, cur[0]); BB_PIN;
    if constexpr (HN) { rdn(c1); BB_PIN; }
    bb_lw<HN ? 4 : 2>(cur[1]);
    mfma(c2, cur[1]); BB_PIN;
    if constexpr (HN) { rdn(c2); BB_PIN; }
    mfma(c3, cur[1]); BB_PIN;
    if constexpr (HN) { rdn(c3); BB_PIN; }
    bb_lw<HN ? 5 : 1>(cur[2]);
    mfma(std::integral_constant<int, 4>{}, cur[2]); BB_PIN;
    gap(c0);
    mfma(std::integral_constant<int, 5>{}, cur[2]); BB_PIN;
    gap(c1);
    bb_lw<HN ? 4 : 0>(cur[3]);
    mfma(std::integral_constant<int, 6>{}, cur[3]); BB_PIN;
    gap(c2);
    mfma(std::integral_constant<int, 7>{}, cur[3]); BB_PIN;
    gap(c3);
    mfma(std::integral_constant<int, 8>{}, cur[0]); BB_PIN;
    tail();
    gap(std::integral_constant<int, 4>{});
    mfma(std::integral_constant<int, 9>{}, cur[0]); BB_PIN;
    gap(std::integral_constant<int, 5>{});
    mfma(std::integral_constant<int, 10>{}, cur[1]); BB_PIN;
    gap(std::integral_constant<int, 6>{});
    mfma(std::integral_constant<int, 11>{}, cur[1]); BB_PIN;
    gap(std::integral_constant<int, 7>{});
  };
  auto no_gap = [&](auto) __attribute__((always_inline)) {};

  // GEMM 2's fragments of column tiles 2 np, 2 np + 1 out of the W2 part of the slot at LDS address `sap`
  auto rd2 = [&](uint32_t sap, f16x8(&d)[4], auto nc, auto kc) __attribute__((always_inline)) {
    constexpr int np = decltype(nc)::value, k = decltype(kc)::value;
    bb_rd<W1_PART + (k >> 1) * W2_PLANE + (2 * np + (k & 1)) * 1024>(d[k], sap);
  };
  // GEMM 2 on the intermediate mh / ml: the chunk's k-step of [32 rows] x [N2], two column tiles per group; the fragments of
  // group 0 are already requested into fr[G1 & 1]; gapf(np, k) / tailf(np): what the caller puts into the groups' gaps
  auto gemm2 = [&](uint32_t sap, f16x8(&fr)[2][4], auto&& gapf, auto&& tailf) __attribute__((always_inline)) {
    bb_static_for<G2>([&](auto nc) __attribute__((always_inline)) {
      constexpr int np = decltype(nc)::value;
      f16x8(&cur)[4] = fr[(G1 + np) & 1];
      f16x8(&nxt)[4] = fr[(G1 + np + 1) & 1];
      auto mfma = [&](auto ic, f16x8& w) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, rt = i & 1, c = (i >> 1) & 1, ph = i >> 2, nt = 2 * np + c;
        if constexpr ((BB_DIAG & 8) != 0) {
          asm volatile("" ::"v"(w));
        } else if constexpr (ph == 0) {
          acc2[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, mh[rt], acc2[rt][nt], 0, 0, 0);
        } else if constexpr (ph == 1) {
          acx2[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, mh[rt], acx2[rt][nt], 0, 0, 0);
        } else {
          acx2[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ml[rt], acx2[rt][nt], 0, 0, 0);
        }
      };
      auto rdn = [&](auto kc) __attribute__((always_inline)) { rd2(sap, nxt, std::integral_constant<int, (np + 1 < G2 ? np + 1 : 0)>{}, kc); };
      auto tail = [&]() __attribute__((always_inline)) { tailf(nc); };
      auto gap = [&](auto kc) __attribute__((always_inline)) { gapf(nc, kc); };
      if constexpr (np + 1 < G2) group(cur, std::true_type{}, mfma, rdn, tail, gap);
      else group(cur, std::false_type{}, mfma, rdn, tail, gap);
    });
  };

  // One chunk.  first: a head's first chunk after a LAST step; LAST: the head's last chunk - the next head's Z fragments
  // are requested k-step by k-step as its GEMM 1 retires this head's, behind ALL of the next chunk's DMA pieces (which a
  // LAST step therefore issues up front), so that the next step's counted wait can tell the two apart.
  auto step = [&](int s, int h, int j, int h_next, bool first, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    const int slot = s & 1;
    // this step requests W1 of the NEXT chunk (into the other slot, whose W1 part GEMM 1 of step s-1 has left) and W2 of ITS
    // OWN chunk (into this slot, whose W2 part GEMM 2 of step s-1 has left: it held chunk s-2's); the last step re-requests
    // its own W1: the stream stays branch-free
    const unsigned char* csrc = chunk_src(h, j);
    const unsigned char* nsrc = LAST ? (h_next >= 0 ? chunk_src(h_next, 0) : csrc) : chunk_src(h, j + 1);
    unsigned char* ndst = ring_w + (slot ^ 1) * SLOT;
    unsigned char* cdst = ring_w + slot * SLOT;
    constexpr bool refill = !(BB_DIAG & 1);
    auto dma = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < NP) {
        if constexpr (refill) bb_glds16((i < KT ? nsrc : csrc) + i * 4096, (i < KT ? ndst : cdst) + i * 4096);
        BB_PIN;
      }
    };
    // (1) this wave's DMA pieces of step s have landed.  first: the ZL loads of this head's Z fragments were issued behind
    // those pieces and stay in flight - a counted wait holds each k-step of GEMM 1 below until ITS fragments are there, so
    // the burst (128 KB per workgroup) lands under the arithmetic; (2) barrier: everybody's pieces have landed, and everybody
    // is done reading step s-1, whose slot this step refills.  (Not first: the BUILTIN wait, which hipcc's wait-count pass
    // sees - behind an asm wait it re-waits vmcnt(0) for the prologue's fragment loads in front of the first MFMA of every step.)
    stamp(5);
    if (first) bb_wait_vm<ZL>();
    else __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), expcnt / lgkmcnt untouched
    stamp(0);
    __builtin_amdgcn_s_barrier();
    stamp(1);
    BB_PIN;
    if constexpr (LAST) bb_static_for<NP>([&](auto pc) __attribute__((always_inline)) { dma(pc); });

    // ---- GEMM 1: [32 rows] x [32 columns of head h]; accumulators seeded with the bias (times s_Z s_1: exact) or zero
    f32x4v acc1[2][2], acx1[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS1) bv = *reinterpret_cast<const f32x4*>(bias_lds + h * G.N1 + j * 32 + 16 * ct + colq);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc1[rt][ct] = f32x4v{bv.x * sAB1, bv.y * sAB1, bv.z * sAB1, bv.w * sAB1};
        acx1[rt][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
    }
    bb_fence8(acc1, acx1, false);
    BB_PIN;
    const uint32_t sa = sa0 + slot * SLOT;                 // W1 of this chunk
    const uint32_t sa_prev = sa0 + (slot ^ 1) * SLOT;      // W2 of the previous chunk
    // LAST: the next head's Z fragments of k-step t into the registers of this head's (retired by GEMM 1 group t).  A
    // workgroup's 128 KB arrive at the CU's fetch rate (~45 GB/s measured: 2.8 us) and a wave whose vector-memory queue is
    // full stalls at the next issue - requested group by group during GEMM 1 they stopped its MFMA stream for ~5.7 k cycles
    // per head.  So the requests are PACED over the whole step: k-steps 0 .. ZG1-1 behind their GEMM 1 groups, the rest
    // spread over the ELU phase and GEMM 2's groups; each is needed one full step later.
    constexpr int ZG1 = KT < 3 ? KT : 3;
    auto z_req = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      if constexpr (LAST && t < KT) {
        if (h_next >= 0 && !(BB_DIAG & 16)) {
          bb_z_load<t * 64>(ah[0][t], G.Zh + zoff[0] + (int64_t)h_next * G.z_hs);
          bb_z_load<t * 64>(ah[1][t], G.Zh + zoff[1] + (int64_t)h_next * G.z_hs);
          bb_z_load<t * 64>(al[0][t], G.Zl + zoff[0] + (int64_t)h_next * G.z_hs);
          bb_z_load<t * 64>(al[1][t], G.Zl + zoff[1] + (int64_t)h_next * G.z_hs);
        }
        BB_PIN;
      }
    };
    // fragment buffers: [buffer][hi tile 0, hi tile 1, lo tile 0, lo tile 1] of a k-step (GEMM 1) / of two column tiles (GEMM 2)
    f16x8 f[2][4];
    auto rd1 = [&](f16x8(&d)[4], auto tc, auto kc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, k = decltype(kc)::value;
      bb_rd<(k >> 1) * W1_PLANE + ((k & 1) * KT + t) * 1024>(d[k], sa);
    };
    rd1(f[0], c0, c0);
    rd1(f[0], c0, c1);
    rd1(f[0], c0, c2);
    rd1(f[0], c0, c3);
    BB_PIN;
    stamp(2);

    bb_static_for<G1>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      f16x8(&cur)[4] = f[t & 1];
      f16x8(&nxt)[4] = f[(t + 1) & 1];
      // this k-step's Z fragments have landed (younger in the queue: the later k-steps' loads and the DMA pieces this
      // step has issued so far; in any step but a head's first nothing of that is outstanding and the wait falls through)
      if constexpr (!LAST) bb_z_wait<4 * (KT - 1 - t) + (t * PP < NP ? t * PP : NP)>(ah[0][t], ah[1][t], al[0][t], al[1][t]);
      auto mfma = [&](auto ic, f16x8& w) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, rt = i & 1, ct = (i >> 1) & 1, ph = i >> 2;
        if constexpr ((BB_DIAG & 2) != 0) {
          asm volatile("" ::"v"(w));
        } else if constexpr (ph == 0) {
          bb_mfma_v(acc1[rt][ct], w, ah[rt][t]);
        } else if constexpr (ph == 1) {
          bb_mfma_v(acx1[rt][ct], w, ah[rt][t]);
        } else {
          bb_mfma_v(acx1[rt][ct], w, al[rt][t]);
        }
      };
      // the next group's fragments: the next k-step's, or - behind the last k-step - GEMM 2's first two column tiles
      auto rdn = [&](auto kc) __attribute__((always_inline)) {
        if constexpr (t + 1 < G1) rd1(nxt, std::integral_constant<int, (t + 1 < G1 ? t + 1 : 0)>{}, kc);
        else rd2(sa_prev, nxt, c0, kc);
      };
      auto tail = [&]() __attribute__((always_inline)) {
        if constexpr (!LAST) bb_static_for<PP>([&](auto pc) __attribute__((always_inline)) { dma(std::integral_constant<int, t * PP + decltype(pc)::value>{}); });
      };
      group(cur, std::true_type{}, mfma, rdn, tail, no_gap);
      if constexpr (t < ZG1) z_req(tc);
    });

    // ---- ELU, scale, hi / lo split of THIS chunk: 64 micro-steps (8 value pairs x 8 stages of two or three instructions)
    // that GEMM 2 of the previous chunk places in its MFMA gaps.  Element e of lane (r, q) of the resulting fragments:
    // column 4 q + e of tile 0 for e < 4, of tile 1 for e >= 4 - the order W2's rows were permuted to.
    bb_fence8(acc1, acx1, true);
    stamp(3);
    float ev[8][2], ee[8][2];
    uint32_t ehp[8], elp[8];
    constexpr float LOG2E = 1.4426950408889634f;
    // (the stages are pure arithmetic: nothing but their operands orders them, and left alone the instruction selector gathers
    // a pair's stages in front of their first use - the empty asm statements tie each stage's results to its place in the stream)
    auto hold2 = [&](float& a, float& b) __attribute__((always_inline)) { asm volatile("" : "+v"(a), "+v"(b)); };
    auto micro = [&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, q = m >> 3, st = m & 7, rt = q >> 2, ct = (q >> 1) & 1, i0 = 2 * (q & 1);
      if constexpr ((BB_DIAG & 4) != 0) {
        if constexpr (st == 7) {
          ehp[q] = __float_as_uint(acc1[rt][ct][i0]);
          elp[q] = __float_as_uint(acx1[rt][ct][i0 + 1]);
        }
      } else if constexpr (st == 0) {
        asm volatile("" : "+v"(acc1[rt][ct]), "+v"(acx1[rt][ct]));
        ev[q][0] = fmaf(acx1[rt][ct][i0], xw1, acc1[rt][ct][i0] * inv1);
        ev[q][1] = fmaf(acx1[rt][ct][i0 + 1], xw1, acc1[rt][ct][i0 + 1] * inv1);
        hold2(ev[q][0], ev[q][1]);
      } else if constexpr (st == 1) {
        ee[q][0] = ev[q][0] * LOG2E;
        ee[q][1] = ev[q][1] * LOG2E;
        hold2(ee[q][0], ee[q][1]);
      } else if constexpr (st == 2) {
        ee[q][0] = __builtin_amdgcn_exp2f(ee[q][0]);
        ee[q][1] = __builtin_amdgcn_exp2f(ee[q][1]);
        hold2(ee[q][0], ee[q][1]);
      } else if constexpr (st == 3) {
        ev[q][0] = ev[q][0] > 0.f ? ev[q][0] : ee[q][0] - 1.0f;
        hold2(ev[q][0], ee[q][1]);
      } else if constexpr (st == 4) {
        ev[q][1] = ev[q][1] > 0.f ? ev[q][1] : ee[q][1] - 1.0f;
        hold2(ev[q][0], ev[q][1]);
      } else if constexpr (st == 5) {
        ev[q][0] *= sC;
        ev[q][1] *= sC;
        ehp[q] = pack_f16(ev[q][0], ev[q][1]);
        hold2(ev[q][0], ev[q][1]);
        asm volatile("" : "+v"(ehp[q]));
      } else if constexpr (st == 6) {
        const f16x2 hh = __builtin_bit_cast(f16x2, ehp[q]);
        ee[q][0] = (float)hh.x;
        ee[q][1] = (float)hh.y;
        hold2(ee[q][0], ee[q][1]);
      } else {
        elp[q] = pack_f16((ev[q][0] - ee[q][0]) * 2048.f, (ev[q][1] - ee[q][1]) * 2048.f);
        asm volatile("" : "+v"(elp[q]));
      }
      BB_PIN;
    };
    constexpr int MS = 64 / (8 * G2) > 0 ? 64 / (8 * G2) : 1;       // micro-steps per gap (8 gaps per group)
    static_assert(MS * 8 * G2 >= 64, "every micro-step has a gap");

    z_req(std::integral_constant<int, ZG1>{});
    stamp(4);

    // ---- GEMM 2 of the PREVIOUS chunk (its k-step of [32 rows] x [N2]; two column tiles per group), this chunk's arithmetic in its gaps
    // (k-steps ZG1+1 .. KT-1 of the next head's Z behind every ZS-th group, the last one early enough to land before its use)
    constexpr int ZREM = KT - ZG1 - 1 > 0 ? KT - ZG1 - 1 : 0, ZS = ZREM > 0 ? (G2 / ZREM > 0 ? G2 / ZREM : 1) : 1;
    gemm2(sa_prev, f,
          [&](auto nc, auto kc) __attribute__((always_inline)) {
            bb_static_for<MS>([&](auto uc) __attribute__((always_inline)) {
              constexpr int m = (decltype(nc)::value * 8 + decltype(kc)::value) * MS + decltype(uc)::value;
              if constexpr (m < 64) micro(std::integral_constant<int, m>{});
            });
          },
          [&](auto nc) __attribute__((always_inline)) {
            constexpr int np = decltype(nc)::value;
            if constexpr (ZREM > 0 && np % ZS == 0 && np / ZS < ZREM) z_req(std::integral_constant<int, ZG1 + 1 + np / ZS>{});
          });
    // the intermediate of this chunk: GEMM 2's operand in the next step
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      mh[rt] = __builtin_bit_cast(f16x8, u32x4{ehp[4 * rt], ehp[4 * rt + 1], ehp[4 * rt + 2], ehp[4 * rt + 3]});
      ml[rt] = __builtin_bit_cast(f16x8, u32x4{elp[4 * rt], elp[4 * rt + 1], elp[4 * rt + 2], elp[4 * rt + 3]});
    }
    BB_PIN;
  };

  // (two step bodies only - with more variants in the loop nest the allocator spilled GEMM 2's 256 accumulators; nj >= 2)
  int s = 0;
  for (int hs = 0; hs < G.H; ++hs) {
    const int h = head_of(hs);
    const int h_next = hs + 1 < G.H ? head_of(hs + 1) : -1;
    for (int j = 0; j + 1 < nj; ++j, ++s) step(s, h, j, h_next, j == 0 && hs > 0, std::false_type{});
    step(s, h, nj - 1, h_next, false, std::true_type{});
    ++s;
  }
  // ---- drain: GEMM 2 of the last chunk (its W2 part was requested by the last step into that step's slot)
  {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    BB_PIN;
    const uint32_t sap = sa0 + ((s - 1) & 1) * SLOT;
    f16x8 fd[2][4];
    rd2(sap, fd[G1 & 1], c0, c0);
    rd2(sap, fd[G1 & 1], c0, c1);
    rd2(sap, fd[G1 & 1], c0, c2);
    rd2(sap, fd[G1 & 1], c0, c3);
    BB_PIN;
    gemm2(sap, fd, [&](auto, auto) __attribute__((always_inline)) {}, [&](auto) __attribute__((always_inline)) {});
  }
  bb_wait_vm<0>();        // the last step's re-requests: landed before the LDS is released

  if constexpr (stamp_on) {
    asm volatile("s_nop 0" ::"v"(acc2[0][0]), "v"(acx2[1][NT2 - 1]));
    stamp(5);
    if (lane == 0 && wave == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(&bb_stamps[i], st_acc[i]);
    }
  }
  // ---- epilogue: out[row][16 nt + 4 q .. + 3]
  const bool full = m0 + BB_ROWS <= G.M;
  float* crow = G.C + (int64_t)rowA * G.ldc + colq;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const bool ok = full || rowA + 16 * rt < G.M;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      f32x4 o;
      o.x = fmaf(acx2[rt][nt][0], xw2, acc2[rt][nt][0] * inv2);
      o.y = fmaf(acx2[rt][nt][1], xw2, acc2[rt][nt][1] * inv2);
      o.z = fmaf(acx2[rt][nt][2], xw2, acc2[rt][nt][2] * inv2);
      o.w = fmaf(acx2[rt][nt][3], xw2, acc2[rt][nt][3] * inv2);
      if (G.act2 == 2) {
        o.x = o.x > 0.f ? o.x : G.slope * o.x;
        o.y = o.y > 0.f ? o.y : G.slope * o.y;
        o.z = o.z > 0.f ? o.z : G.slope * o.z;
        o.w = o.w > 0.f ? o.w : G.slope * o.w;
      }
      if (ok) st4(crow + (int64_t)rt * 16 * G.ldc + 16 * nt, o);
    }
  }
}

template <int KT, int NT2, bool BIAS1>
int launch_b2b_b(const B2BArgs& G, hipStream_t st) {
  const int lds_bytes = 2 * ((2 * 2 * KT + 2 * NT2) * 1024) + (BIAS1 ? G.H * G.N1 * 4 : 0);
  static int set = 0;
  auto fn = proj_fuse_kernel<KT, NT2, BIAS1>;
  if (lds_bytes > 160 * 1024) return fail(-1, "proj_fuse: %d B of LDS needed (H * N1 too large for the bias stage)", lds_bytes);
  if (set < lds_bytes) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return fail((int)e, "proj_fuse: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(e));
    set = lds_bytes;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)((G.M + BB_ROWS - 1) / BB_ROWS)), dim3(256), lds_bytes, st, G);
  return check_launch("proj_fuse_kernel");
}
template <int KT, int NT2>
int launch_b2b(const B2BArgs& G, hipStream_t st) {
  return G.bias1 ? launch_b2b_b<KT, NT2, true>(G, st) : launch_b2b_b<KT, NT2, false>(G, st);
}

}  // namespace

}  // namespace disgat

extern "C" int disgat_proj_fuse(const uint16_t* Z_hi, const uint16_t* Z_lo, int64_t ldz, int64_t z_head_stride, const float* z_bound,
                                const uint16_t* W_chunks, const float* w1_scale, const float* bias1, const float* w2_scale,
                                const float* bias2, const float* mid_bound, float* C, int64_t ldc, int M, int H, int K1, int N1,
                                int N2, int act2, float slope, disgat_stream_t stream) {
  using namespace disgat;
  if (M == 0) return 0;
  DISGAT_REQUIRE(Z_hi && Z_lo && z_bound && W_chunks && w1_scale && w2_scale && mid_bound && C && M > 0 && H > 0,
                 "proj_fuse: null pointer / bad sizes");
  DISGAT_REQUIRE((K1 == 64 || K1 == 128 || K1 == 256) && N1 >= 64 && N1 % 32 == 0 && N1 <= 256 && (N2 == 64 || N2 == 128 || N2 == 256),
                 "proj_fuse: K1=%d must be 64 / 128 / 256, N1=%d a multiple of 32 in [64, 256], N2=%d one of 64 / 128 / 256", K1, N1, N2);
  DISGAT_REQUIRE(ldz % 8 == 0 && z_head_stride % 8 == 0 && aligned16(Z_hi) && aligned16(Z_lo) && aligned16(W_chunks),
                 "proj_fuse: plane rows must be 16-byte aligned (ldz, head stride multiples of 8 halfs)");
  DISGAT_REQUIRE(ldc % 4 == 0 && aligned16(C) && (!bias1 || aligned16(bias1)) && (!bias2 || aligned16(bias2)),
                 "proj_fuse: C rows and the biases must be 16-byte aligned");
  DISGAT_REQUIRE(act2 == 0 || act2 == 2, "proj_fuse: act2 must be 0 (none) or 2 (leaky relu)");
  B2BArgs G{Z_hi, Z_lo, ldz, z_head_stride, z_bound, W_chunks, w1_scale, bias1, w2_scale, bias2, mid_bound,
            C, ldc, M, H, N1, act2, slope};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int kt = K1 / 32, nt2 = N2 / 16;
#define BB_CASE(KT_, NT_) \
  if (kt == KT_ && nt2 == NT_) return launch_b2b<KT_, NT_>(G, st);
  BB_CASE(8, 16)
  BB_CASE(4, 8)
  BB_CASE(2, 4)
  BB_CASE(4, 16)
  BB_CASE(8, 8)
#undef BB_CASE
  return fail(-1, "proj_fuse: no instantiation for K1=%d, N2=%d", K1, N2);
}

// Diagnostic (a -DBB_DIAG=32 build, + ablation bits): s_memtime ticks per step phase summed over wave 0 of every
// workgroup, [8]: vmcnt wait, barrier, accumulator seed + first reads, GEMM 1, ELU / split, GEMM 2 (+ loop tail).
// Synchronises the device.
extern "C" int disgat_debug_stamps_b2b(unsigned long long* out16, int reset) {
  using namespace disgat;
  if (hipDeviceSynchronize() != hipSuccess) return fail(-1, "debug_stamps_b2b: sync failed");
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(bb_stamps), sizeof(bb_stamps)) != hipSuccess) return fail(-1, "debug_stamps_b2b: copy failed");
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(bb_stamps), z, sizeof(z)) != hipSuccess) return fail(-1, "debug_stamps_b2b: reset failed");
  }
  return 0;
}
