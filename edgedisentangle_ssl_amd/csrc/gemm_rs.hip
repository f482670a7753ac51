// fp32-accurate GEMM C = act(A B + bias + init) for K <= 256 with the A operand REGISTER-stationary: the f16x3 scheme of
// gemm_split.hip (two fp16 planes per operand, three products, fp32 accumulate) for the shapes of the DISGAT path whose
// result is 8x larger than their input - the score operands P = x W_top, Q = x W_bot ([N, 256] x [256, 2048],
// /root/reference/layers.py:374-379), the per-head projections and classifier layers (layers.py:397-399, models.py:538) and
// their data gradients in the backward.
//
// Why another kernel.  gemm_f16x3_as_kernel keeps a 128 x K A tile in LDS and reads its weight fragments from global memory
// inside the MFMA loop; per tile it spends ~50 us on MFMAs and ~49 us draining 1 MB of results at the CU's share of the HBM
// write rate - and the two do not overlap (3.0-3.5 ms at M = 1e6): the weight-fragment loads of the next column step queue
// behind the stores in the CU's in-order vector-memory path, and the tile's load-split-barrier prologue overlaps nothing.
// Measured (same box, interleaved, M = 1e6; tools/gemm_ab.sh): P/Q [256 -> 2048] 3.16-3.21 ms against 3.40-3.47, the 8-head
// projection [256 -> 256] 4.41-4.59 against 4.79-4.89, [64 -> 512] 0.51-0.55 against 0.60-0.62.  What bounds it
// (in-kernel stamps, tools/rs_stamps.sh; the chip holds ~1.5-1.7 GHz under this load, so the 3.0 M MFMA cycles per SIMD
// alone are 1.8-2.0 ms): per 32-column chunk the two waves of a SIMD need 2 x 96 MFMAs x 16 cycles = 3.1 k cycles of one
// shared matrix pipe and spend 5.3 k - 0.6-0.8 k each in their four stores (the queue is full: the CU's 32 KB per chunk drain
// in ~2.7 k cycles at its share of the HBM write rate), 0.2-0.5 k issuing DMA pieces, ~1 k of barrier skew (the younger half
// of the block loses the issue arbitration; static priority only moves the deficit to the other half).  Here
//   * a wave keeps its 32 rows of A as MFMA fragments in REGISTERS (2 row tiles x K/32 k-steps x hi/lo = 128 VGPRs at
//     K = 256), converted from fp32 once per 256-row block - no operand arithmetic in the loop, VALU : MFMA ~ 0.2;
//   * the weights stream through a 4-slot LDS ring of 32-column chunks by LDS-DMA (global_load_lds_dwordx4; the planes are
//     stored fragment-major - disgat_split_f16 - so a chunk is two contiguous runs and its LDS image is the memory image):
//     three chunks ahead of the arithmetic, so a DMA that queues behind stores costs nothing - the MFMA loop itself touches
//     LDS only.  All eight waves share every weight fragment: 256 rows per fetch from L2 instead of 128;
//   * per chunk a wave retires 96 MFMAs (12 per fragment pair) and four 16-byte stores per lane (16 rows x 64 B per
//     instruction), one raw s_barrier and one counted s_waitcnt vmcnt (DMAs and stores count together, in issue order);
//   * waves 4-7 retire a chunk one step later than waves 0-3, so one wave of every SIMD computes while its partner stores.
#include <type_traits>
#include <utility>

#include "disgat_api.h"
#include "gemm_common.h"

namespace disgat {

namespace {

constexpr int RS_BM = 256;        // rows per block: 8 waves x 2 row tiles of 16
constexpr int RS_CH = 32;         // output columns per chunk (2 n-tiles of 16)
constexpr int RS_NS = 4;          // ring slots: the chunk being read + 3 in flight
#ifndef RS_DIAG
#define RS_DIAG 0                 // 1: ablation switches compiled in (DISGAT_RS_DEBUG: 1 no stores, 2 no MFMAs, 4 no ring refills)
#endif

__device__ unsigned long long rs_stamps[16];      // RS_DIAG: s_memtime cycles per loop phase, [early wave 0 | late wave 4][phase]

template <int N>
__device__ __forceinline__ void rs_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* rs_lds_ptr_t;
typedef const __attribute__((address_space(3))) unsigned char* rs_lds_cptr_t;

// the four weight fragments (hi / lo x two n-tiles) of k-step T of the chunk at LDS byte address `sa` (+ the lane's 16 bytes)
template <int KT, int T>
__device__ __forceinline__ void rs_read4(f16x8 (&d)[4], uint32_t sa) {
  constexpr int PLANE_B = 2 * KT * 1024;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[0]) : "v"(sa), "n"((0 * KT + T) * 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[1]) : "v"(sa), "n"((1 * KT + T) * 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[2]) : "v"(sa), "n"(PLANE_B + (0 * KT + T) * 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[3]) : "v"(sa), "n"(PLANE_B + (1 * KT + T) * 1024));
}
// wait until at most N LDS operations are outstanding; the fragments pass through so that their users stay behind the wait
template <int N>
__device__ __forceinline__ void rs_wait_lds(f16x8 (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "n"(N));
}

template <int... I, class F>
__device__ __forceinline__ void static_for_rs_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for_rs(F&& f) {
  static_for_rs_impl(std::make_integer_sequence<int, N>{}, f);
}
typedef const __attribute__((address_space(1))) void* rs_glb_ptr_t;

__device__ __forceinline__ void rs_glds16(const uint16_t* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((rs_glb_ptr_t)src, (rs_lds_ptr_t)dst, 16, 0, 0);
}

// 8 consecutive fp32 of an A row (two 16-byte loads) -> the lane's hi / lo MFMA fragments
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, float s, f16x8& hi, f16x8& lo) {
  u32x2 h0, l0, h1, l1;
  split4h(v0 * s, h0, l0);
  split4h(v1 * s, h1, l1);
  const u32x4 h = {h0.x, h0.y, h1.x, h1.y}, l = {l0.x, l0.y, l1.x, l1.y};
  hi = *reinterpret_cast<const f16x8*>(&h);
  lo = *reinterpret_cast<const f16x8*>(&l);
}

template <int ACT, int KT>
__global__ __launch_bounds__(512, 2) void gemm_f16x3_rs_kernel(const GemmHArgs G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_rs[];
  constexpr int PLANE_B = 2 * KT * 1024;          // bytes of one plane of a chunk: 2 n-tiles x KT fragments of 1 KB
  constexpr int SLOT = 2 * PLANE_B;
  constexpr int P = (SLOT / 1024) / 8;            // DMA pieces (1 KB wave instructions) per chunk and wave: KT / 2
  constexpr int S = 4;                            // store instructions per chunk and wave
  static_assert(KT == 2 || KT == 4 || KT == 8, "K = 64, 128 or 256");
  static_assert(2 * P + 3 * S <= 63, "vmcnt is a 6-bit counter");
  const int lane = threadIdx.x & 63;
  const int wave = rfl(threadIdx.x >> 6);
  const int bz = blockIdx.y;
  // G.mt = column splits (launcher): few row blocks (the bundled graphs: 9-80 of them for 256 CUs) share the chunks of a
  // row block among `ns` workgroups, each with its own copy of the A fragments; 1 at any size that fills the chip by rows
  const int ns = G.mt;
  const int mb = ns == 1 ? (int)blockIdx.x : (int)blockIdx.x / ns;
  const int c_base = ns == 1 ? 0 : ((int)blockIdx.x - mb * ns) * (G.N / RS_CH / ns);
  const int m0 = mb * RS_BM;
  const int K = KT * 32;
  const int n_ch = G.N / RS_CH / ns;

  const float* A = G.A + (int64_t)bz * G.a_bs;
  const uint16_t* Bt = G.Bt + (int64_t)bz * 2 * G.N * K;
  float* C = G.C + (int64_t)bz * G.c_bs;
  const float* bias = G.bias ? G.bias + (int64_t)bz * G.N : nullptr;
  const float* init = G.init ? G.init + (int64_t)bz * G.i_bs : nullptr;
  const float sA = f16_scale(*G.a_amax);
  const float sAB = sA * *G.b_scale;
  const float inv = 1.0f / sAB;
  const float xw = inv * (1.0f / 2048.f);
  const bool full = m0 + RS_BM <= G.M;            // a ragged block may skip store instructions: it waits strictly

  // ---- weight ring: chunk c = the 2 KT fragment blocks of n-tiles 2c, 2c+1 of either plane (contiguous in memory)
  const int64_t b_plane = (int64_t)G.N * K;
  // blocks walk the chunks in rotated order: every block streams the same planes from its XCD's L2, and blocks that start
  // together would otherwise ask one L2 channel for the same lines at the same time
  const int rot = mb % n_ch;
  auto chunk_of = [&](int c) __attribute__((always_inline)) {
    const int cc = (c < n_ch ? c : n_ch - 1) + rot;                 // past the end: the last chunk again (the counts assume an issue per step)
    return c_base + (cc >= n_ch ? cc - n_ch : cc);
  };
  auto issue = [&](int c) __attribute__((always_inline)) {
    const int cc = chunk_of(c);
    unsigned char* dst = lds_rs + (c & (RS_NS - 1)) * SLOT;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int q = wave + 8 * i;                                   // piece: plane q / (2 KT), fragment block q % (2 KT)
      const int pl = q / (2 * KT), fb = q % (2 * KT);
      rs_glds16(Bt + pl * b_plane + ((int64_t)cc * 2 * KT + fb) * 512 + lane * 8, dst + q * 1024);
    }
  };
  issue(0);
  issue(1);
  issue(2);

  // ---- this wave's 32 rows of A -> fragments in registers.  Lane (r = lane & 15, q = lane >> 4) holds k = 32 t + 8 q .. + 7
  // of rows 16 rt + r: 32 contiguous bytes per k-step, the four q lanes of a row read one 128-byte line.
  f16x8 ah[2][KT], al[2][KT];
  {
    const int r0 = m0 + wave * 32 + (lane & 15);
    const float* ap[2] = {A + (int64_t)min(r0, G.M - 1) * G.lda + (lane >> 4) * 8,
                          A + (int64_t)min(r0 + 16, G.M - 1) * G.lda + (lane >> 4) * 8};     // rows past M: valid memory, never stored
    constexpr int TB = KT < 4 ? KT : 4;           // k-steps per batch of loads (16 VGPRs of fp32 each)
#pragma unroll
    for (int tb = 0; tb < KT; tb += TB) {
      f32x4 v[TB][2][2];
#pragma unroll
      for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          v[t][rt][0] = ld4(ap[rt] + (tb + t) * 32);
          v[t][rt][1] = ld4(ap[rt] + (tb + t) * 32 + 4);
        }
#pragma unroll
      for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) split8(v[t][rt][0], v[t][rt][1], sA, ah[rt][tb + t], al[rt][tb + t]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (the compiler waited for the A loads with vmcnt(0): the three chunks requested above have landed as well)

  const int fo = lane * 16;                         // a lane's 16 bytes of a fragment block
  const int rowA = m0 + wave * 32 + (lane & 15);    // + 16 rt
  const int colq = 4 * (lane >> 4);                 // + 32 c + 16 ct
  int64_t ldc = G.ldc;
  asm volatile("" : "+s"(ldc));
  float* crow = C + (int64_t)rowA * ldc + colq;

  // ---- main loop.  A chunk's 32 KB of results leave the CU at its share of the HBM write rate - about as long as the
  // chunk's MFMAs take - and a wave whose store cannot enter the full queue just waits.  So the two waves of a SIMD (w and
  // w + 4) take turns: waves 0-3 retire a chunk right after its MFMAs, waves 4-7 keep theirs in the accumulators across the
  // barrier and retire it at the START of the next step - while one of the pair sits in its stores the other has the matrix
  // pipe to itself (in lock-step both stored, then both computed: MFMA time + store time, 3.2 ms on the P/Q shape).
  const bool late = wave >= 4;
  f32x4v acc[2][2], acx[2][2];
  auto seed = [&](int cw) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        acc[rt][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
        acx[rt][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
    // (a variant compiled without this branch, whose first MFMA of a chunk takes a literal zero as its C operand instead of
    // accumulators zeroed by 32 v_mov, ran 3.25 ms against 3.19 on the P/Q shape, same box, interleaved)
    if (bias || init) {
      // bias and the additive matrix seed the hi*hi accumulator (times s_A s_B, a power of two: exact).  Their loads join the
      // vector-memory queue (the compiler waits for them, and with them for everything older): the counted wait stays valid
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int cj = cw * RS_CH + 16 * ct + colq;
        const f32x4 bv = bias ? ld4(bias + cj) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = rowA + 16 * rt;
          const f32x4 iv = (init && row < G.M) ? ld4(init + (int64_t)row * G.ldi + cj) : f32x4{0.f, 0.f, 0.f, 0.f};
          acc[rt][ct] = f32x4v{(bv.x + iv.x) * sAB, (bv.y + iv.y) * sAB, (bv.z + iv.z) * sAB, (bv.w + iv.w) * sAB};
        }
      }
    }
  };
  // k-steps [T0, T1) of the chunk in ring slot `slot`: transposed product (weight fragment first), so a lane ends with 4
  // consecutive columns of one row per 16 x 16 tile.  Per k-step 12 MFMAs on four accumulator pairs, ordered so that no
  // accumulator is touched again within 4 instructions (the compiler hoists the next k-step's fragment reads as registers allow).
  auto ksteps = [&](const unsigned char* slot, auto t0c, auto t1c) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
    // Fragment reads as inline asm with COUNTED waits: left to the compiler every wait for a k-step's fragments is
    // lgkmcnt(0), which also waits for the next k-step's reads issued just before - the prefetch never runs ahead and the
    // matrix pipe idles for an LDS round trip per step.  LDS reads return in order: with the next step's four reads in flight
    // the current step's have landed at lgkmcnt(4).
    const uint32_t sa = (uint32_t)(uintptr_t)(rs_lds_cptr_t)slot;
    f16x8 f[2][4];            // [buffer][bh0, bh1, bl0, bl1]
    rs_read4<KT, T0>(f[T0 & 1], sa);
    static_for_rs<T1 - T0>([&](auto ic) __attribute__((always_inline)) {
      constexpr int t = T0 + decltype(ic)::value;
      f16x8(&cur)[4] = f[t & 1];
      if constexpr (t + 1 < T1) {
        rs_read4<KT, t + 1>(f[(t + 1) & 1], sa);
        rs_wait_lds<4>(cur);
      } else {
        rs_wait_lds<0>(cur);
      }
      if (RS_DIAG && (G.nt & 2)) {
        asm volatile("" ::"v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]));
      } else {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[ct], ah[rt][t], acc[rt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acx[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[2 + ct], ah[rt][t], acx[rt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acx[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[ct], al[rt][t], acx[rt][ct], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
  const bool stamp_on = RS_DIAG && (G.nt & 32) != 0;
  auto stamp = [&](int ph) __attribute__((always_inline)) {
    if (!stamp_on) return;
    unsigned long long tt;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (ph >= 0) st_acc[ph] += tt - st_prev;
    st_prev = tt;
  };
  stamp(-1);
  auto retire = [&](int cw) __attribute__((always_inline)) {
    float* cp = crow + cw * RS_CH;
    f32x4 out[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        out[rt][ct].x = act_ct<ACT>(fmaf(acx[rt][ct][0], xw, acc[rt][ct][0] * inv), G.slope);
        out[rt][ct].y = act_ct<ACT>(fmaf(acx[rt][ct][1], xw, acc[rt][ct][1] * inv), G.slope);
        out[rt][ct].z = act_ct<ACT>(fmaf(acx[rt][ct][2], xw, acc[rt][ct][2] * inv), G.slope);
        out[rt][ct].w = act_ct<ACT>(fmaf(acx[rt][ct][3], xw, acc[rt][ct][3] * inv), G.slope);
      }
    if (RS_DIAG && stamp_on) {
      asm volatile("s_nop 0" ::"v"(out[0][0]), "v"(out[0][1]), "v"(out[1][0]), "v"(out[1][1]));
      stamp(7);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const bool ok = full || rowA + 16 * rt < G.M;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        if (RS_DIAG && (G.nt & 1)) asm volatile("" ::"v"(out[rt][ct]));
        else if (ok) st4(cp + (int64_t)rt * 16 * ldc + 16 * ct, out[rt][ct]);
      }
    }
  };
  constexpr std::integral_constant<int, 0> k_lo{};
  constexpr std::integral_constant<int, KT> k_hi{};

  for (int c = 0; c < n_ch; ++c) {
    // (1) this wave's pieces of chunk c have landed: behind them in the queue are D(c+1), D(c+2) and the stores of three
    // chunks - of two at c == 3 for waves 4-7, which retire nothing at step 0 (there the looser count would pass with D(3)
    // still in flight); (2) barrier: everybody's have, and everybody is done reading chunk c-1, whose slot D(c+3) refills
    stamp(5);
    if (!full || (RS_DIAG && (G.nt & 1))) rs_wait_vm<0>();      // (the no-store ablation issues no stores to count)
    else if (late && c == 3) rs_wait_vm<2 * P + 2 * S>();
    else rs_wait_vm<2 * P + 3 * S>();
    stamp(0);
    __builtin_amdgcn_s_barrier();
    stamp(1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(RS_DIAG && (G.nt & 4))) issue(c + 3);
    __builtin_amdgcn_sched_barrier(0);
    stamp(2);
    const int cw = chunk_of(c);
    if (late && c > 0) {
      retire(chunk_of(c - 1));
      stamp(6);
    }
    seed(cw);
    // (static priority for the younger half, waves 4-7, which loses the issue arbitration - k-steps 3.0 k cycles against 2.4 k -
    // only moves the deficit to the other half: 3.39 vs 3.20 ms; the pair shares one matrix pipe)
    stamp(3);
    ksteps(lds_rs + (c & (RS_NS - 1)) * SLOT + fo, k_lo, k_hi);
    if (stamp_on) asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acx[0][0]), "v"(acc[1][1]), "v"(acx[1][1]));
    stamp(4);
    if (!late) retire(cw);
  }
  if (late) retire(chunk_of(n_ch - 1));
  if (stamp_on && lane == 0 && (wave == 0 || wave == 4)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&rs_stamps[(wave >> 2) * 8 + i], st_acc[i]);
  }
  rs_wait_vm<0>();        // the ring still holds DMAs in flight (re-loads of the last chunk): they land before the LDS is released
}

template <int ACT>
int launch_rs_kt(const GemmHArgs& G, hipStream_t st) {
  const dim3 grid((unsigned)((G.M + RS_BM - 1) / RS_BM) * G.mt, G.batch);
  auto go = [&](auto fn, int kt) {
    const int lds_bytes = RS_NS * 2 * (2 * kt * 1024);
    static int lds_set[3][9] = {};
    if (lds_set[ACT][kt] < lds_bytes) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      if (e != hipSuccess) return fail((int)e, "gemm_f16x3 (rs): cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(e));
      lds_set[ACT][kt] = lds_bytes;
    }
    hipLaunchKernelGGL(fn, grid, dim3(512), lds_bytes, st, G);
    return check_launch("gemm_f16x3_rs_kernel");
  };
  switch (G.K) {
    case 64: return go(gemm_f16x3_rs_kernel<ACT, 2>, 2);
    case 128: return go(gemm_f16x3_rs_kernel<ACT, 4>, 4);
    default: return go(gemm_f16x3_rs_kernel<ACT, 8>, 8);
  }
}

}  // namespace

// K in {64, 128, 256}, N % 32 == 0; fragment-major weight planes.
bool gemm_rs_takes(int N, int K) { return (K == 64 || K == 128 || K == 256) && N % RS_CH == 0 && N >= RS_CH; }

int launch_gemm_f16x3_rs(const GemmHArgs& G0, hipStream_t st) {
  GemmHArgs G = G0;
  G.nt = (RS_DIAG && getenv("DISGAT_RS_DEBUG")) ? atoi(getenv("DISGAT_RS_DEBUG")) : 0;     // (nt is unused by this kernel)
  // column splits (G.mt, see the kernel): double while the grid stays within two rounds of the chip's 256 CUs x 2 resident blocks and a
  // block keeps at least 2 chunks (the ring's prologue requests 3)
  const int64_t row_blocks = (int64_t)((G.M + RS_BM - 1) / RS_BM) * G.batch;
  int ns = 1;
  for (int n_ch = G.N / RS_CH; n_ch % 2 == 0 && n_ch >= 4 && row_blocks * ns * 2 <= 1024; n_ch /= 2) ns *= 2;
  G.mt = ns;
  if (G.act == 1) return launch_rs_kt<1>(G, st);
  if (G.act == 2) return launch_rs_kt<2>(G, st);
  return launch_rs_kt<0>(G, st);
}

}  // namespace disgat

// Diagnostic (a -DRS_DIAG=1 build, DISGAT_RS_DEBUG & 32): cycles per loop phase of waves 0 and 4 of every block, [2][8]:
// vmcnt wait, barrier, DMA issue, accumulator seed, k-steps (LDS reads + MFMAs), retire (early: = 5, the tail to the next
// wait), retire of the previous chunk (late).  Synchronises the device.
extern "C" int disgat_debug_stamps_rs(unsigned long long* out16, int reset) {
  using namespace disgat;
  if (hipDeviceSynchronize() != hipSuccess) return fail(-1, "debug_stamps_rs: sync failed");
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(rs_stamps), sizeof(rs_stamps)) != hipSuccess) return fail(-1, "debug_stamps_rs: copy failed");
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rs_stamps), z, sizeof(z)) != hipSuccess) return fail(-1, "debug_stamps_rs: reset failed");
  }
  return 0;
}
