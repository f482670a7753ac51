// SSL pair sampler: SupEdgeTrainer.sample_train / GeneratedEdgeTrainer.sample_train of the reference
// (/root/reference/pretrainer.py:683-707, 524-576) without the dense N x N tensors.
//
// The reference builds   mask = (rand(N, N) < 3 rho)  |  {first third of the shuffled positives},   rho = n_pos / N^2,
// and returns mask.nonzero() (row-major order) with labels = (adj != 0) at those entries.  The same distribution, entry for
// entry, drawn in O(output) work on the device with no global sort, no searchsorted and no host number:
//   * the Bernoulli(3 rho) part is a Bernoulli process along every row, and a Bernoulli process restricted to disjoint column
//     intervals is independent per interval.  A work item is one interval (row, [col_lo, col_hi)) holding at most kPCap
//     positives and an expected <= ~96 random entries; ONE WAVE walks it by geometric skipping: every lane draws a gap
//     G = floor(log(U) / log(1 - p)) from a counter-based generator (Philox2x32-10), an in-wave inclusive scan of G + 1 gives
//     64 sorted, distinct columns per round - exactly the iid Bernoulli(p) process, already sorted and de-duplicated;
//   * "a third of the shuffled positives" is a uniform subset of EXACTLY floor(n_pos / 3) positives.  Positive j is in it
//     iff pi(j) < n_pos / 3 for a keyed pseudo-random PERMUTATION pi of [0, n_pos): an 8-round alternating Feistel network on
//     Z_a x Z_b, a = 2^k ~ sqrt(n_pos), b = ceil(n_pos / a) (xor on the power-of-two side, addition mod b on the other), with
//     cycle walking over the < a surplus values (one in ~sqrt(n_pos)): a bijection, so the count is exact, with no pass over
//     the positives and no state;
//   * the wave merges the two sorted sets by rank (own index + lower bound in the other set, both sets in LDS), labels fall
//     out of the merge (a random column that coincides with any positive of the row has label 1);
//   * output offsets come from one prefix sum over per-block counts: count pass -> one-block scan -> emit pass (the emit pass
//     regenerates the same draws from the same counters; nothing is stored in between but 4 bytes per item).
// The generator's (seed, step) live on the device and the scan kernel advances the step: a train_step captured in a HIP graph
// draws a fresh list on every replay without any host input.
//
// log() is evaluated by an explicit sequence of IEEE double operations (no library call, no fused multiply-add), so that
// oracle/sampler_oracle.py reproduces every draw bit for bit (integer outputs: the parity bar is equality).
#include "disgat_api.h"
#include "disgat_common.h"

namespace {

constexpr int kPCap = DISGAT_SAMPLE_PCAP;    // positives per item (the host's item builder cuts rows accordingly)
constexpr int kRCap = DISGAT_SAMPLE_RCAP;    // random columns per item kept (expected <= DISGAT_SAMPLE_RMEAN: never reached in practice)
constexpr int kWaves = 4;
constexpr int kFeistelRounds = 8;
constexpr uint32_t kFlag = 0x80000000u;       // rbuf entry: bit 31 = "this random column is a positive of the row"
static_assert(kPCap == 256 && kRCap == 256, "the chunk loops below assume four 64-entry chunks");

struct Item {            // int32 [n_items][8]
  int row, clo, chi, plo, phi, r0, r1, r2;
};

struct Keys {
  uint32_t k0, k1;
  uint32_t fk[kFeistelRounds];
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  return h ^ (h >> 16);
}

__device__ __forceinline__ Keys make_keys(uint64_t seed, uint64_t step) {
  Keys k;
  const uint64_t m = splitmix64(seed + step * 0x9E3779B97F4A7C15ull);
  k.k0 = (uint32_t)m;
  k.k1 = (uint32_t)(m >> 32);
#pragma unroll
  for (int r = 0; r < kFeistelRounds; ++r) k.fk[r] = (uint32_t)splitmix64(m + (uint64_t)(r + 1));
  return k;
}

// Philox2x32-10 (Salmon et al. 2011): 64-bit counter (c0, c1), 32-bit key; returns both output words as one 64-bit value.
__device__ __forceinline__ uint64_t philox_u64(uint32_t c0, uint32_t c1, uint32_t key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p = (uint64_t)0xD256D193u * c0;
    c0 = (uint32_t)(p >> 32) ^ key ^ c1;
    c1 = (uint32_t)p;
    key += 0x9E3779B9u;
  }
  return ((uint64_t)c0 << 32) | c1;
}

// ln(u) for u in (0, 1], as a fixed sequence of correctly rounded double operations: u = m 2^e with m in [sqrt(1/2), sqrt 2),
// s = (m - 1) / (m + 1), ln m = 2 s (1 + z/3 + z^2/5 + ... + z^7/15), z = s^2 <= 0.0295 (truncation < 4e-14 relative: a gap
// changes where log(u) / log(1 - p) lies that close to an integer - one draw in 10^13).
// No contraction into fused multiply-adds from here to the end of the file: the oracle rounds after every operation.
#pragma clang fp contract(off)
__device__ __forceinline__ double det_log(double u) {
  int e;
  double m = frexp(u, &e);                       // m in [0.5, 1): exact
  if (m < 0.70710678118654752440) {
    m = m * 2.0;
    e -= 1;
  }
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
  double p = 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z + 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z + 1.0 / 3.0;
  p = p * z + 1.0;
  const double lm = (2.0 * s) * p;
  return (double)e * 0.69314718055994530942 + lm;
}

// columns skipped + 1 before the next switched-on entry, capped at `cap` (= interval length + 1: "none left")
__device__ __forceinline__ uint32_t draw_increment(uint64_t bits64, double inv_log1m_p, uint32_t cap) {
  const uint64_t b53 = bits64 >> 11;
  const double u = ((double)b53 + 0.5) * 0x1p-53;             // (0, 1]
  const double g = floor(det_log(u) * inv_log1m_p);           // >= 0 (both factors <= 0)
  return g >= (double)(cap - 1u) ? cap : (uint32_t)g + 1u;
}

__device__ __forceinline__ uint32_t scan_sat(uint32_t v, uint32_t cap, int lane) {   // inclusive, min(sum, cap); cap <= 2^30 + 1
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v = min(v + t, cap);
  }
  return v;
}

// One pass of the permutation network over Z_a x Z_b (v = R * a + L, a = 2^ka, b = fb): even rounds L ^= F(R) mod a, odd rounds
// R = (R + F(L) mod b) mod b, the reduction of the 32-bit F to [0, b) by multiply-shift.  Every round is invertible.
__device__ __forceinline__ uint32_t feistel(uint32_t v, const Keys& k, int ka, uint32_t fb) {
  const uint32_t amask = (1u << ka) - 1u;
  uint32_t L = v & amask, R = v >> ka;
#pragma unroll
  for (int r = 0; r < kFeistelRounds; r += 2) {
    L ^= fmix32(R + k.fk[r]) & amask;
    R += __umulhi(fmix32(L + k.fk[r + 1]), fb);
    R = R >= fb ? R - fb : R;
  }
  return (R << ka) | L;
}

__device__ __forceinline__ bool selected(uint32_t j, const Keys& k, int ka, uint32_t fb, uint32_t n_pos, uint32_t n_sel) {
  uint32_t v = j;
  do v = feistel(v, k, ka, fb);                   // cycle walking: j < n_pos lies on a cycle that re-enters [0, n_pos)
  while (v >= n_pos);
  return v < n_sel;
}

__device__ __forceinline__ void wave_lds_sync() {  // LDS operations of one wave execute in issue order: only the compiler needs telling
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// number of entries of the sorted a[0..n) whose low 31 bits are < v
__device__ __forceinline__ int lower_bound_lds(const uint32_t* a, int n, uint32_t v) {
  int lo = 0;
  for (int s = n > 0 ? 1 << (31 - __builtin_clz(n)) : 0; s >= 1; s >>= 1) {
    const int m = lo + s;
    if (m <= n && (a[m - 1] & ~kFlag) < v) lo = m;
  }
  return lo;
}

struct Params {
  const Item* items;
  int n_items, ipw;
  const int32_t* pos_col;
  uint32_t n_pos, n_sel;
  int fka;                     // the permutation's domain: Z_(2^fka) x Z_fb >= n_pos
  uint32_t fb;
  double inv_log1m_p;          // 1 / log(1 - p); has_random = 0 when p <= 0
  int has_random;
  int64_t* meta;               // [0] seed  [1] step  [2] step of the last plan  [3] length of the last list  [4] events: item over capacity  [5] events: list over capacity
};

// One work item by one wave.  EMIT = false: returns the item's output count.  EMIT = true: writes its outputs from `base` on.
template <bool EMIT>
__device__ __forceinline__ int process_item(const Params& pr, const Keys& keys, int item_id, uint32_t* pc, uint32_t* rb, int64_t base,
                                            int64_t capacity, int64_t* __restrict__ idx_out, float* __restrict__ lab_out, bool& over) {
  const int lane = threadIdx.x & 63;
  const Item it = pr.items[item_id];
  int P = it.phi - it.plo;
  if (P > kPCap) {                // an item table this kernel was not built for: stay inside LDS, report
    P = kPCap;
    over = true;
  }
  // ---- the row's positives in this interval: columns to LDS, "in the selected third" as one ballot per 64
  uint64_t selm[4] = {0, 0, 0, 0};
  uint32_t myc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k * 64 < P) {
      const int q = k * 64 + lane;
      bool sel = false;
      if (q < P) {
        const uint32_t j = (uint32_t)it.plo + q;
        const uint32_t c = (uint32_t)pr.pos_col[j];
        pc[q] = c;
        myc[k] = c;
        if (pr.n_sel) sel = selected(j, keys, pr.fka, pr.fb, pr.n_pos, pr.n_sel);
      }
      selm[k] = __ballot(sel);
    }
  }
  // ---- the Bernoulli process over [clo, chi): sorted distinct columns, 64 per round
  int nR = 0;
  if (pr.has_random) {
    uint32_t first = (uint32_t)it.clo;
    const uint32_t chi = (uint32_t)it.chi;
    const uint32_t cap = chi - first + 1u;
    for (uint32_t round = 0; first < chi; ++round) {
      const uint64_t bits = philox_u64((uint32_t)item_id ^ keys.k1, round * 64u + lane, keys.k0);
      const uint32_t inc = draw_increment(bits, pr.inv_log1m_p, cap);
      const uint32_t incl = scan_sat(inc, cap, lane);
      const bool valid = incl <= chi - first;
      const uint32_t x = first - 1u + incl;
      const int nv = __popcll(__ballot(valid));
      if (valid && nR + lane < kRCap) rb[nR + lane] = x;
      if (nR + nv > kRCap) {
        nR = kRCap;
        over = true;
        break;
      }
      nR += nv;
      if (nv < 64) break;
      first = (uint32_t)__builtin_amdgcn_readlane((int)x, 63) + 1u;
    }
  }
  wave_lds_sync();
  // ---- selected positives that the random part did not already switch on (S'), and the positives' marks on the random part
  uint64_t keepm[4] = {0, 0, 0, 0};
  int lbv[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k * 64 < P) {
      const int q = k * 64 + lane;
      const bool sel = (selm[k] >> lane) & 1ull;
      bool keep = false;
      if (q < P && (EMIT || sel)) {
        const uint32_t c = myc[k];
        const int lb = lower_bound_lds(rb, nR, c);
        const bool in_r = lb < nR && (rb[lb] & ~kFlag) == c;
        if (EMIT && in_r) rb[lb] = c | kFlag;
        keep = sel && !in_r;
        lbv[k] = lb;
      }
      keepm[k] = __ballot(keep);
    }
  }
  const int s1 = __popcll(keepm[0]), s2 = s1 + __popcll(keepm[1]), s3 = s2 + __popcll(keepm[2]), s4 = s3 + __popcll(keepm[3]);
  if constexpr (!EMIT) return nR + s4;
  wave_lds_sync();
  const int64_t row = it.row;
  auto put = [&](int64_t pos, uint32_t col, float lab) {
    if (pos < capacity) {
      idx_out[pos] = row;
      idx_out[capacity + pos] = (int64_t)col;
      lab_out[pos] = lab;
    }
  };
  const uint64_t below = (1ull << lane) - 1ull;
  const int spre[4] = {0, s1, s2, s3};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((keepm[k] >> lane) & 1ull) put(base + spre[k] + __popcll(keepm[k] & below) + lbv[k], myc[k], 1.0f);
  for (int i = lane; i < nR; i += 64) {
    const uint32_t w = rb[i];
    const uint32_t x = w & ~kFlag;
    const int q = lower_bound_lds(pc, P, x);
    const int kq = q >> 6;
    const uint64_t km = kq == 0 ? keepm[0] : kq == 1 ? keepm[1] : kq == 2 ? keepm[2] : kq == 3 ? keepm[3] : 0ull;
    const int sp = kq == 0 ? 0 : kq == 1 ? s1 : kq == 2 ? s2 : kq == 3 ? s3 : s4;
    put(base + i + sp + __popcll(km & ((1ull << (q & 63)) - 1ull)), x, (w & kFlag) ? 1.0f : 0.0f);
  }
  wave_lds_sync();                // the next item of this wave reuses pc / rb
  return nR + s4;
}

__global__ __launch_bounds__(64 * kWaves) void pair_sample_count_kernel(const Params pr, int32_t* __restrict__ item_cnt,
                                                                        int32_t* __restrict__ blk_cnt) {
  __shared__ uint32_t lds[kWaves][kPCap + kRCap];
  __shared__ int wsum[kWaves];
  const int wave = disgat::rfl(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const Keys keys = make_keys((uint64_t)pr.meta[0], (uint64_t)pr.meta[1]);
  const int first = (blockIdx.x * kWaves + wave) * pr.ipw;
  int total = 0;
  bool over = false;
  for (int t = 0; t < pr.ipw && first + t < pr.n_items; ++t) {
    const int c = process_item<false>(pr, keys, first + t, lds[wave], lds[wave] + kPCap, 0, 0, nullptr, nullptr, over);
    if (lane == 0) item_cnt[first + t] = c;
    total += c;
  }
  if (lane == 0) wsum[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// One block: exclusive prefix of the per-block counts, the list length, and the generator's step.
__global__ __launch_bounds__(1024) void pair_sample_scan_kernel(const int32_t* __restrict__ blk_cnt, int n_blocks,
                                                                int64_t* __restrict__ blk_off, int64_t* __restrict__ meta,
                                                                int64_t capacity, double* __restrict__ count_out) {
  __shared__ int64_t wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n_blocks + 1023) / 1024;
  const int b0 = min(tid * per, n_blocks), b1 = min(b0 + per, n_blocks);
  int64_t mine = 0;
  for (int b = b0; b < b1; ++b) mine += blk_cnt[b];
  int64_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int64_t before = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) before += wtot[w];
    total += wtot[w];
  }
  int64_t run = before + incl - mine;
  for (int b = b0; b < b1; ++b) {
    blk_off[b] = run;
    run += blk_cnt[b];
  }
  if (tid == 0) {
    const int64_t kept = total < capacity ? total : capacity;
    meta[3] = kept;
    if (total > capacity) meta[5] += 1;
    if (count_out) *count_out = (double)kept;
    meta[2] = meta[1];
    meta[1] = meta[1] + 1;
  }
}

__global__ __launch_bounds__(64 * kWaves) void pair_sample_emit_kernel(const Params pr, const int32_t* __restrict__ item_cnt,
                                                                       const int64_t* __restrict__ blk_off, int n_item_blocks,
                                                                       int64_t capacity, int64_t* __restrict__ idx_out,
                                                                       float* __restrict__ lab_out, int64_t pad_row, int64_t pad_col) {
  __shared__ uint32_t lds[kWaves][kPCap + kRCap];
  const int wave = disgat::rfl(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= n_item_blocks) {         // the tail of a fixed-capacity list: padding pairs, label -1
    const int64_t total = pr.meta[3];
    const int64_t stride = (int64_t)(gridDim.x - n_item_blocks) * blockDim.x;
    for (int64_t p = total + (int64_t)(blockIdx.x - n_item_blocks) * blockDim.x + threadIdx.x; p < capacity; p += stride) {
      idx_out[p] = pad_row;
      idx_out[capacity + p] = pad_col;
      lab_out[p] = -1.0f;
    }
    return;
  }
  const Keys keys = make_keys((uint64_t)pr.meta[0], (uint64_t)pr.meta[2]);
  const int blk_first = blockIdx.x * kWaves * pr.ipw;
  int c = (lane < kWaves * pr.ipw && blk_first + lane < pr.n_items) ? item_cnt[blk_first + lane] : 0;
  int incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  const int excl = incl - c;
  const int64_t base0 = blk_off[blockIdx.x];
  bool over = false;
  for (int t = 0; t < pr.ipw; ++t) {
    const int slot = wave * pr.ipw + t;
    if (blk_first + slot >= pr.n_items) break;
    const int64_t base = base0 + __builtin_amdgcn_readlane(excl, slot);
    process_item<true>(pr, keys, blk_first + slot, lds[wave], lds[wave] + kPCap, base, capacity, idx_out, lab_out, over);
  }
  if (over && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(pr.meta + 4), 1ull);
}

int make_params(Params& pr, const int32_t* items, int n_items, int items_per_wave, const int32_t* pos_col, int64_t n_pos, int64_t n_sel,
                double p, int64_t* meta, const char* who) {
  DISGAT_REQUIRE(items && n_items > 0 && meta, "%s: null / empty item table", who);
  DISGAT_REQUIRE(items_per_wave >= 1 && items_per_wave * kWaves <= 64, "%s: items_per_wave %d outside [1, 16]", who, items_per_wave);
  DISGAT_REQUIRE(n_pos >= 0 && n_pos < (int64_t(1) << 31) && n_sel >= 0 && n_sel <= n_pos, "%s: n_pos %lld / n_sel %lld", who,
                 (long long)n_pos, (long long)n_sel);
  DISGAT_REQUIRE(n_pos == 0 || pos_col, "%s: null positive columns", who);
  DISGAT_REQUIRE(p >= 0.0 && p <= 1.0, "%s: probability %g", who, p);
  pr.items = reinterpret_cast<const Item*>(items);
  pr.n_items = n_items;
  pr.ipw = items_per_wave;
  pr.pos_col = pos_col;
  pr.n_pos = (uint32_t)n_pos;
  pr.n_sel = (uint32_t)n_sel;
  int bits = 0;
  while ((int64_t(1) << bits) < n_pos) ++bits;           // a = 2^(bits / 2) ~ sqrt(n_pos), b = ceil(n_pos / a): a b - n_pos < a
  pr.fka = bits / 2;
  pr.fb = n_pos > 0 ? (uint32_t)((n_pos + (int64_t(1) << pr.fka) - 1) >> pr.fka) : 1u;
  pr.has_random = p > 0.0;
  pr.inv_log1m_p = p > 0.0 ? 1.0 / log1p(-p) : 0.0;     // p = 1: -0.0, every gap is 0
  pr.meta = meta;
  return 0;
}

}  // namespace

extern "C" int disgat_pair_sample_plan(const int32_t* items, int n_items, int items_per_wave, const int32_t* pos_col, int64_t n_pos,
                                       int64_t n_sel, double p, int64_t capacity, int64_t* meta, int32_t* item_count,
                                       int32_t* block_count, int64_t* block_off, double* count_out, disgat_stream_t stream) {
  Params pr;
  if (int rc = make_params(pr, items, n_items, items_per_wave, pos_col, n_pos, n_sel, p, meta, "pair_sample_plan")) return rc;
  DISGAT_REQUIRE(item_count && block_count && block_off && capacity >= 0, "pair_sample_plan: null scratch / negative capacity");
  const int per_block = kWaves * items_per_wave;
  const int n_blocks = (n_items + per_block - 1) / per_block;
  hipLaunchKernelGGL(pair_sample_count_kernel, dim3(n_blocks), dim3(64 * kWaves), 0, (hipStream_t)stream, pr, item_count, block_count);
  if (int rc = disgat::check_launch("pair_sample_count")) return rc;
  hipLaunchKernelGGL(pair_sample_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, block_count, n_blocks, block_off, meta,
                     capacity, count_out);
  return disgat::check_launch("pair_sample_scan");
}

extern "C" int disgat_pair_sample_emit(const int32_t* items, int n_items, int items_per_wave, const int32_t* pos_col, int64_t n_pos,
                                       int64_t n_sel, double p, int64_t n_rows, int64_t n_cols, int64_t capacity, int64_t* meta,
                                       const int32_t* item_count, const int64_t* block_off, int64_t* idx_out, float* lab_out,
                                       int pad_tail, disgat_stream_t stream) {
  Params pr;
  if (int rc = make_params(pr, items, n_items, items_per_wave, pos_col, n_pos, n_sel, p, meta, "pair_sample_emit")) return rc;
  DISGAT_REQUIRE(item_count && block_off && capacity >= 0 && (capacity == 0 || (idx_out && lab_out)), "pair_sample_emit: null buffer");
  DISGAT_REQUIRE(n_rows >= 1 && n_cols >= 1 && n_cols <= (int64_t(1) << 30), "pair_sample_emit: %lld x %lld outside the envelope",
                 (long long)n_rows, (long long)n_cols);
  const int per_block = kWaves * items_per_wave;
  const int n_blocks = (n_items + per_block - 1) / per_block;
  const int pad_blocks = pad_tail ? 64 : 0;
  hipLaunchKernelGGL(pair_sample_emit_kernel, dim3(n_blocks + pad_blocks), dim3(64 * kWaves), 0, (hipStream_t)stream, pr, item_count,
                     block_off, n_blocks, capacity, idx_out, lab_out, n_rows - 1, n_cols - 1);
  return disgat::check_launch("pair_sample_emit");
}
