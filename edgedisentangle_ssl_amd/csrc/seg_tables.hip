// Work-item tables of the backward's segment passes (disgat_seg_grad_sign / _att3 / _hx) over a key-sorted pair list of FIXED
// capacity - the lists a train_step captured in a HIP graph scores (sampling.PairSampler.sample_static): no size is read
// back, every table has a fixed shape and its unused tail is padding (key -1: the segment kernels and disgat_seg_combine skip
// it).  The reference has no counterpart (autograd's index / scatter_add_ backward, pretrainer.py:752, 631); this replaces
// the ~35 ATen launches (searchsorted, cumsum x4, scatter x2, where, stack, ...) per list and side that built the same
// tables, which on Cora / chameleon-sized graphs was a third of a step's launches.
//
// A key with more than `chunk` list entries is cut into near-equal slices exactly as graph.build_items cuts rows: deg = entries
// of the key, n = ceil(deg / chunk), slice j covers  begin + j * (deg / n) + min(j, deg % n)  for  deg / n + (j < deg % n)
// entries and writes partial record `slot`; disgat_seg_combine adds a key's records in slice order (deterministic sums).
#include "disgat_api.h"
#include "disgat_common.h"

namespace {

// ptr[k] = number of list entries with key < k (k = 0 .. n_keys); the list is keys[perm[m]] (perm NULL: keys[m]), sorted.
// The same launch narrows the sort's int64 permutation to the int32 form the segment kernels take.
__global__ __launch_bounds__(256) void seg_ptr_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ perm, int64_t C,
                                                      int n_keys, int32_t* __restrict__ ptr, int32_t* __restrict__ perm32) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (perm32 && i < C) perm32[i] = (int32_t)perm[i];
  if (i > n_keys) return;
  int64_t lo = 0, hi = C;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int64_t v = keys[perm ? perm[mid] : mid];
    if (v < i) lo = mid + 1; else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

struct Sums {
  int items, split, slots;
};
__device__ __forceinline__ Sums operator+(Sums a, Sums b) { return Sums{a.items + b.items, a.split + b.split, a.slots + b.slots}; }

__device__ __forceinline__ Sums of_key(const int32_t* ptr, int k, int chunk) {
  const int deg = ptr[k + 1] - ptr[k];
  const int n = deg > chunk ? (deg + chunk - 1) / chunk : 1;
  return Sums{n, n > 1 ? 1 : 0, n > 1 ? n : 0};
}

// One block: exclusive prefixes over the keys of (items, split keys, slots of split keys) -> off[k] = first item of key k,
// the split-key tables (rows / ptr, -1 / total padded) and the three totals.
__global__ __launch_bounds__(1024) void seg_scan_kernel(const int32_t* __restrict__ ptr, int n_keys, int chunk, int32_t* __restrict__ off,
                                                        int32_t* __restrict__ split_rows, int32_t* __restrict__ split_ptr,
                                                        int n_split_cap, int32_t* __restrict__ totals) {
  __shared__ Sums wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n_keys + 1023) / 1024;
  const int k0 = min(tid * per, n_keys), k1 = min(k0 + per, n_keys);
  Sums mine{0, 0, 0};
  for (int k = k0; k < k1; ++k) mine = mine + of_key(ptr, k, chunk);
  Sums incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const Sums t{__shfl_up(incl.items, d, 64), __shfl_up(incl.split, d, 64), __shfl_up(incl.slots, d, 64)};
    if (lane >= d) incl = incl + t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  Sums run{0, 0, 0}, total{0, 0, 0};
  for (int w = 0; w < 16; ++w) {
    if (w < wave) run = run + wtot[w];
    total = total + wtot[w];
  }
  run = Sums{run.items + incl.items - mine.items, run.split + incl.split - mine.split, run.slots + incl.slots - mine.slots};
  for (int k = k0; k < k1; ++k) {
    const Sums s = of_key(ptr, k, chunk);
    off[2 * k] = run.items;
    off[2 * k + 1] = s.split ? run.slots : -1;          // first slot of a split key
    if (s.split && run.split < n_split_cap) {
      split_rows[run.split] = k;
      split_ptr[run.split] = run.slots;
    }
    run = run + s;
  }
  for (int r = total.split + tid; r < n_split_cap; r += 1024) {
    split_rows[r] = -1;
    split_ptr[r] = total.slots;
  }
  if (tid == 0) {
    split_ptr[n_split_cap] = total.slots;
    totals[0] = total.items;
    totals[1] = total.split;
    totals[2] = total.slots;
  }
}

// One thread per key writes its slices; the threads past the keys pad the table's tail.
__global__ __launch_bounds__(256) void seg_fill_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ off, int n_keys, int chunk,
                                                       int4* __restrict__ items, int cap_items, const int32_t* __restrict__ totals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_keys) {
    const int k = (int)i;
    const int begin = ptr[k], deg = ptr[k + 1] - begin;
    const int n = deg > chunk ? (deg + chunk - 1) / chunk : 1;
    const int size = deg / n, rem = deg % n;
    const int first = off[2 * k], slot0 = off[2 * k + 1];
    for (int j = 0; j < n; ++j) {
      const int b = begin + j * size + min(j, rem);
      if (first + j < cap_items) items[first + j] = make_int4(k, b, b + size + (j < rem ? 1 : 0), n > 1 ? slot0 + j : -1);
    }
    return;
  }
  const int64_t p = (int64_t)totals[0] + (i - n_keys);
  if (p < cap_items) items[p] = make_int4(-1, 0, 0, -1);
}

}  // namespace

extern "C" int disgat_seg_tables(const int64_t* keys, const int64_t* perm, int64_t C, int n_keys, int chunk, int32_t* ptr,
                                 int32_t* key_off, int32_t* perm32, int32_t* items, int cap_items, int32_t* split_rows,
                                 int32_t* split_ptr, int n_split_cap, int32_t* totals, disgat_stream_t stream) {
  DISGAT_REQUIRE(keys && ptr && key_off && items && split_rows && split_ptr && totals, "seg_tables: null pointer");
  DISGAT_REQUIRE(C >= 0 && C < (int64_t(1) << 31) && n_keys >= 1 && chunk >= 1, "seg_tables: C %lld, n_keys %d, chunk %d", (long long)C,
                 n_keys, chunk);
  DISGAT_REQUIRE((perm == nullptr) == (perm32 == nullptr), "seg_tables: perm and perm32 go together");
  DISGAT_REQUIRE(cap_items >= n_keys + (int)(C / chunk) && n_split_cap >= 1 && n_split_cap >= (int)(C / chunk),
                 "seg_tables: capacities %d / %d too small for %lld entries in slices of %d", cap_items, n_split_cap, (long long)C, chunk);
  const int64_t na = (perm32 && C > n_keys + 1) ? C : (int64_t)n_keys + 1;
  hipLaunchKernelGGL(seg_ptr_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys, perm, C, n_keys, ptr, perm32);
  if (int rc = disgat::check_launch("seg_ptr")) return rc;
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ptr, n_keys, chunk, key_off, split_rows, split_ptr,
                     n_split_cap, totals);
  if (int rc = disgat::check_launch("seg_scan")) return rc;
  const int64_t nc = (int64_t)n_keys + (cap_items - n_keys);     // keys, then at most cap_items - (items written) <= cap_items - n_keys pads
  hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, key_off, n_keys, chunk,
                     reinterpret_cast<int4*>(items), cap_items, totals);
  return disgat::check_launch("seg_fill");
}
