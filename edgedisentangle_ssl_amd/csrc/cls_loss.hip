// Classification losses of the trainers in one pass over the logits.
//
// Reference: trainer.py:186-199 - output = F.log_softmax(classifier(em)); F.nll_loss(output[idx_train], labels[idx_train]);
// utils.accuracy; the same two numbers on idx_val for the log line - and pretrainer.py:819-832 (DifHead): NLLLoss of
// log_softmax(classifier(cat(in, head_i))) against the constant label i.  Through ATen that is log_softmax, two index
// gathers, gather, sum, neg, argmax, eq, sum, two casts and two divides per split (24 launches for train + val) and a dozen
// more in the backward (scatter_add into a zero-filled matrix, index_put, log_softmax_backward) - on Cora-sized graphs, whose
// train_steps are launch-bound, a sixth of a CLS step.  Here: one forward launch (two when there are more than 8 192 rows),
// one backward launch.
//
// Row r carries a code: -1 = in no split; otherwise label + (split << 16), split 0 = the one the loss is taken on (its
// gradient flows), split 1 = reported only.  Without a code table every row is in split 0 with label r % label_mod (DifHead:
// rows are (node, head) pairs, the label is the head).  Sums are taken in double, in a fixed order (thread-strided rows, wave
// shuffles, waves in order, blocks in order): the value is run-to-run deterministic.
#include "disgat_api.h"
#include "disgat_common.h"

namespace {
using namespace disgat;

constexpr int CLS_MAX_BLOCKS = 1024;
constexpr int64_t CLS_ONE_BLOCK_ROWS = 8192;

struct ClsArgs {
  const float* x;
  int64_t ld;
  const int32_t* code;
  int label_mod;
  int64_t n_rows;
  int n_cls;
  float* logp;
  int64_t ld_lp;
  double* part;      // [gridDim.x][4], or (one-block form) unused
  double div0, div1;
  float* loss;       // [1]: split 0's mean NLL as fp32 (the differentiable value)
  double* res;       // [4]: NLL / div0, correct / div0, NLL / div1, correct / div1
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sums of acc[0..4) into thread 0, waves added in order
__device__ __forceinline__ void block_sum4(double (&acc)[4], double* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) sh[w * 4 + k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double s = 0.0;
      for (int i = 0; i < nw; ++i) s += sh[i * 4 + k];
      acc[k] = s;
    }
  }
}

__device__ __forceinline__ void finish(const ClsArgs& A, const double (&s)[4]) {
  A.res[0] = s[0] / A.div0;
  A.res[1] = s[1] / A.div0;
  A.res[2] = s[2] / A.div1;
  A.res[3] = s[3] / A.div1;
  A.loss[0] = (float)(s[0] / A.div0);
}

// CV > 0: the class count is 4 * CV (4, 8, 16 - DifHead's head-index logits) and rows are 16-byte aligned: a row lives in
// CV float4 registers, read once and written once (the generic loop walks a row three times with 4-byte accesses: 0.45 ms
// for DifHead's 8M x 8 logits at C4, 3x the bytes' worth).  CV == 0: any class count.
template <bool ONE_BLOCK, int CV>
__global__ __launch_bounds__(1024) void cls_loss_fwd_kernel(const ClsArgs A) {
  __shared__ double sh[16 * 4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int C = CV > 0 ? 4 * CV : A.n_cls;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float* xr = A.x + r * A.ld;
    const int code = A.code != nullptr ? A.code[r] : (int)(r % A.label_mod);
    float mx, ls, xlab = 0.f;
    int am = 0;
    const int lab = code & 0xffff;
    if constexpr (CV > 0) {
      float v[4 * CV];
#pragma unroll
      for (int j = 0; j < CV; ++j) {
        const f32x4 t = ld4(xr + 4 * j);
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
      }
      mx = v[0];
#pragma unroll
      for (int c = 1; c < 4 * CV; ++c)
        if (v[c] > mx) { mx = v[c]; am = c; }
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4 * CV; ++c) s += expf(v[c] - mx);
      ls = logf(s);
#pragma unroll
      for (int c = 0; c < 4 * CV; ++c) xlab = (c == lab) ? v[c] : xlab;
      if (A.logp != nullptr) {
        float* lp = A.logp + r * A.ld_lp;
#pragma unroll
        for (int j = 0; j < CV; ++j)
          st4(lp + 4 * j, f32x4{(v[4 * j] - mx) - ls, (v[4 * j + 1] - mx) - ls, (v[4 * j + 2] - mx) - ls, (v[4 * j + 3] - mx) - ls});
      }
    } else {
      mx = xr[0];
      for (int c = 1; c < C; ++c) {
        const float v = xr[c];
        if (v > mx) { mx = v; am = c; }          // first maximum wins, as torch.argmax on equal values
      }
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(xr[c] - mx);
      ls = logf(s);
      if (A.logp != nullptr) {
        float* lp = A.logp + r * A.ld_lp;
        for (int c = 0; c < C; ++c) lp[c] = (xr[c] - mx) - ls;
      }
      if (code >= 0) xlab = xr[lab];
    }
    if (code >= 0) {
      const int set = (code >> 16) & 1;
      acc[2 * set] += (double)(ls - (xlab - mx));
      acc[2 * set + 1] += (am == lab) ? 1.0 : 0.0;
    }
  }
  block_sum4(acc, sh);
  if (threadIdx.x == 0) {
    if (ONE_BLOCK) {
      finish(A, acc);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) A.part[(int64_t)blockIdx.x * 4 + k] = acc[k];
    }
  }
}

__global__ __launch_bounds__(256) void cls_loss_finish_kernel(const ClsArgs A, int n_part) {
  __shared__ double sh[4 * 4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n_part; i += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += A.part[(int64_t)i * 4 + k];
  }
  block_sum4(acc, sh);
  if (threadIdx.x == 0) finish(A, acc);
}

// grad of the logits: rows of split 0 get (softmax - onehot(label)) * g / div0, every other row zeros
template <int CV>
__global__ __launch_bounds__(256) void cls_loss_bwd_kernel(const float* __restrict__ logp, int64_t ld_lp,
                                                           const int32_t* __restrict__ code, int label_mod, int64_t n_rows,
                                                           int n_cls, const float* __restrict__ g, double div0,
                                                           float* __restrict__ gx, int64_t ld_gx) {
  const int C = CV > 0 ? 4 * CV : n_cls;
  const float coef = (float)((double)g[0] / div0);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
    const int cd = code != nullptr ? code[r] : (int)(r % label_mod);
    float* o = gx + r * ld_gx;
    const bool in = cd >= 0 && ((cd >> 16) & 1) == 0;
    const int lab = cd & 0xffff;
    const float* lp = logp + r * ld_lp;
    if constexpr (CV > 0) {
#pragma unroll
      for (int j = 0; j < CV; ++j) {
        f32x4 t{0.f, 0.f, 0.f, 0.f};
        if (in) {
          const f32x4 l = ld4(lp + 4 * j);
          t.x = (expf(l.x) - (4 * j == lab ? 1.f : 0.f)) * coef;
          t.y = (expf(l.y) - (4 * j + 1 == lab ? 1.f : 0.f)) * coef;
          t.z = (expf(l.z) - (4 * j + 2 == lab ? 1.f : 0.f)) * coef;
          t.w = (expf(l.w) - (4 * j + 3 == lab ? 1.f : 0.f)) * coef;
        }
        st4(o + 4 * j, t);
      }
    } else if (in) {
      for (int c = 0; c < C; ++c) o[c] = (expf(lp[c]) - (c == lab ? 1.f : 0.f)) * coef;
    } else {
      for (int c = 0; c < C; ++c) o[c] = 0.f;
    }
  }
}

}  // namespace

extern "C" int disgat_cls_loss(const float* logits, int64_t ld, const int32_t* row_code, int label_mod, int64_t n_rows,
                               int n_cls, double div0, double div1, float* logp, int64_t ld_logp, double* block_partials,
                               float* loss, double* res, disgat_stream_t stream) {
  using namespace disgat;
  DISGAT_REQUIRE(loss && res && n_rows >= 0 && n_cls > 0 && n_cls <= 65535, "cls_loss: null output / bad sizes (n_cls=%d)", n_cls);
  DISGAT_REQUIRE(n_rows == 0 || logits, "cls_loss: null logits");
  DISGAT_REQUIRE(ld >= n_cls && (logp == nullptr || ld_logp >= n_cls), "cls_loss: row strides shorter than n_cls");
  DISGAT_REQUIRE(row_code != nullptr || (label_mod > 0 && label_mod <= n_cls), "cls_loss: without a code table label_mod must be in [1, n_cls]");
  DISGAT_REQUIRE(div0 > 0.0 && div1 > 0.0, "cls_loss: divisors must be positive");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ClsArgs A{logits, ld, row_code, label_mod, n_rows, n_cls, logp, ld_logp, block_partials, div0, div1, loss, res};
  // rows as float4 registers when the class count is 4, 8 or 16 and every row is 16-byte aligned
  const bool vec = (n_cls == 4 || n_cls == 8 || n_cls == 16) && ld % 4 == 0 && aligned16(logits) &&
                   (logp == nullptr || (ld_logp % 4 == 0 && aligned16(logp)));
  const int cv = vec ? n_cls / 4 : 0;
#define DISGAT_CLS_FWD(ONE_, grid_, block_)                                                                          \
  switch (cv) {                                                                                                      \
    case 1: hipLaunchKernelGGL((cls_loss_fwd_kernel<ONE_, 1>), dim3(grid_), dim3(block_), 0, st, A); break;          \
    case 2: hipLaunchKernelGGL((cls_loss_fwd_kernel<ONE_, 2>), dim3(grid_), dim3(block_), 0, st, A); break;          \
    case 4: hipLaunchKernelGGL((cls_loss_fwd_kernel<ONE_, 4>), dim3(grid_), dim3(block_), 0, st, A); break;          \
    default: hipLaunchKernelGGL((cls_loss_fwd_kernel<ONE_, 0>), dim3(grid_), dim3(block_), 0, st, A); break;         \
  }
  if (n_rows <= CLS_ONE_BLOCK_ROWS) {
    DISGAT_CLS_FWD(true, 1, 1024);
    return check_launch("cls_loss_fwd_kernel");
  }
  DISGAT_REQUIRE(block_partials != nullptr, "cls_loss: more than %lld rows need the block_partials scratch", (long long)CLS_ONE_BLOCK_ROWS);
  const int64_t want = (n_rows + 255) / 256;
  const int grid = (int)(want < CLS_MAX_BLOCKS ? want : CLS_MAX_BLOCKS);
  DISGAT_CLS_FWD(false, grid, 256);
#undef DISGAT_CLS_FWD
  hipLaunchKernelGGL(cls_loss_finish_kernel, dim3(1), dim3(256), 0, st, A, grid);
  return check_launch("cls_loss_finish_kernel");
}

extern "C" int disgat_cls_loss_bwd(const float* logp, int64_t ld_logp, const int32_t* row_code, int label_mod, int64_t n_rows,
                                   int n_cls, const float* g, double div0, float* grad_logits, int64_t ld_grad,
                                   disgat_stream_t stream) {
  using namespace disgat;
  if (n_rows == 0) return 0;
  DISGAT_REQUIRE(logp && g && grad_logits && n_rows > 0 && n_cls > 0 && n_cls <= 65535, "cls_loss_bwd: null pointer / bad sizes");
  DISGAT_REQUIRE(ld_logp >= n_cls && ld_grad >= n_cls, "cls_loss_bwd: row strides shorter than n_cls");
  DISGAT_REQUIRE(row_code != nullptr || (label_mod > 0 && label_mod <= n_cls), "cls_loss_bwd: without a code table label_mod must be in [1, n_cls]");
  DISGAT_REQUIRE(div0 > 0.0, "cls_loss_bwd: divisor must be positive");
  const int64_t want = (n_rows + 255) / 256;
  const int grid = (int)(want < 4096 ? want : 4096);
  const bool vec = (n_cls == 4 || n_cls == 8 || n_cls == 16) && ld_logp % 4 == 0 && ld_grad % 4 == 0 && aligned16(logp) && aligned16(grad_logits);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define DISGAT_CLS_BWD(CV_) \
  hipLaunchKernelGGL((cls_loss_bwd_kernel<CV_>), dim3(grid), dim3(256), 0, st, logp, ld_logp, row_code, label_mod, n_rows, n_cls, g, div0, grad_logits, ld_grad)
  switch (vec ? n_cls / 4 : 0) {
    case 1: DISGAT_CLS_BWD(1); break;
    case 2: DISGAT_CLS_BWD(2); break;
    case 4: DISGAT_CLS_BWD(4); break;
    default: DISGAT_CLS_BWD(0); break;
  }
#undef DISGAT_CLS_BWD
  return check_launch("cls_loss_bwd_kernel");
}
