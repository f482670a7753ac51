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

template <bool ONE_BLOCK>
__global__ __launch_bounds__(1024) void cls_loss_fwd_kernel(const ClsArgs A) {
  __shared__ double sh[16 * 4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int C = A.n_cls;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float* xr = A.x + r * A.ld;
    float mx = xr[0];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      const float v = xr[c];
      if (v > mx) { mx = v; am = c; }          // first maximum wins, as torch.argmax on equal values
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(xr[c] - mx);
    const float ls = logf(s);
    if (A.logp != nullptr) {
      float* lp = A.logp + r * A.ld_lp;
      for (int c = 0; c < C; ++c) lp[c] = (xr[c] - mx) - ls;
    }
    const int code = A.code != nullptr ? A.code[r] : (int)(r % A.label_mod);
    if (code >= 0) {
      const int set = (code >> 16) & 1, lab = code & 0xffff;
      acc[2 * set] += (double)(ls - (xr[lab] - mx));
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
__global__ __launch_bounds__(256) void cls_loss_bwd_kernel(const float* __restrict__ logp, int64_t ld_lp,
                                                           const int32_t* __restrict__ code, int label_mod, int64_t n_rows,
                                                           int C, const float* __restrict__ g, double div0,
                                                           float* __restrict__ gx, int64_t ld_gx) {
  const float coef = (float)((double)g[0] / div0);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
    const int cd = code != nullptr ? code[r] : (int)(r % label_mod);
    float* o = gx + r * ld_gx;
    if (cd >= 0 && ((cd >> 16) & 1) == 0) {
      const int lab = cd & 0xffff;
      const float* lp = logp + r * ld_lp;
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
  if (n_rows <= CLS_ONE_BLOCK_ROWS) {
    hipLaunchKernelGGL(cls_loss_fwd_kernel<true>, dim3(1), dim3(1024), 0, st, A);
    return check_launch("cls_loss_fwd_kernel");
  }
  DISGAT_REQUIRE(block_partials != nullptr, "cls_loss: more than %lld rows need the block_partials scratch", (long long)CLS_ONE_BLOCK_ROWS);
  const int64_t want = (n_rows + 255) / 256;
  const int grid = (int)(want < CLS_MAX_BLOCKS ? want : CLS_MAX_BLOCKS);
  hipLaunchKernelGGL(cls_loss_fwd_kernel<false>, dim3(grid), dim3(256), 0, st, A);
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
  hipLaunchKernelGGL(cls_loss_bwd_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logp, ld_logp, row_code,
                     label_mod, n_rows, n_cls, g, div0, grad_logits, ld_grad);
  return check_launch("cls_loss_bwd_kernel");
}
