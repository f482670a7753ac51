// Multi-tensor Adam: one launch updates every parameter tensor of a trainer step (encoder heads, fusers,
// classifiers: ~60 small tensors), each with the hyper-parameters and step count of the per-module optimiser it
// belongs to.  Replaces the 3-5 torch.optim.Adam(...).step() calls of /root/reference/trainer.py:58-60, 205-206 and
// pretrainer.py:754-756, 633-635, 838-840 (one Adam per sub-module, all stepped every train_step): same arithmetic
// as torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected moments, eps outside the sqrt),
// state kept per tensor.  HBM-bound elementwise work: 4 reads + 3 writes of 4 B per element, float4 accesses.
#include "disgat_api.h"

namespace {

constexpr int kMaxTensors = DISGAT_ADAM_MAX_TENSORS;
constexpr int kChunk = 2048;        // elements per block: 256 threads x 2 float4

struct AdamArgs {                   // passed by value in the kernel arguments (no table upload, graph-capturable)
  float* p[kMaxTensors];
  const float* g[kMaxTensors];
  float* m[kMaxTensors];
  float* v[kMaxTensors];
  int n[kMaxTensors];
  int first_block[kMaxTensors + 1];
  union alignas(8) {
    struct {
      float step_size[kMaxTensors];     // lr / (1 - beta1^t)
      float inv_sqrt_bc2[kMaxTensors];  // 1 / sqrt(1 - beta2^t)
    };
    double lr[kMaxTensors];             // DEV: the plain learning rate (the corrections are formed in the kernel)
  };
  float wd[kMaxTensors];
  int count;
};

// DEV: the step counts live on the device (steps[t], already advanced): what a train_step captured in a HIP graph needs -
// the host-side corrections of the eager form would be baked into the graph.  Same double arithmetic as the host's.
template <bool DEV>
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamArgs a, float omb1, float beta2, float omb2, float eps,
                                                         const int* __restrict__ steps, double beta1_d, double beta2_d) {
  // block -> tensor: binary search over <= 64 prefix entries (uniform per block: scalar registers)
  int lo = 0, hi = a.count;
  const int b = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.first_block[mid] <= b) lo = mid; else hi = mid;
  }
  const int t = lo;
  const int n = a.n[t];
  const int base = (b - a.first_block[t]) * kChunk;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  float ss, ib;
  if constexpr (DEV) {
    const double tt = (double)steps[t];
    ss = (float)(a.lr[t] / (1.0 - pow(beta1_d, tt)));
    ib = (float)(1.0 / sqrt(1.0 - pow(beta2_d, tt)));
  } else {
    ss = a.step_size[t];
    ib = a.inv_sqrt_bc2[t];
  }
  const float wd = a.wd[t];
  auto upd = [&](float& pw, float gw, float& mw, float& vw) {
    gw = fmaf(wd, pw, gw);                       // grad = grad + wd * param
    mw = mw + (gw - mw) * omb1;                  // exp_avg.lerp_(grad, 1 - beta1)
    vw = fmaf(vw, beta2, omb2 * gw * gw);        // exp_avg_sq = beta2 * v + (1 - beta2) g^2
    const float denom = sqrtf(vw) * ib + eps;    // sqrt(v) / sqrt(bc2) + eps
    pw = pw - ss * (mw / denom);
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = base + (r * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      float4 pw = *reinterpret_cast<float4*>(p + i);
      const float4 gw = *reinterpret_cast<const float4*>(g + i);
      float4 mw = *reinterpret_cast<float4*>(m + i);
      float4 vw = *reinterpret_cast<float4*>(v + i);
      upd(pw.x, gw.x, mw.x, vw.x);
      upd(pw.y, gw.y, mw.y, vw.y);
      upd(pw.z, gw.z, mw.z, vw.z);
      upd(pw.w, gw.w, mw.w, vw.w);
      *reinterpret_cast<float4*>(p + i) = pw;
      *reinterpret_cast<float4*>(m + i) = mw;
      *reinterpret_cast<float4*>(v + i) = vw;
    } else {
      for (int k = i; k < n && k < i + 4; ++k) {
        float pw = p[k], mw = m[k], vw = v[k];
        upd(pw, g[k], mw, vw);
        p[k] = pw;
        m[k] = mw;
        v[k] = vw;
      }
    }
  }
}

}  // namespace

static int adam_launch(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, const float* step_size, const float* inv_sqrt_bc2,
                       const double* lr, const float* weight_decay, const int32_t* steps, double beta1, double beta2,
                       float eps, disgat_stream_t stream) {
  DISGAT_REQUIRE(count >= 0, "adam_multi: negative tensor count");
  const double beta1_d = beta1, beta2_d = beta2;
  for (int off = 0; off < count; off += kMaxTensors) {
    AdamArgs a;
    a.count = count - off < kMaxTensors ? count - off : kMaxTensors;
    int blocks = 0;
    for (int i = 0; i < a.count; ++i) {
      const int64_t n = numel[off + i];
      DISGAT_REQUIRE(n >= 0 && n < (int64_t(1) << 31) - kChunk, "adam_multi: tensor %d has %lld elements", off + i, (long long)n);
      DISGAT_REQUIRE(params[off + i] && grads[off + i] && exp_avg[off + i] && exp_avg_sq[off + i],
                     "adam_multi: null pointer for tensor %d", off + i);
      a.p[i] = params[off + i];
      a.g[i] = grads[off + i];
      a.m[i] = exp_avg[off + i];
      a.v[i] = exp_avg_sq[off + i];
      a.n[i] = (int)n;
      a.first_block[i] = blocks;
      if (steps) {
        a.lr[i] = lr[off + i];
      } else {
        a.step_size[i] = step_size[off + i];
        a.inv_sqrt_bc2[i] = inv_sqrt_bc2[off + i];
      }
      a.wd[i] = weight_decay[off + i];
      blocks += (int)((n + kChunk - 1) / kChunk);
    }
    a.first_block[a.count] = blocks;
    if (blocks == 0) continue;
    // 1 - beta in double, as torch forms the lerp / addcmul weights (1.0f - 0.999f is off by 1.3e-5 relative)
    if (steps)
      hipLaunchKernelGGL(adam_multi_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, (float)(1.0 - beta1_d),
                         (float)beta2_d, (float)(1.0 - beta2_d), eps, steps + off, beta1_d, beta2_d);
    else
      hipLaunchKernelGGL(adam_multi_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, (float)(1.0 - beta1_d),
                         (float)beta2_d, (float)(1.0 - beta2_d), eps, (const int*)nullptr, beta1_d, beta2_d);
    if (int rc = disgat::check_launch("adam_multi")) return rc;
  }
  return 0;
}

extern "C" int disgat_adam_multi(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const int64_t* numel, const float* step_size,
                                 const float* inv_sqrt_bc2, const float* weight_decay, double beta1, double beta2,
                                 float eps, disgat_stream_t stream) {
  DISGAT_REQUIRE(count == 0 || (step_size && inv_sqrt_bc2 && weight_decay && numel), "adam_multi: null table");
  return adam_launch(count, params, grads, exp_avg, exp_avg_sq, numel, step_size, inv_sqrt_bc2, nullptr, weight_decay, nullptr,
                     beta1, beta2, eps, stream);
}

extern "C" int disgat_adam_multi_dev(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                     float* const* exp_avg_sq, const int64_t* numel, const double* lr,
                                     const float* weight_decay, const int32_t* steps, double beta1, double beta2, float eps,
                                     disgat_stream_t stream) {
  DISGAT_REQUIRE(count == 0 || (lr && weight_decay && numel && steps), "adam_multi_dev: null table");
  return adam_launch(count, params, grads, exp_avg, exp_avg_sq, numel, nullptr, nullptr, lr, weight_decay, steps, beta1, beta2,
                     eps, stream);
}
