// Multi-tensor Adam step for the NGP parameters (table 929 336 x 2 + six MLP tensors): ONE launch per
// optimizer.step() instead of torch's per-op foreach kernels (two steps per distillation iteration,
// sparsefusion/distillation.py:165,246,352: torch.optim.Adam(ngp_network.get_params(lr=5e-4))).
// Arithmetic follows torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False):
//   m <- lerp(m, g, 1-b1) ; v <- b2 v + (1-b2) g^2 ; p <- p - (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Pure streaming: 16 B/param read + 12 B/param written, HBM bound (1.87 M params = 52 MB per step).

#include "sf_common.h"

__global__ __launch_bounds__(256) void k_adam_multi(sf_adam_args a) {
  // block -> tensor by the prefix table of 1024-element chunks
  int t = 0;
  while (t + 1 < (int)a.n_tensors && blockIdx.x >= a.chunk_start[t + 1]) ++t;
  const sf_adam_tensor& d = a.t[t];
  const long base = (long)(blockIdx.x - a.chunk_start[t]) * 1024;
  const float w1 = a.one_minus_beta1, w2 = a.one_minus_beta2;     // 1 - beta in double on the host, as torch passes them
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i >= (long)d.n) break;
    const float g = d.grad[i];
    float m = d.exp_avg[i], v = d.exp_avg_sq[i];
    m = fmaf(w1, g - m, m);
    v = fmaf(w2 * g, g, a.beta2 * v);
    d.exp_avg[i] = m;
    d.exp_avg_sq[i] = v;
    const float denom = __fdiv_rn(__fsqrt_rn(v), a.bias_correction2_sqrt) + a.eps;
    d.param[i] = fmaf(-d.step_size, __fdiv_rn(m, denom), d.param[i]);
  }
}

extern "C" int sf_adam_multi(const sf_adam_args* args, void* stream) {
  if (!args) SF_FAIL(SF_ERR_INVALID, "adam: null arguments");
  sf_adam_args a = *args;
  if (a.n_tensors == 0) return SF_OK;
  if (a.n_tensors > SF_ADAM_MAX_TENSORS) SF_FAIL(SF_ERR_INVALID, "adam: at most %d tensors per launch", SF_ADAM_MAX_TENSORS);
  uint32_t chunks = 0;
  for (uint32_t t = 0; t < a.n_tensors; ++t) {
    if (!a.t[t].param || !a.t[t].grad || !a.t[t].exp_avg || !a.t[t].exp_avg_sq) SF_FAIL(SF_ERR_INVALID, "adam: null tensor %u", t);
    a.chunk_start[t] = chunks;
    chunks += sf_div_up(a.t[t].n, 1024);
  }
  if (chunks == 0) return SF_OK;
  k_adam_multi<<<chunks, 256, 0, (hipStream_t)stream>>>(a);
  SF_CHECK_LAUNCH("adam_multi");
  return SF_OK;
}
