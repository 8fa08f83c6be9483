// Loss glue of the distillation step (sparsefusion/distillation.py:217-241, :287-288, :306-343) as a handful of launches:
//   sf_upsample2x_forward / _backward   F.interpolate(x, scale_factor=2, mode='bilinear') of the rendered image and
//                                       silhouette (align_corners=False, source index clamped at 0) and its adjoint
//   sf_render_loss_forward / _backward  stage A (input views): |huber(rgb)| + |huber(sil)| + opacity + entropy terms
//   sf_fusion_loss_forward / _backward  stage B (novel views): (1 - alpha_bar) * |render - pred| + opacity + entropy terms
// Forward kernels leave per-workgroup partial sums of every term (summed in a fixed order by the caller: the loss value is
// reproducible and identical on every replica); backward kernels are elementwise.  torch ran each of these as 6-15 tiny
// launches plus their autograd twins.
#include "sf_common.h"

#define LOSS_THREADS 256
#define LOSS_MAX_BLOCKS 256

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block sums of K running values -> partial[blockIdx.x][K] (fixed order inside the block: shuffle tree, then wave 0..3)
template <int K>
__device__ __forceinline__ void block_partials(float (&acc)[K], float* __restrict__ partial) {
  __shared__ float red[4][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float s = wave_sum_f(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) partial[blockIdx.x * K + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// torch's upsample_bilinear2d source coordinate for scale 2, align_corners=False: src = max(0, (dst + 0.5) / 2 - 0.5)
__device__ __forceinline__ void up2_src(int dst, int n_in, int& i0, int& i1, float& l1) {
  float s = ((float)dst + 0.5f) * 0.5f - 0.5f;
  if (s < 0.0f) s = 0.0f;
  i0 = (int)s;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ __launch_bounds__(LOSS_THREADS) void k_upsample2x(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w) {
  const long total = (long)planes * 4 * h * w;
  const int W2 = 2 * w, H2 = 2 * h;
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LOSS_THREADS) {
    const int x = (int)(i % W2), y = (int)((i / W2) % H2);
    const long pl = i / ((long)W2 * H2);
    int x0, x1, y0, y1;
    float lx, ly;
    up2_src(x, w, x0, x1, lx);
    up2_src(y, h, y0, y1, ly);
    const float* p = in + pl * h * w;
    const float hx = 1.0f - lx, hy = 1.0f - ly;
    out[i] = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
  }
}

// adjoint: every low-resolution pixel gathers the <= 4 x 4 high-resolution gradients whose stencil contains it
__global__ __launch_bounds__(LOSS_THREADS) void k_upsample2x_bwd(const float* __restrict__ gout, float* __restrict__ gin, int planes, int h, int w) {
  const long total = (long)planes * h * w;
  const int W2 = 2 * w, H2 = 2 * h;
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LOSS_THREADS) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const long pl = i / ((long)w * h);
    const float* g = gout + pl * H2 * W2;
    float acc = 0.0f;
    for (int yy = 2 * y - 2; yy <= 2 * y + 2; ++yy) {
      if (yy < 0 || yy >= H2) continue;
      int y0, y1;
      float ly;
      up2_src(yy, h, y0, y1, ly);
      const float wy = (y0 == y ? 1.0f - ly : 0.0f) + (y1 == y ? ly : 0.0f);
      if (wy == 0.0f) continue;
      for (int xx = 2 * x - 2; xx <= 2 * x + 2; ++xx) {
        if (xx < 0 || xx >= W2) continue;
        int x0, x1;
        float lx;
        up2_src(xx, w, x0, x1, lx);
        const float wx = (x0 == x ? 1.0f - lx : 0.0f) + (x1 == x ? lx : 0.0f);
        if (wx != 0.0f) acc = fmaf(wy * wx, g[yy * W2 + xx], acc);
      }
    }
    gin[i] = acc;
  }
}

// ---- shared term definitions
__device__ __forceinline__ float huber_abs(float x, float y, float scaling) {        // |huber(x, y)| of utils/common_utils.py:183-190
  const float d = x - y;
  float t = 1.0f + d * d / (scaling * scaling);
  t = fmaxf(t, 1e-4f);
  return fabsf((sqrtf(t) - 1.0f) * scaling);
}
__device__ __forceinline__ float huber_abs_grad(float x, float y, float scaling) {   // d |huber| / dx (sqrt(t) >= 1: the abs is inactive)
  const float d = x - y;
  const float t = 1.0f + d * d / (scaling * scaling);
  if (t < 1e-4f) return 0.0f;
  return d / (scaling * sqrtf(t));
}
__device__ __forceinline__ float opacity_term(float s) { return sqrtf(s * s + 0.01f); }
__device__ __forceinline__ float opacity_grad(float s) { return s / sqrtf(s * s + 0.01f); }
__device__ __forceinline__ float entropy_term(float s) {
  const float a = fminf(fmaxf(s, 1e-5f), 1.0f - 1e-5f);
  return -a * log2f(a) - (1.0f - a) * log2f(1.0f - a);
}
__device__ __forceinline__ float entropy_grad(float s) {                              // clamp passes no gradient outside (1e-5, 1 - 1e-5)
  if (s < 1e-5f || s > 1.0f - 1e-5f) return 0.0f;
  return log2f(1.0f - s) - log2f(s);
}

// ---- stage A: partial[b][4] = sums of |huber(rgb)|, |huber(sil)|, sqrt(sil^2 + .01), entropy(sil)
__global__ __launch_bounds__(LOSS_THREADS) void k_render_loss_fwd(const float* __restrict__ img, const float* __restrict__ sil,
                                                                   const float* __restrict__ trgb, const float* __restrict__ tmask,
                                                                   long n_img, long n_sil, float scaling, float* __restrict__ partial) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_img; i += (long)gridDim.x * LOSS_THREADS)
    acc[0] += huber_abs(img[i], trgb[i], scaling);
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_sil; i += (long)gridDim.x * LOSS_THREADS) {
    const float s = sil[i];
    if (tmask) acc[1] += huber_abs(s, tmask[i], scaling);
    acc[2] += opacity_term(s);
    acc[3] += entropy_term(s);
  }
  block_partials<4>(acc, partial);
}

// coef = upstream gradient * (lambda / element count) of each term
__global__ __launch_bounds__(LOSS_THREADS) void k_render_loss_bwd(const float* __restrict__ img, const float* __restrict__ sil,
                                                                   const float* __restrict__ trgb, const float* __restrict__ tmask,
                                                                   long n_img, long n_sil, float scaling, float c_rgb, float c_sil, float c_op,
                                                                   float c_ent, const float* __restrict__ gup, float* __restrict__ g_img,
                                                                   float* __restrict__ g_sil) {
  const float gs = gup ? gup[0] : 1.0f;          // upstream gradient of the scalar loss, read on the device (no host sync)
  c_rgb *= gs; c_sil *= gs; c_op *= gs; c_ent *= gs;
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_img; i += (long)gridDim.x * LOSS_THREADS)
    g_img[i] = c_rgb * huber_abs_grad(img[i], trgb[i], scaling);
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_sil; i += (long)gridDim.x * LOSS_THREADS) {
    const float s = sil[i];
    float g = c_op * opacity_grad(s) + c_ent * entropy_grad(s);
    if (tmask) g += c_sil * huber_abs_grad(s, tmask[i], scaling);
    g_sil[i] = g;
  }
}

// ---- stage B: partial[b][3] = sums of weight[view] * |render - pred|, sqrt(sil^2 + .01), entropy(sil)
__global__ __launch_bounds__(LOSS_THREADS) void k_fusion_loss_fwd(const float* __restrict__ img, const float* __restrict__ pred,
                                                                   const float* __restrict__ weight, const float* __restrict__ sil,
                                                                   long per_view_img, long n_img, long n_sil, float* __restrict__ partial) {
  float acc[3] = {0.f, 0.f, 0.f};
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_img; i += (long)gridDim.x * LOSS_THREADS)
    acc[0] += weight[i / per_view_img] * fabsf(img[i] - pred[i]);
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_sil; i += (long)gridDim.x * LOSS_THREADS) {
    const float s = sil[i];
    acc[1] += opacity_term(s);
    acc[2] += entropy_term(s);
  }
  block_partials<3>(acc, partial);
}

__global__ __launch_bounds__(LOSS_THREADS) void k_fusion_loss_bwd(const float* __restrict__ img, const float* __restrict__ pred,
                                                                   const float* __restrict__ weight, const float* __restrict__ sil,
                                                                   long per_view_img, long n_img, long n_sil, float c_l1, float c_op, float c_ent,
                                                                   const float* __restrict__ gup, float* __restrict__ g_img,
                                                                   float* __restrict__ g_sil) {
  const float gs = gup ? gup[0] : 1.0f;
  c_l1 *= gs; c_op *= gs; c_ent *= gs;
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_img; i += (long)gridDim.x * LOSS_THREADS) {
    const float d = img[i] - pred[i];
    g_img[i] = c_l1 * weight[i / per_view_img] * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
  }
  for (long i = blockIdx.x * (long)LOSS_THREADS + threadIdx.x; i < n_sil; i += (long)gridDim.x * LOSS_THREADS) {
    const float s = sil[i];
    g_sil[i] = c_op * opacity_grad(s) + c_ent * entropy_grad(s);
  }
}

static uint32_t loss_grid(long n) {
  const long b = (n + LOSS_THREADS - 1) / LOSS_THREADS;
  return (uint32_t)(b < 1 ? 1 : (b > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : b));
}

extern "C" uint32_t sf_loss_partial_rows(void) { return LOSS_MAX_BLOCKS; }

extern "C" int sf_upsample2x_forward(const float* in, float* out, uint32_t planes, uint32_t h, uint32_t w, void* stream) {
  if (!in || !out || !planes || !h || !w) SF_FAIL(SF_ERR_INVALID, "upsample2x: bad arguments");
  k_upsample2x<<<sf_grid_cap(sf_div_up((uint64_t)planes * 4 * h * w, LOSS_THREADS)), LOSS_THREADS, 0, (hipStream_t)stream>>>(in, out, (int)planes, (int)h, (int)w);
  SF_CHECK_LAUNCH("upsample2x");
  return SF_OK;
}

extern "C" int sf_upsample2x_backward(const float* grad_out, float* grad_in, uint32_t planes, uint32_t h, uint32_t w, void* stream) {
  if (!grad_out || !grad_in || !planes || !h || !w) SF_FAIL(SF_ERR_INVALID, "upsample2x_backward: bad arguments");
  k_upsample2x_bwd<<<sf_grid_cap(sf_div_up((uint64_t)planes * h * w, LOSS_THREADS)), LOSS_THREADS, 0, (hipStream_t)stream>>>(grad_out, grad_in, (int)planes, (int)h, (int)w);
  SF_CHECK_LAUNCH("upsample2x_backward");
  return SF_OK;
}

// partial: [sf_loss_partial_rows()][4] floats; rows beyond the launched grid are zeroed
extern "C" int sf_render_loss_forward(const float* img, const float* sil, const float* target_rgb, const float* target_mask, uint64_t n_img,
                                      uint64_t n_sil, float scaling, float* partial, void* stream) {
  if (!img || !sil || !target_rgb || !partial || !n_img || !n_sil) SF_FAIL(SF_ERR_INVALID, "render_loss: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(partial, 0, LOSS_MAX_BLOCKS * 4 * sizeof(float), st) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "render_loss: memset failed");
  k_render_loss_fwd<<<loss_grid((long)(n_img > n_sil ? n_img : n_sil)), LOSS_THREADS, 0, st>>>(img, sil, target_rgb, target_mask, (long)n_img, (long)n_sil, scaling, partial);
  SF_CHECK_LAUNCH("render_loss_forward");
  return SF_OK;
}

extern "C" int sf_render_loss_backward(const float* img, const float* sil, const float* target_rgb, const float* target_mask, uint64_t n_img,
                                       uint64_t n_sil, float scaling, float c_rgb, float c_sil, float c_opacity, float c_entropy,
                                       const float* grad_loss, float* grad_img, float* grad_sil, void* stream) {
  if (!img || !sil || !target_rgb || !grad_img || !grad_sil) SF_FAIL(SF_ERR_INVALID, "render_loss_backward: bad arguments");
  k_render_loss_bwd<<<sf_grid_cap(sf_div_up(n_img > n_sil ? n_img : n_sil, LOSS_THREADS)), LOSS_THREADS, 0, (hipStream_t)stream>>>(
      img, sil, target_rgb, target_mask, (long)n_img, (long)n_sil, scaling, c_rgb, c_sil, c_opacity, c_entropy, grad_loss, grad_img, grad_sil);
  SF_CHECK_LAUNCH("render_loss_backward");
  return SF_OK;
}

// partial: [sf_loss_partial_rows()][3] floats
extern "C" int sf_fusion_loss_forward(const float* img, const float* pred, const float* view_weight, const float* sil, uint32_t views,
                                      uint64_t per_view_img, uint64_t per_view_sil, float* partial, void* stream) {
  if (!img || !pred || !view_weight || !sil || !partial || !views || !per_view_img || !per_view_sil) SF_FAIL(SF_ERR_INVALID, "fusion_loss: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(partial, 0, LOSS_MAX_BLOCKS * 3 * sizeof(float), st) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "fusion_loss: memset failed");
  const long n_img = (long)views * per_view_img, n_sil = (long)views * per_view_sil;
  k_fusion_loss_fwd<<<loss_grid(n_img), LOSS_THREADS, 0, st>>>(img, pred, view_weight, sil, (long)per_view_img, n_img, n_sil, partial);
  SF_CHECK_LAUNCH("fusion_loss_forward");
  return SF_OK;
}

extern "C" int sf_fusion_loss_backward(const float* img, const float* pred, const float* view_weight, const float* sil, uint32_t views,
                                       uint64_t per_view_img, uint64_t per_view_sil, float c_l1, float c_opacity, float c_entropy,
                                       const float* grad_loss, float* grad_img, float* grad_sil, void* stream) {
  if (!img || !pred || !view_weight || !sil || !grad_img || !grad_sil || !views) SF_FAIL(SF_ERR_INVALID, "fusion_loss_backward: bad arguments");
  const long n_img = (long)views * per_view_img, n_sil = (long)views * per_view_sil;
  k_fusion_loss_bwd<<<sf_grid_cap(sf_div_up((uint64_t)n_img, LOSS_THREADS)), LOSS_THREADS, 0, (hipStream_t)stream>>>(
      img, pred, view_weight, sil, (long)per_view_img, n_img, n_sil, c_l1, c_opacity, c_entropy, grad_loss, grad_img, grad_sil);
  SF_CHECK_LAUNCH("fusion_loss_backward");
  return SF_OK;
}
