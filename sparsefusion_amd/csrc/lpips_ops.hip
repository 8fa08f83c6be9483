// Extra plan ops for the perceptual-loss term of the distillation step:
// LPIPS(net='vgg') as called by external/external_utils.py:11-50 / sparsefusion/distillation.py:312-314.
// The VGG16 convs run on k_conv_igemm (unet_ops.hip; ReLU in the epilogue); this file holds what is left:
//   SF_OP_POOL   2x2 max pooling, forward and backward
//   SF_OP_LPIPS  per-layer head: channel-unit-normalise both feature maps, squared difference, non-negative
//                1x1 "lin" weights, spatial mean -- forward (scalar per sample) and backward (d / d features of
//                the first image; the second image is the no-grad target)
//   eltwise helpers: ReLU backward, the input scaling layer and its gradient
// All of it is elementwise / per-pixel reductions in fp32: HBM bound, one wave per pixel for the head.

#include "sf_common.h"
#include "plan_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef sf_opnd bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lp_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- max pooling -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool_fwd(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                  int C) {
  const int Ho = H / 2, Wo = W / 2, c4 = C / 4;
  const long total = (long)B * Ho * Wo * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    long r = i / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const float* p = in + (((long)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), bq = *reinterpret_cast<const f32x4*>(p + C);
    const f32x4 cq = *reinterpret_cast<const f32x4*>(p + (long)W * C), d = *reinterpret_cast<const f32x4*>(p + (long)W * C + C);
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a[j], bq[j]), fmaxf(cq[j], d[j]));
    *reinterpret_cast<f32x4*>(out + (((long)b * Ho + oy) * Wo + ox) * C + c) = m;
  }
}

// din gets dout at the FIRST window position (row-major) that holds the maximum, 0 elsewhere (torch's argmax rule)
__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ dout, const float* __restrict__ in,
                                                  float* __restrict__ din, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, c4 = C / 4;
  const long total = (long)B * Ho * Wo * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    long r = i / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const long base = (((long)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const long offs[4] = {0, C, (long)W * C, (long)W * C + C};
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(in + base + offs[k]);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + (((long)b * Ho + oy) * Wo + ox) * C + c);
    f32x4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int arg = 0;
      float m = v[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][j] > m) { m = v[k][j]; arg = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][j] = (k == arg) ? g[j] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(din + base + offs[k]) = o[k];
  }
}

// ---- LPIPS head ---------------------------------------------------------------------------------------------
// feats [2V, HW, C]: samples 0..V-1 = in0 (gradient flows), V..2V-1 = in1.  One wave per pixel.
//   u = f0 / (|f0| + eps), v = f1 / (|f1| + eps), d = sum_c w_c (u_c - v_c)^2, out[s] += mean_pixels d
#define LPIPS_EPS 1e-10f
__global__ __launch_bounds__(256) void k_lpips_head_fwd(const float* __restrict__ feats, const float* __restrict__ w,
                                                        float* __restrict__ out, int V, int HW, int C) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per_sample_blocks = gridDim.x / V;
  const int s = blockIdx.x / per_sample_blocks, bl = blockIdx.x % per_sample_blocks;
  float acc = 0.0f;
  for (int p = bl * 4 + wv; p < HW; p += per_sample_blocks * 4) {
    const float* f0 = feats + ((long)s * HW + p) * C;
    const float* f1 = feats + ((long)(V + s) * HW + p) * C;
    float n0 = 0.0f, n1 = 0.0f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(f0 + c), b = *reinterpret_cast<const f32x4*>(f1 + c);
      n0 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
      n1 += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    }
    n0 = lp_wave_sum(n0); n1 = lp_wave_sum(n1);
    const float r0 = 1.0f / (sqrtf(n0) + LPIPS_EPS), r1 = 1.0f / (sqrtf(n1) + LPIPS_EPS);
    float d = 0.0f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(f0 + c), b = *reinterpret_cast<const f32x4*>(f1 + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = a[j] * r0 - b[j] * r1;
        d = fmaf(ww[j] * e, e, d);
      }
    }
    acc += lp_wave_sum(d);
  }
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(out + s),
                                                  (red[0] + red[1] + red[2] + red[3]) / (float)HW);
}

// dfeat[s, p, :] = gscale[s] / HW * d d_p / d f0:  a_c = 2 w_c (u_c - v_c);  df = a / n - f0 * (a . f0) / (n^2 |f0|)
__global__ __launch_bounds__(256) void k_lpips_head_bwd(const float* __restrict__ feats, const float* __restrict__ w,
                                                        const float* __restrict__ gscale, float* __restrict__ dfeat, int V,
                                                        int HW, int C) {
  const int lane = threadIdx.x & 63;
  const long pixels = (long)V * HW;
  for (long q = blockIdx.x * 4L + (threadIdx.x >> 6); q < pixels; q += (long)gridDim.x * 4) {
    const int s = (int)(q / HW);
    const float* f0 = feats + q * C;
    const float* f1 = feats + ((long)V * HW + q) * C;
    float n0 = 0.0f, n1 = 0.0f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(f0 + c), b = *reinterpret_cast<const f32x4*>(f1 + c);
      n0 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
      n1 += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    }
    n0 = lp_wave_sum(n0); n1 = lp_wave_sum(n1);
    const float l0 = sqrtf(n0);
    const float r0 = 1.0f / (l0 + LPIPS_EPS), r1 = 1.0f / (sqrtf(n1) + LPIPS_EPS);
    float dot = 0.0f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(f0 + c), b = *reinterpret_cast<const f32x4*>(f1 + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) dot = fmaf(2.0f * ww[j] * (a[j] * r0 - b[j] * r1), a[j], dot);
    }
    dot = lp_wave_sum(dot);
    const float g = gscale[s] / (float)HW;
    const float k2 = l0 > 0.0f ? dot * r0 * r0 / l0 : 0.0f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(f0 + c), b = *reinterpret_cast<const f32x4*>(f1 + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = g * (2.0f * ww[j] * (a[j] * r0 - b[j] * r1) * r0 - a[j] * k2);
      *reinterpret_cast<f32x4*>(dfeat + q * C + c) = o;
    }
  }
}

// ---- elementwise helpers ------------------------------------------------------------------------------------
// out bf16 = dy * (y > 0)      (ReLU backward; y is the post-ReLU activation)
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                  sf_opnd* __restrict__ out, long n4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4), a = *reinterpret_cast<const f32x4*>(y + i * 4);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (sf_opnd)(a[j] > 0.0f ? g[j] : 0.0f);
    *reinterpret_cast<bf16x4*>(out + i * 4) = o;
  }
}

// ScalingLayer: out NHWC [B, HW, Cp] (zero padded) = (x NCHW [B, 3, HW] - shift[c]) / scale[c] ; k = (shift3, scale3)
__global__ __launch_bounds__(256) void k_scale_in(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ out,
                                                  int B, int HW, int Cp) {
  const long n = (long)B * HW * Cp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const long bp = i / Cp;
    const int p = (int)(bp % HW), b = (int)(bp / HW);
    out[i] = c < 3 ? (x[((long)b * 3 + c) * HW + p] - k[c]) / k[3 + c] : 0.0f;
  }
}
// its gradient: dx NCHW [B, 3, HW] = g NHWC [B, HW, ld][.., c] / scale[c]
__global__ __launch_bounds__(256) void k_scale_in_bwd(const float* __restrict__ g, const float* __restrict__ k,
                                                      float* __restrict__ dx, int B, int HW, int ld) {
  const long n = (long)B * 3 * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long bc = i / HW;
    const int c = (int)(bc % 3), b = (int)(bc / 3);
    dx[i] = g[((long)b * HW + p) * ld + c] / k[3 + c];
  }
}

int sf_plan_extra_op(const sf_op* opp, void* stream) {
  const sf_op& op = *opp;
  hipStream_t st = (hipStream_t)stream;
  switch (op.type) {
    case SF_OP_POOL: {
      const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
      if (op.flags == 2) return sf_plan_eft_op(opp, stream);          // 3x3 / 2 stem pooling (eft_ops.hip)
      if ((H | W) & 1 || C % 4) SF_FAIL(SF_ERR_INVALID, "pool: H, W must be even and C a multiple of 4");
      const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
      if (op.flags == 0) {
        if (!op.p[0] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "pool: null tensor");
        k_pool_fwd<<<sf_grid_cap(sf_div_up(total, 256)), 256, 0, st>>>((const float*)op.p[0], (float*)op.p[3], B, H, W, C);
      } else {
        if (!op.p[0] || !op.p[1] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "pool backward: null tensor");
        k_pool_bwd<<<sf_grid_cap(sf_div_up(total, 256)), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1],
                                                                      (float*)op.p[3], B, H, W, C);
      }
      SF_CHECK_LAUNCH("pool");
      return SF_OK;
    }
    case SF_OP_LPIPS: {
      const int V = op.i[0], HW = op.i[1], C = op.i[2];
      if (V < 1 || C % 4 || !op.p[0] || !op.p[1] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "lpips head: bad operands");
      if (op.flags == 0) {
        int per = sf_div_up((uint64_t)HW, 4);
        if (per > 256) per = 256;                            // one atomic per workgroup: keep them few
        k_lpips_head_fwd<<<V * per, 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (float*)op.p[3], V, HW, C);
      } else {
        if (!op.p[2]) SF_FAIL(SF_ERR_INVALID, "lpips head backward: missing upstream gradient");
        k_lpips_head_bwd<<<sf_grid_cap(sf_div_up((uint64_t)V * HW, 4)), 256, 0, st>>>(
            (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[3], V, HW, C);
      }
      SF_CHECK_LAUNCH("lpips_head");
      return SF_OK;
    }
    case SF_OP_ELTWISE:
      switch (op.flags) {
        case 7: {
          const long n4 = (long)(uint32_t)op.i[0] / 4;
          k_relu_bwd<<<sf_grid_cap(sf_div_up(n4, 256)), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (sf_opnd*)op.p[3], n4);
          break;
        }
        case 8:
          k_scale_in<<<sf_grid_cap(sf_div_up((long)op.i[0] * op.i[1] * op.i[2], 256)), 256, 0, st>>>(
              (const float*)op.p[0], (const float*)op.p[1], (float*)op.p[3], op.i[0], op.i[1], op.i[2]);
          break;
        case 9:
          k_scale_in_bwd<<<sf_grid_cap(sf_div_up((long)op.i[0] * 3 * op.i[1], 256)), 256, 0, st>>>(
              (const float*)op.p[0], (const float*)op.p[1], (float*)op.p[3], op.i[0], op.i[1], op.i[2]);
          break;
        default: SF_FAIL(SF_ERR_INVALID, "eltwise: unknown mode %d", op.flags);
      }
      SF_CHECK_LAUNCH("eltwise");
      return SF_OK;
    default: SF_FAIL(SF_ERR_INVALID, "plan: unknown op type %d", op.type);
  }
}
