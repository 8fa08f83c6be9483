// LDS-resident field weights of the fused Instant-NGP kernels (ngp_render.hip, raymarch_occ.hip).
#pragma once
#include "sf_common.h"
#include "ngp_device.h"

struct FieldPtrs {
  const float* table;
  const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
  float bound;
};

__device__ __forceinline__ void load_weights_lds(float* W, const FieldPtrs& f) {
  for (int i = threadIdx.x; i < NGP_HID * NGP_FEAT; i += blockDim.x) W[NGP_W0 + i] = f.w0[i];
  for (int i = threadIdx.x; i < NGP_HID * NGP_HID; i += blockDim.x) W[NGP_W1 + i] = f.w1[i];
  for (int i = threadIdx.x; i < NGP_OUT * NGP_HID; i += blockDim.x) W[NGP_W2 + i] = f.w2[i];
  for (int i = threadIdx.x; i < NGP_HID; i += blockDim.x) { W[NGP_B0 + i] = f.b0[i]; W[NGP_B1 + i] = f.b1[i]; }
  if (threadIdx.x < NGP_OUT) W[NGP_B2 + threadIdx.x] = f.b2[threadIdx.x];
}

// host: level geometry of a field (ngp_render.hip)
int sf_ngp_make_levels(const sf_ngp_field* f, NgpLevels* out, hipStream_t st);
static inline FieldPtrs sf_ngp_field_ptrs(const sf_ngp_field* f) {
  return FieldPtrs{f->embeddings, f->w0, f->b0, f->w1, f->b1, f->w2, f->b2, f->bound};
}
