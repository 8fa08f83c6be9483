// oracle/ref_exports.cpp -- TEST INFRASTRUCTURE ONLY.  C-ABI doors into the REFERENCE's own host entry points
// (external/gridencoder/src/gridencoder.h:12-13, raymarching/src/raymarching.h:7-17), compiled on the host from the
// reference's .cu files through oracle/cuda_shim (recipe: oracle/build_ref.py -> oracle/_ref/libref_native.so).
// Same argument order as the oracle_* functions of ngp_ref.c, so tests can call both with one argument list.
#include <torch/torch.h>

thread_local shim_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

// the reference's own declarations, read from the headers where they lie (-I <reference>/raymarching/src and
// -I <reference>/external/gridencoder/src, oracle/build_ref.py): raymarching.h:7-17, gridencoder.h:12-13
#include "raymarching.h"
#include "gridencoder.h"

namespace {
at::Tensor F(const void* p) { return {const_cast<void*>(p), at::ScalarType::Float}; }
at::Tensor I(const void* p) { return {const_cast<void*>(p), at::ScalarType::Int}; }
at::Tensor U8(const void* p) { return {const_cast<void*>(p), at::ScalarType::Byte}; }
at::optional<at::Tensor> OF(const void* p) { return p ? at::optional<at::Tensor>(F(p)) : at::optional<at::Tensor>(); }
char g_err[512];
template <class Fn> int guarded(Fn&& f) {
  try { f(); g_err[0] = 0; return 0; } catch (const std::exception& e) { snprintf(g_err, sizeof g_err, "%s", e.what()); return 1; }
}
}  // namespace

extern "C" {
const char* ref_last_error() { return g_err; }

int ref_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype,
                            int align_corners) {
  return guarded([&] { grid_encode_forward(F(inputs), F(embeddings), I(offsets), F(outputs), B, D, C, L, S, H, OF(dy_dx), gridtype, align_corners != 0); });
}
int ref_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                             float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners) {
  return guarded([&] { grid_encode_backward(F(grad), F(inputs), F(embeddings), I(offsets), F(grad_embeddings), B, D, C, L, S, H, OF(dy_dx), OF(grad_inputs), gridtype, align_corners != 0); });
}
int ref_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears, float* fars) {
  return guarded([&] { near_far_from_aabb(F(rays_o), F(rays_d), F(aabb), N, min_near, F(nears), F(fars)); });
}
int ref_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
  return guarded([&] { sph_from_ray(F(rays_o), F(rays_d), radius, N, F(coords)); });
}
int ref_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) { return guarded([&] { morton3D(I(coords), N, I(indices)); }); }
int ref_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) { return guarded([&] { morton3D_invert(I(indices), N, I(coords)); }); }
int ref_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
  return guarded([&] { packbits(F(grid), N, density_thresh, U8(bitfield)); });
}
int ref_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                         float* deltas, int32_t* rays, int32_t* counter, const float* noises) {
  return guarded([&] { march_rays_train(F(rays_o), F(rays_d), U8(grid), bound, dt_gamma, max_steps, N, C, H, M, F(nears), F(fars), F(xyzs), F(dirs), F(deltas), I(rays), I(counter), F(noises)); });
}
int ref_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                                     float T_thresh, float* weights_sum, float* depth, float* image) {
  return guarded([&] { composite_rays_train_forward(F(sigmas), F(rgbs), F(deltas), I(rays), M, N, T_thresh, F(weights_sum), F(depth), F(image)); });
}
int ref_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                      const float* deltas, const int32_t* rays, const float* weights_sum, const float* image, uint32_t M,
                                      uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs) {
  return guarded([&] { composite_rays_train_backward(F(grad_weights_sum), F(grad_image), F(sigmas), F(rgbs), F(deltas), I(rays), F(weights_sum), F(image), M, N, T_thresh, F(grad_sigmas), F(grad_rgbs)); });
}
int ref_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o, const float* rays_d,
                   float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                   const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises) {
  return guarded([&] { march_rays(n_alive, n_step, I(rays_alive), F(rays_t), F(rays_o), F(rays_d), bound, dt_gamma, max_steps, C, H, U8(grid), F(nears), F(fars), F(xyzs), F(dirs), F(deltas), F(noises)); });
}
int ref_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t, const float* sigmas,
                       const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image) {
  return guarded([&] { composite_rays(n_alive, n_step, T_thresh, I(rays_alive), F(rays_t), F(sigmas), F(rgbs), F(deltas), F(weights_sum), F(depth), F(image)); });
}
}
