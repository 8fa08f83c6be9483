/*
 * sparsefusion_hip.h -- C ABI of libsparsefusion_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the score-distillation hot path of zhizdev/sparsefusion
 * (SURVEY.md section 8(b)).  Plain pointers and sizes only: no torch types.
 * Every pointer is a DEVICE pointer unless its name starts with `h_`.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream); all work
 * is enqueued on it and nothing synchronises.  Callee never allocates: every
 * output is caller-allocated and mutated in place (reference ownership rule,
 * external/gridencoder/grid.py:42-47, raymarching/raymarching.py:42-45).
 *
 * Return value: 0 on success, non-zero SF_ERR_* on failure; sf_last_error()
 * returns a thread-local message (the Python wrappers raise RuntimeError with
 * it, mirroring TORCH_CHECK / std::runtime_error in the reference bindings).
 */
#ifndef SPARSEFUSION_HIP_H
#define SPARSEFUSION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_ERR_INVALID 1   /* bad argument (unsupported C / D / shape)        */
#define SF_ERR_LAUNCH 2    /* hipLaunch / hipGetLastError reported an error   */

const char* sf_last_error(void);
int sf_abi_version(void);
/* 1 when this build multiplies IEEE-half operands (libsparsefusion_hip_f16.so, -DSF_OPERAND_F16=1), 0 for bf16 (default): host-side
 * weight tables handed to the library must be rounded to that type. */
int sf_operand_is_f16(void);

/* ------------------------------------------------------------------------ */
/* _gridencoder  (external/gridencoder/src/bindings.cpp:6-7)                 */
/* ------------------------------------------------------------------------ */

/* Replaces grid_encode_forward (gridencoder.cu:424-447, kernel_grid :75-223).
 * inputs [B,D] f32 in [0,1]; embeddings [rows,C] f32; offsets [L+1] i32;
 * outputs [L,B,C] f32 (level-major, as the reference); dy_dx [B,L*D*C] or NULL.
 * S = log2(per_level_scale), H = base resolution, gridtype 0=hash 1=tiled.
 * h_offsets: host copy of offsets (L+1 ints) -- needed to launch without a
 * device->host sync; pass NULL to let the library read it back (syncs). */
int sf_grid_encode_forward(const float* inputs, const float* embeddings,
                           const int32_t* offsets, float* outputs,
                           uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                           float S, uint32_t H, float* dy_dx,
                           uint32_t gridtype, int align_corners,
                           const int32_t* h_offsets, void* stream);

/* Replaces grid_encode_backward (gridencoder.cu:449-479, kernel_grid_backward
 * :226-313, kernel_input_backward :316-342).  grad [L,B,C]; grad_embeddings
 * [rows,C] must be zero-initialised by the caller (grid.py:72); dy_dx /
 * grad_inputs may be NULL together. */
int sf_grid_encode_backward(const float* grad, const float* inputs,
                            const float* embeddings, const int32_t* offsets,
                            float* grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, const float* dy_dx,
                            float* grad_inputs, uint32_t gridtype,
                            int align_corners, const int32_t* h_offsets,
                            void* stream);

/* ------------------------------------------------------------------------ */
/* _raymarching  (raymarching/src/bindings.cpp:7-18)                         */
/* ------------------------------------------------------------------------ */

/* Replaces near_far_from_aabb (raymarching.cu:91-156). rays_o/d [N,3],
 * aabb [6], nears/fars [N]; all f32. */
int sf_near_far_from_aabb(const float* rays_o, const float* rays_d,
                          const float* aabb, uint32_t N, float min_near,
                          float* nears, float* fars, void* stream);

/* Replaces morton3D / morton3D_invert (raymarching.cu:214-254). coords [N,3]
 * i32, indices [N] i32. Bit-exact integer work. */
int sf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);
int sf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);

/* Replaces packbits (raymarching.cu:267-289). grid [N*8] f32 -> bitfield [N] u8,
 * bit i of byte n set iff grid[n*8+i] > density_thresh. */
int sf_packbits(const float* grid, uint32_t N, float density_thresh,
                uint8_t* bitfield, void* stream);

/* ------------------------------------------------------------------------ */
/* `_raymarching` occupancy-grid entry points (cuda_ray=True path):          */
/* raymarching/src/bindings.cpp:8,12-18, raymarching.h:8,12-18.              */
/* All tensors f32 / i32 / u8 device pointers, caller-allocated, mutated in  */
/* place exactly where the reference mutates them.                           */
/* ------------------------------------------------------------------------ */

/* Replaces sph_from_ray (raymarching.cu:159-204). coords [N,2] in [-1,1]. */
int sf_sph_from_ray(const float* rays_o, const float* rays_d, float radius,
                    uint32_t N, float* coords, void* stream);

/* Replaces march_rays_train (raymarching.cu:302-492).
 * grid: density bitfield [C*H^3/8]; xyzs/dirs [M,3], deltas [M,2] (caller
 * zero-fills), rays [N,3] i32 = (ray id, point offset, point count),
 * counter [2] i32 += (points, rays) as the reference's two atomicAdd do.
 * Point slots are assigned in ray order by a scan (one of the reference's
 * valid outcomes, deterministic). `workspace` holds the per-ray counts:
 * sf_march_rays_train_workspace_bytes(N) bytes, caller-allocated. */
uint64_t sf_march_rays_train_workspace_bytes(uint32_t N);
int sf_march_rays_train(const float* rays_o, const float* rays_d,
                        const uint8_t* grid, float bound, float dt_gamma,
                        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                        uint32_t M, const float* nears, const float* fars,
                        float* xyzs, float* dirs, float* deltas, int32_t* rays,
                        int32_t* counter, const float* noises,
                        void* workspace, uint64_t workspace_bytes, void* stream);

/* Replaces composite_rays_train_forward / _backward (raymarching.cu:495-688). */
int sf_composite_rays_train_forward(const float* sigmas, const float* rgbs,
                                    const float* deltas, const int32_t* rays,
                                    uint32_t M, uint32_t N, float T_thresh,
                                    float* weights_sum, float* depth,
                                    float* image, void* stream);
int sf_composite_rays_train_backward(const float* grad_weights_sum,
                                     const float* grad_image,
                                     const float* sigmas, const float* rgbs,
                                     const float* deltas, const int32_t* rays,
                                     const float* weights_sum,
                                     const float* image, uint32_t M, uint32_t N,
                                     float T_thresh, float* grad_sigmas,
                                     float* grad_rgbs, void* stream);

/* Replaces march_rays / composite_rays (inference, raymarching.cu:695-913).
 * rays_alive [>= n_alive] i32 (set to -1 when a ray terminates), rays_t [N],
 * weights_sum/depth/image accumulated in place. */
int sf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                  const float* rays_t, const float* rays_o, const float* rays_d,
                  float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                  uint32_t H, const uint8_t* grid, const float* nears,
                  const float* fars, float* xyzs, float* dirs, float* deltas,
                  const float* noises, void* stream);
int sf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                      int32_t* rays_alive, float* rays_t, const float* sigmas,
                      const float* rgbs, const float* deltas,
                      float* weights_sum, float* depth, float* image,
                      void* stream);

/* ------------------------------------------------------------------------ */
/* Fused NGP render (external/nerf/renderer_df.py:310-468 `run`,             */
/* network_grid.py:69-88 `common_forward`) -- see DESIGN.md section 3.       */
/* ------------------------------------------------------------------------ */

/* Field parameters: tiled grid table + 32-64-64-4 MLP (network_grid.py:50-52). */
typedef struct {
  const float* embeddings;   /* [rows, 2]                                   */
  const int32_t* h_offsets;  /* host, [L+1]                                 */
  uint32_t L;                /* 16                                          */
  float S;                   /* log2(per_level_scale)                       */
  uint32_t H;                /* base resolution                             */
  uint32_t gridtype;         /* 0 hash, 1 tiled                             */
  const float* w0; const float* b0;   /* [64,32],[64]  sigma_net.net.0     */
  const float* w1; const float* b1;   /* [64,64],[64]  sigma_net.net.1     */
  const float* w2; const float* b2;   /* [4,64],[4]    sigma_net.net.2     */
  float bound;               /* scene bound (inputs mapped (x+b)/(2b))      */
} sf_ngp_field;

/* Gradient outputs of the field (all caller-zeroed, accumulated with atomics) */
typedef struct {
  float* g_embeddings;
  float* g_w0; float* g_b0;
  float* g_w1; float* g_b1;
  float* g_w2; float* g_b2;
} sf_ngp_field_grad;

/* sigma/albedo at arbitrary points (NeRFNetwork.density, network_grid.py:200).
 * xyz [P,3] in [-bound,bound] -> sigma [P], albedo [P,3]. */
int sf_ngp_density(const sf_ngp_field* f, const float* xyz, uint32_t P,
                   float* sigma, float* albedo, void* stream);

/* Full training/eval render of N rays with T coarse + T fine samples (T<=64).
 * lin [T] = linspace(0,1,T); u_coarse [N,T] in [0,1) or NULL (perturb=False);
 * u_fine: uniforms for the inverse-CDF draw, row n at u_fine + n*u_fine_row_stride
 * ([N,T] with stride T for training, or one [T] row linspace(.5/T,1-.5/T,T) with
 * stride 0 for det=True).  Saved for backward: z_sorted [N,2T], sigma_s [N,2T],
 * rgb_s [N,2T,3], nears/fars [N].  Outputs image [N,3], depth [N], weights_sum [N].
 * bg_color: scalar background.  workspace: sf_ngp_render_forward_workspace_bytes(N,T) (the backward's is larger and also fits). */
int sf_ngp_render_forward(const sf_ngp_field* f, const float* rays_o,
                          const float* rays_d, const float* aabb, uint32_t N,
                          uint32_t T, float min_near, const float* lin,
                          const float* u_coarse, const float* u_fine,
                          uint32_t u_fine_row_stride, float bg_color, float* nears,
                          float* fars, float* z_sorted, float* sigma_s,
                          float* rgb_s, float* image, float* depth,
                          float* weights_sum, float* field_cache, float* workspace,
                          uint64_t workspace_bytes, void* stream);

/* Backward of sf_ngp_render_forward w.r.t. the field parameters given
 * grad_image [N,3] and grad_weights_sum [N] or NULL (depth carries no gradient
 * in the reference losses).  Gradients are ACCUMULATED into caller-zeroed buffers.
 * rays_per_row: image width if the N rays are a row-major H x W image (lets the table
 * scatter work on 8x8 patches), 0 if unknown. */
int sf_ngp_render_backward(const sf_ngp_field* f, const sf_ngp_field_grad* g,
                           const float* rays_o, const float* rays_d,
                           const float* aabb, uint32_t N, uint32_t T,
                           const float* nears, const float* fars,
                           const float* z_sorted, const float* sigma_s,
                           const float* rgb_s, float bg_color,
                           const float* grad_image, const float* grad_weights_sum,
                           uint32_t rays_per_row, const float* field_cache,
                           float* workspace, uint64_t workspace_bytes,
                           void* stream);

/* workspace of the backward (also enough for the forward): composite / field gradients per sample (72 N T floats) plus, since r04,
 * the bins of the table-gradient scatter (csrc/ngp_scatter_bin.h): 16-byte entries, 96 per sample of ONE ray chunk -- chunks hold at
 * most 8192 rays for any N (r05: unequal chunks where N has no equal split), i.e. <= 1.6 GB of bins at T = 64 whatever N is
 * (128 x 128 rays x 64 + 64 samples: 1.6 GB of bins + 0.3 GB of per-sample gradients). */
uint64_t sf_ngp_render_workspace_bytes(uint32_t N, uint32_t T);
/* workspace of the forward alone (10 * N * T floats): what an evaluation render needs */
uint64_t sf_ngp_render_forward_workspace_bytes(uint32_t N, uint32_t T);

/* field_cache (both calls; NULL = none): sf_ngp_render_cache_bytes(N, T) bytes the forward fills with the hash-grid features of
 * every sample ([N*T][32] coarse, [N*T][32] fine) and the sort permutation ([N][2T] u32); the backward then reads a sample's
 * features back instead of re-gathering 16 levels x 8 corners (the reference's autograd keeps the encoder output alive the
 * same way: external/gridencoder/grid.py:42-47 saves inputs / embeddings / dy_dx for its backward).  The caller keeps the
 * buffer untouched between the two calls. */
uint64_t sf_ngp_render_cache_bytes(uint32_t N, uint32_t T);

/* Fused evaluation render through the occupancy grid (`cuda_ray=True`, eval mode): replaces the host loop
 * `while step < max_steps: march_rays -> network -> composite_rays` of external/nerf/renderer_df.py:543-584 by ONE
 * launch (one ray per lane walks to the end).  `grid` = the density bitfield, C cascades of H^3 cells;
 * `noises` [N] = the jitter of the first sample (the reference perturbs the first round only) or NULL.
 * Outputs weights_sum [N], depth [N] (un-normalised, as composite_rays leaves it), image [N,3] without background. */
int sf_ngp_render_occ_eval(const sf_ngp_field* f, const float* rays_o, const float* rays_d,
                           const float* nears, const float* fars, const uint8_t* grid,
                           float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                           const float* noises, float T_thresh, uint32_t N,
                           float* weights_sum, float* depth, float* image, void* stream);

/* ------------------------------------------------------------------------ */
/* UNet op executor (external/imagen_pytorch.py:1470-1671) -- the host side  */
/* builds a static launch plan once (sparsefusion_amd/unet.py) and the       */
/* library replays it on a stream; see DESIGN.md section 4.                  */
/* ------------------------------------------------------------------------ */

enum {
  SF_OP_CONV = 1,      /* implicit-GEMM conv / linear on MFMA bf16          */
  SF_OP_GN_ACT = 2,    /* GroupNorm(+scale/shift)+SiLU -> bf16              */
  SF_OP_LN = 3,        /* LayerNorm over channels (+GELU) -> bf16 / f32     */
  SF_OP_GEMV = 4,      /* small-M linear (time path, gca net, context kv)   */
  SF_OP_ATTN = 5,      /* 16-token attention core                           */
  SF_OP_GCA_POOL = 6,  /* GlobalContext softmax pooling                     */
  SF_OP_ELTWISE = 7,   /* gate*h + residual, adds, pixel-shuffle, packing   */
  SF_OP_MEMSET = 8,    /* zero a region of the activation arena             */
  SF_OP_TIME_EMB = 9,  /* learned sinusoidal embedding of log-snr           */
  SF_OP_SPLITK_REDUCE = 10, /* stand-alone reduction of deferred split-K partials */
  SF_OP_POOL = 11,     /* 2x2 max pooling fwd / bwd (LPIPS-VGG)             */
  SF_OP_LPIPS = 12,    /* LPIPS per-layer head fwd / bwd                    */
  SF_OP_EFT = 13,      /* EFT pre-pass: resize, grid-sample gather, harmonic embedding, short-sequence attention, softmax pooling */
  SF_OP_FCONV = 14,    /* [GroupNorm | LayerNorm] (+scale/shift, SiLU) fused into the conv's A-operand prologue (Block, :641-662) */
  SF_OP_SLOTS = 15,    /* (sum, sum of squares) slots of a tensor for the next fused GroupNorm; optional gate*h + residual first */
  SF_OP_GCA = 16,      /* fused GlobalContext stages (imagen_pytorch.py:916-941) */
  SF_OP_INITX = 17,    /* latent half of the init CrossEmbed conv inside a sampler trajectory: x0 = base + conv_{3,7,15}(x) (:1017-1042) */
  SF_OP_GN_FINALIZE = 18   /* GroupNorm statistics from the per-tile partial sums a conv epilogue left (experimental VAE path) */
};

/* One op = one or two kernel launches.  Interpretation of p[]/i[]/f[] per op type is
 * documented next to each kernel in sparsefusion_amd/csrc/unet_ops.hip. */
typedef struct {
  int32_t type;
  int32_t flags;
  void* p[24];
  int32_t i[32];
  float f[8];
} sf_op;

/* ------------------------------------------------------------------------ */
/* Multi-tensor Adam: replaces the two torch.optim.Adam.step() calls of every */
/* distillation iteration (sparsefusion/distillation.py:165,246,352) by one    */
/* launch each. torch.optim.Adam arithmetic, amsgrad/weight_decay off.         */
/* ------------------------------------------------------------------------ */
#define SF_ADAM_MAX_TENSORS 16
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  uint64_t n;
  float step_size;          /* lr / (1 - beta1^step), per tensor (per param group) */
  float pad_;
} sf_adam_tensor;
typedef struct {
  sf_adam_tensor t[SF_ADAM_MAX_TENSORS];
  uint32_t chunk_start[SF_ADAM_MAX_TENSORS];   /* filled by the library */
  uint32_t n_tensors;
  float beta1, beta2, eps;
  float bias_correction2_sqrt;                 /* sqrt(1 - beta2^step) */
  float one_minus_beta1, one_minus_beta2;      /* computed in double by the caller */
} sf_adam_args;
int sf_adam_multi(const sf_adam_args* args, void* stream);

/* ---- loss glue of the distillation step (csrc/loss_ops.hip).  Replaces the torch elementwise chains of
 * sparsefusion/distillation.py:217-241 (stage A: |huber| colour + silhouette, opacity, entropy), :287-288
 * (F.interpolate(scale_factor=2, mode='bilinear') of the render) and :310-343 (stage B: (1 - alpha_bar) * L1 to the decoded
 * prediction, opacity, entropy).  Forward entry points leave per-workgroup partial sums, `sf_loss_partial_rows()` rows of
 * 4 (stage A: |huber rgb|, |huber sil|, opacity, entropy) or 3 (stage B: weighted L1, opacity, entropy) floats; the caller
 * sums the rows and applies lambda / element count.  Backward entry points take coefficient = lambda / element count per term and `grad_loss`, a DEVICE
 * scalar holding the upstream gradient of the loss (null = 1).  Images are NCHW fp32; `target_mask` may be null (scene without masks). */
uint32_t sf_loss_partial_rows(void);
int sf_upsample2x_forward(const float* in, float* out, uint32_t planes, uint32_t h, uint32_t w, void* stream);
int sf_upsample2x_backward(const float* grad_out, float* grad_in, uint32_t planes, uint32_t h, uint32_t w, void* stream);
int sf_render_loss_forward(const float* img, const float* sil, const float* target_rgb, const float* target_mask, uint64_t n_img,
                           uint64_t n_sil, float scaling, float* partial, void* stream);
int sf_render_loss_backward(const float* img, const float* sil, const float* target_rgb, const float* target_mask, uint64_t n_img,
                            uint64_t n_sil, float scaling, float c_rgb, float c_sil, float c_opacity, float c_entropy,
                            const float* grad_loss, float* grad_img, float* grad_sil, void* stream);
int sf_fusion_loss_forward(const float* img, const float* pred, const float* view_weight, const float* sil, uint32_t views,
                           uint64_t per_view_img, uint64_t per_view_sil, float* partial, void* stream);
int sf_fusion_loss_backward(const float* img, const float* pred, const float* view_weight, const float* sil, uint32_t views,
                            uint64_t per_view_img, uint64_t per_view_sil, float c_l1, float c_opacity, float c_entropy,
                            const float* grad_loss, float* grad_img, float* grad_sil, void* stream);

int sf_plan_run(const sf_op* ops, uint32_t n_ops, void* stream);
/* sf_plan_run with a HIP event before every op on the launch stream; h_ms[n_ops] (host) gets per-op
 * elapsed milliseconds.  Synchronises; measurement aid for bench.py (per-kernel roofline). */
int sf_plan_profile(const sf_op* ops, uint32_t n_ops, void* stream, float* h_ms);

/* Weight packing helpers (host pointers in, device-ready blobs out). */
/* Pack a conv weight [Cout, Cin, kh, kw] f32 (host) into MFMA-fragment order
 * bf16 (host buffer `out`, size sf_conv_packed_elems()*2 bytes). Cin is padded
 * to cin_pad (multiple of 32), Cout to a multiple of 16. */
uint64_t sf_conv_packed_elems(uint32_t Cout, uint32_t cin_pad, uint32_t kh, uint32_t kw);
int sf_conv_pack_weights(const float* h_w, uint32_t Cout, uint32_t Cin,
                         uint32_t cin_pad, uint32_t kh, uint32_t kw,
                         uint16_t* h_out);

/* Fused PLMS latent update (external/plms.py:158-214 get_model_output):
 * x0 = clamp((x - sigma*e)/max(alpha,1e-8), +-clip); mean = a_next*(x*(1-c)/alpha + c*x0);
 * x_prev = mean + noise_scale*noise.  coef = {alpha, sigma, alpha_next, c, noise_scale, clip}. */
int sf_plms_update(const float* x, const float* eps, const float* noise,
                   const float* h_coef6, uint64_t n, float* x_prev, float* x0,
                   void* stream);
/* e' = c0*e0 + c1*e1 + c2*e2 + c3*e3 (Adams-Bashforth combination, plms.py:137-152);
 * keep_e0 (or NULL) also receives a copy of e0 -- the history entry of this step. */
int sf_plms_combine(const float* e0, const float* e1, const float* e2,
                    const float* e3, const float* h_c4, uint64_t n, float* out,
                    float* keep_e0, void* stream);
/* sf_plms_combine followed by sf_plms_update on its result, in one launch (the steady-state step of
 * external/plms.py:137-152 + :122-135): x_prev = update(x, c0*e0 + .. + c3*e3, noise).
 * row_n > 0: extra workgroups of the same launch copy row_n floats (a multiple of 4, 16-byte aligned) from row_src to
 * row_dst -- the UNet's time-block row of the NEXT eval into the plan's arena (Unet.eval_prepared(.., row_ready=True)). */
int sf_plms_step(const float* e0, const float* e1, const float* e2, const float* e3,
                 const float* h_c4, float* keep_e0, const float* x, const float* noise,
                 const float* h_coef6, uint64_t n, float* x_prev,
                 const float* row_src, float* row_dst, uint64_t row_n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEFUSION_HIP_H */
