#!/usr/bin/env python
"""Headline benchmark: hot-path score-distillation steps on MI355X (BASELINE.json metric / config).

One "step" = one pass of the north-star hot path for one novel view per GPU, synthetic 2-view scene:
  A  input-view NGP render (128x128 rays, 64+64 samples) fwd + bwd + Adam           distillation.py:185-247
  B  novel-view NGP render fwd -> x2 bilinear -> SD-VAE encode(.).mode() * z_scale -> PLMSSampler.sample(
     max_thres drawn per step in [0.5, 0.99) (r05; `--max-thres 0.5` = the fixed schedule of the r01-r04 lines): 50 steps = 51 UNet evals
     at 32x32 latents, 256-ch view features) -> SD-VAE decode ->
     (1-alpha_bar)*L1 + 1e-3*opacity + 1e-3*entropy -> bwd + Adam                    distillation.py:262-352
     + 0.1 * LPIPS-VGG(render, decoded)  (lambda_percep of itr > 1000, distillation.py:176-178,312-314)
Everything runs on this repo's HIP path: NGP render, UNet/PLMS, the SD-VAE and the LPIPS
term (row 2; VGG16 / lin weights are synthetic, the `lpips` package is not available).  fp32 everywhere except the
conv / linear MFMA operands of the UNet, VAE and VGG (bf16, fp32 accumulate).

`--config 2` (BASELINE configs[2]): 6 input views, and the 256-channel view features of every novel view are NOT cached: each
step first renders them through the Epipolar Feature Transformer (`renderer_feat(cameras=, volumetric_function=
eft.batched_forward, n_batches=16, input_cameras=, input_rgb=)`, distillation.py:99-109: 32x32 rays x 20 depths x 6 views).
`--config 3` = `--views-per-gpu 4` (BASELINE configs[3], 4 novel views per GPU and step); the DEFAULT line also carries a short
measurement of that regime (`also_measured.config3_B4`: 1 + 3 steps at 4 views on this GPU, after the headline's timed region).
`--config 4` (BASELINE configs[4] on the GPUs given): IEEE-half MFMA operands for the UNet, the full 50-step PLMS trajectory
(max_thres 0.999: 51 evals), and the 512 x 512 evaluation render through `render_batched` timed in `breakdown_ms`.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
N > 1 runs one rank per GPU over RCCL -- under torch.distributed.run (RANK / WORLD_SIZE in the environment), or spawned by
bench.py itself when it is started plainly (it re-execs under torch.distributed.run): every rank distils its own novel
view (weak scaling: the headline); the ranks all-gather the rendered latents and all-reduce (mean) the NGP gradients
before each optimiser step so the replicas stay identical; that they are is checked once, after the timed region.
Every line (N = 1 included) also carries `also_measured.config3_total32`: BASELINE configs[3] as the strong-scaling experiment it
names -- 32 novel views per step in total, block-sharded over the ranks (4 per GPU at N = 8) -- so that value(N) / value(1) of
that object is the ">= 3.5x at 8 GPUs" figure."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def pinhole_rays(n_side, view, n_views, device, radius=6.0, focal=2.0, elevation=0.3):
    """Camera on a circle around the origin looking at it; pytorch3d-style non-unit directions."""
    ang = 2 * np.pi * view / n_views
    eye = torch.tensor([radius * np.cos(ang), radius * elevation, radius * np.sin(ang)], dtype=torch.float32)
    fwd = -eye / eye.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
    right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    ax = torch.linspace(1 - 1 / n_side, -1 + 1 / n_side, n_side)
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    d = fwd[None, None] + (xx[..., None] * right[None, None] + yy[..., None] * up[None, None]) / focal
    d = d.reshape(1, -1, 3).contiguous()
    return eye.view(1, 1, 3).expand_as(d).contiguous().to(device), d.to(device)


def entropy(sil):
    """opacity entropy regulariser of the reference loss (sparsefusion/distillation.py:237-241, :337-341)."""
    a = sil.clamp(1e-5, 1 - 1e-5)
    return (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).mean()


def huber(x, y, scaling=0.1):
    diff_sq = (x - y) ** 2
    return ((1 + diff_sq / (scaling ** 2)).clamp(1e-4).sqrt() - 1) * float(scaling)


class HotPath:
    def __init__(self, device, rank, world, max_thres, views=1, seed=0, n_input_views=2, eft_features=False, unet_operand=None,
                 thres_range=None):
        from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
        from sparsefusion_amd.unet import Unet
        from sparsefusion_amd.vldm import DDPM
        from sparsefusion_amd.plms import PLMSSampler
        self.dev, self.rank, self.world, self.max_thres, self.views = device, rank, world, max_thres, views
        # r05: the reference draws max_thres anew on EVERY step (distillation.py:303 torch.rand(1).clamp(0, .99)), so the
        # sampler's schedule -- and the UNet time table built from it -- changes every step.  `thres_range` = (lo, hi): this step's
        # max_thres is drawn uniformly from it on a seeded HOST generator (identical on every rank); None = the fixed `max_thres`.
        self.thres_range, self.thres_gen = thres_range, torch.Generator().manual_seed(seed + 4242)
        self.evals_run = 0                                        # UNet evals of the steps so far (counted on the host from the schedule)
        torch.manual_seed(seed)                                   # identical replicas on every rank
        self.opt = get_default_torch_ngp_opt()
        ngp = NeRFNetwork(self.opt)
        ngp.encoder.embeddings.data.uniform_(-0.5, 0.5)           # "trained-like" table, semi-transparent scene
        ngp.sigma_net.net[2].bias.data[0] = -3.0
        self.ngp = ngp.to(device).train()
        from sparsefusion_amd.optim import FusedAdam
        self.optim = FusedAdam(self.ngp.get_params(lr=5e-4))              # torch.optim.Adam arithmetic, one launch per step
        from sparsefusion_amd.distributed import FlatGradBucket
        self.grads = FlatGradBucket(self.ngp.parameters())       # .grad = views of one flat buffer: zero-copy all-reduce
        self.multi = world > 1                                   # collectives are issued (bench main sets it for the single-rank RCCL run as well)
        self.check_replicas = world > 1                          # cheap (two 4-element collectives per step): on by default
        self.time_collectives = False
        unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2),
                    layer_attns=(False, False, False, True), layer_cross_attns=(False, False, False, False),
                    cond_images_channels=256, attn_pool_text=False)
        with torch.no_grad():                                     # the reference zero-inits final_conv: use a trained-like one
            unet.final_conv.weight.normal_(0, 0.02)
        if unet_operand:                                          # BASELINE configs[4] "fp16 UNet": IEEE-half MFMA operands, fp32 accumulate
            unet.set_operand(unet_operand)
        self.vldm = DDPM(channels=4, unets=(unet,), conditional_encoder=None, conditional_embed_dim=None, image_sizes=(32,),
                         timesteps=500, cond_drop_prob=0.1, pred_objectives='noise', conditional=False,
                         auto_normalize_img=False, clip_output=True, dynamic_thresholding=False,
                         dynamic_thresholding_percentile=.68, clip_value=10).to(device)
        self.unet = self.vldm.unets[0]
        # hipGraph replay of the plan body also under RCCL: the capture is thread-local and contains no collective; if the
        # capture fails in a process with RCCL threads, Unet falls back to plain replay (same GPU time: the GPU is the bottleneck)
        self.unet.use_hip_graph = os.environ.get("SF_BENCH_GRAPH", "1") != "0"
        self.plms = PLMSSampler(self.vldm, 50)
        from sparsefusion_amd.vae import AutoencoderKL
        self.vae = AutoencoderKL().to(device)                     # sd-vae.yaml architecture, default (kaiming-range) init
        self.z_scale = 0.18215                                    # args.z_scale_factor of the reference's demo
        from sparsefusion_amd.lpips import PerceptualLoss
        self.percep = PerceptualLoss('vgg', device=device)        # distillation.py:161
        self.percep.model.accept_synthetic_weights()              # synthetic VGG16 / lin weights, stated in `data` (no `lpips` package here)
        self.lambda_percep = 0.1                                  # value after start_percep_step (:176-178)
        g = torch.Generator().manual_seed(100 + rank)
        self.rays_in = pinhole_rays(128, rank % 2, 34, device)                 # one of the 2 input views
        self.target_rgb = torch.rand(1, 3, 128, 128, generator=g).to(device)
        self.target_mask = (torch.rand(1, 1, 128, 128, generator=g) > 0.5).float().to(device)
        self.set_views(views, g)
        self.flat_grads = None
        self.n_input_views, self.eft = n_input_views, None
        self.coll_us = {"all_gather_latents": [], "all_reduce_grads": []}     # host-timed collectives (multi-rank runs)
        if eft_features:
            # BASELINE configs[2]: the view features come from the EFT pre-pass over n_input_views input images every step
            from sparsefusion_amd.eft import EpipolarFeatureTransformer
            from sparsefusion_amd.utils.cameras import PinholeCameras
            from sparsefusion_amd.utils.render_utils import init_light_field_renderer
            self.eft = EpipolarFeatureTransformer(use_r=True, encoder='resnet18', return_features=True, remove_unused_layers=False).to(device)
            self.in_rgb = torch.rand(n_input_views, 3, 256, 256, generator=g).to(device)
            self.in_cams = self._circle_cameras([0.45 * i - 0.3 for i in range(n_input_views)], PinholeCameras).to(device)
            self.novel_cams = [self._circle_cameras([0.25 + 0.2 * (rank * views + v)], PinholeCameras).to(device) for v in range(views)]
            _, _, self.renderer_feat = init_light_field_renderer(device, 256, 256, min=1.0, max=8.0, scale_factor=8.0)   # distillation.py:86
            self.eft.encode(self.in_cams, self.in_rgb)           # once per scene (distillation.py:92-98)

    def set_views(self, views, g=None):
        """this rank's novel views of a step (cached-feature configurations): rays and the 256-channel conditioning of each"""
        g = g or torch.Generator().manual_seed(200 + self.rank)
        self.views = views
        self.rays_novel = [pinhole_rays(128, 2 + self.rank * views + v, 34, self.dev) for v in range(views)]
        self.features = torch.randn(views, 256, 32, 32, generator=g).to(self.dev)  # cached EFT features of the novel views

    @staticmethod
    def _circle_cameras(angles, cls):
        import math
        Rs = [torch.tensor([[math.cos(a), 0, -math.sin(a)], [0, 1, 0], [math.sin(a), 0, math.cos(a)]], dtype=torch.float32) for a in angles]
        Ts = [torch.tensor([0.02 * i, -0.01 * i, 4.0]) for i in range(len(angles))]
        return cls(torch.stack(Rs), torch.stack(Ts), torch.full((len(angles), 2), 2.2))

    @torch.no_grad()
    def render_features(self):
        """distillation.py:99-117 for this rank's novel views: EFT feature render -> [V, 256, 32, 32] UNet conditioning."""
        out = []
        for cam in self.novel_cams:
            feats, _, _ = self.renderer_feat(cameras=cam, volumetric_function=self.eft.batched_forward, n_batches=16,
                                             input_cameras=self.in_cams, input_rgb=self.in_rgb)
            out.append(feats[..., 3:].permute(0, 3, 1, 2))
        return torch.cat(out, 0).contiguous()

    def sampler_ctx(self):
        """a prepared trajectory context (time table + conditioning part of the init conv) for timing single evals"""
        if getattr(self, "_sctx", None) is None or self._sctx["generation"] != self._sctx["plan"].generation:
            self._sctx = self.unet.begin_sampling(self.features[:1], torch.linspace(-3, 3, 4, device=self.dev))
            self.sampler_x = torch.zeros(1, 4, 32, 32, device=self.dev)
        return self._sctx

    def render(self, rays):
        o, d = rays
        out = self.ngp.render(o, d, staged=False, perturb=True, bg_color=0, ambient_ratio=1.0, shading='albedo',
                              force_all_rays=True, **vars(self.opt))
        img = out['image'].reshape(1, 128, 128, 3).permute(0, 3, 1, 2).contiguous()
        sil = out['weights_sum'].reshape(1, 128, 128, 1).permute(0, 3, 1, 2).contiguous()
        return img, sil

    def sync_grads(self):
        """mean all-reduce of the NGP gradients over the replicas: ONE in-place RCCL call on the flat gradient buffer.  It
        cannot overlap the novel-view render: that render reads the parameters the optimiser step behind this reduce writes
        (distillation.py:247 -> :282), so the collective is issued asynchronously only to keep the host ahead, and waited
        on the stream right before the step."""
        if self.multi and self.time_collectives:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        work = self.grads.all_reduce(async_op=True)
        if work is not None:
            work.wait()                                          # stream-side wait for RCCL (host returns at once)
        if self.multi and self.time_collectives:
            torch.cuda.synchronize()
            self.coll_us["all_reduce_grads"].append((time.perf_counter() - t0) * 1e6)

    def verify_replicas(self):
        """all ranks hold bit-identical NGP parameters?  Two small collectives and a host sync: called ONCE, after the timed region
        (r04 ran it inside every timed step)."""
        if self.check_replicas and self.multi:
            from sparsefusion_amd.distributed import replicas_identical
            self.replicas_ok = replicas_identical(self.ngp)
            assert self.replicas_ok, "NGP replicas diverged"

    def draw_max_thres(self):
        if self.thres_range is None:
            return self.max_thres
        lo, hi = self.thres_range
        return lo + (hi - lo) * float(torch.rand(1, generator=self.thres_gen))

    def step(self):
        # A: input view
        from sparsefusion_amd.utils.losses import fusion_loss, render_loss, upsample2x
        img, sil = self.render(self.rays_in)
        loss = render_loss(img, sil, self.target_rgb, self.target_mask, 1.0, 1.0, 1e-3, 1e-3)      # distillation.py:217-241, one launch each way
        self.grads.zero()
        loss.backward()
        self.sync_grads()
        self.optim.step()
        # B: novel view(s) + diffusion distillation (V views per GPU share one batched PLMS call)
        self.grads.zero()
        if self.eft is not None:
            self.features = self.render_features()
        imgs, sils = zip(*[self.render(r) for r in self.rays_novel])
        img, sil = torch.cat(imgs, 0), torch.cat(sils, 0)
        img256, sil256 = upsample2x(img), upsample2x(sil)                       # :287-288 (bilinear x2) on the HIP kernel + its adjoint
        with torch.no_grad():
            latents = self.vae.encode(img256 * 2 - 1).mode() * self.z_scale          # distillation.py:299
            from sparsefusion_amd.distributed import all_gather_latents
            if self.multi and self.time_collectives:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            self.step_latents = all_gather_latents(latents)        # latents of all novel views of this step (8(e))
            if self.multi and self.time_collectives:
                torch.cuda.synchronize()
                self.coll_us["all_gather_latents"].append((time.perf_counter() - t0) * 1e6)
            mt = self.draw_max_thres()
            n_steps = 50 if mt >= .99 else min(int(mt * 100), 50)
            self.evals_run += n_steps + 1 if n_steps else 0
            pred_x0, x_noisy, noise, acp = self.plms.sample(latents, cond_images=self.features, use_tqdm=False,
                                                            return_noise=True, max_thres=mt)
            pred_img = ((self.vae.decode(pred_x0 / self.z_scale) + 1) * 0.5).clip(0.0, 1.0)   # distillation.py:309
        loss = fusion_loss(img256, sil256, pred_img, 1 - acp, 1e-3, 1e-3)       # (1 - a_bar) * L1 + opacity + entropy (:310-343)
        loss = loss + self.percep(img256, pred_img, normalize=True).mean() * self.lambda_percep          # :312-314
        loss.backward()
        self.sync_grads()
        self.optim.step()


def time_region(fn, iters):
    """GPU-side milliseconds per call (events on the current stream, which is the stream the plans launch on)."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 (MI355X_MICROARCH.md; AMD's 2:1-sparsity figures are not priced against)


def _op_weight_bytes(o):
    """bf16 weight bytes one conv / linear op streams (each read once per eval)."""
    from sparsefusion_amd.unet import OP_CONV, OP_FCONV
    if o.type == OP_FCONV:
        return 2 * o.i[5] * (o.i[3] + o.i[4]) * o.i[8] * o.i[8]
    if o.type == OP_CONV:
        return 2 * o.i[6] * o.i[3] * o.i[9] * o.i[10]
    return 0


def _fconv_flops(o):
    """algorithmic 2 * M * N * K of a fused conv op: i = (B, H, W, C1, C2, Cout, ldc, co_off, k, ...)."""
    return 2.0 * o.i[0] * o.i[1] * o.i[2] * o.i[5] * (o.i[3] + o.i[4]) * o.i[8] * o.i[8]


def _graph_time_ms(op_array, n_ops, reps=30):
    """Milliseconds per replay of a hipGraph holding exactly `op_array[:n_ops]`, ONE event pair around `reps` replays on the
    launch stream: no per-op events inside the timed region (an event pair around a kernel costs more than a kernel boundary)."""
    from sparsefusion_amd import _lib
    lib = _lib.lib()
    run = lambda: _lib.check(lib.sf_plan_run(op_array, n_ops, _lib.stream_ptr()), "bench sub-plan")
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def unet_roofline(hp, B=1):
    """Roofline of the UNet's dominant kernel family -- the fused GroupNorm | LayerNorm + conv launches (k_conv_fused,
    k_conv_fused_pipe and their _pair forms) -- measured IN THIS RUN: the fused-conv launches of one eval (batch B), and only
    those, are captured into a hipGraph in plan order and replayed; time per launch = graph time / launches.  Units are
    consistent: everything is per LAUNCH (a conv1 || res_conv pair is two ops in one launch).  Both rooflines are stated
    (SURVEY.md 8(d) "governing roofline"): weight bytes / HBM peak and 2MNK / dense bf16 MFMA peak."""
    from sparsefusion_amd import _lib
    from sparsefusion_amd.unet import OP_CONV, OP_FCONV
    unet = hp.unet
    ctx = unet.begin_sampling(hp.features[:B], torch.linspace(-3, 3, 4, device=hp.dev))
    unet.eval_prepared(ctx, torch.zeros(B, 4, 32, 32, device=hp.dev), 0)
    plan = ctx["plan"]
    ops = [plan.body_array[k] for k in range(plan.n_body_ops)]
    # pairs stay adjacent: both halves are FCONV ops -- or (r04, the 4x4 gca blocks) a res_conv and the GlobalContext pooling op that
    # shares its launch (k_gca_pool_rc): that op belongs to the launch and comes along
    idx = [k for k, o in enumerate(ops) if o.type == OP_FCONV or (k and ops[k - 1].type == OP_FCONV and ops[k - 1].flags & 16)]
    sub = (_lib.SfOp * len(idx))(*[ops[k] for k in idx])
    n_launch = len(idx) - sum(1 for k in idx if ops[k].type == OP_FCONV and ops[k].flags & 16)
    fconv_bytes = int(sum(_op_weight_bytes(ops[k]) for k in idx))
    fconv_flops = float(sum(_fconv_flops(ops[k]) for k in idx if ops[k].type == OP_FCONV))
    fconv_ms = _graph_time_ms(sub, len(idx))
    eval_ms = _graph_time_ms(plan.body_array, plan.n_body_ops)
    all_conv_bytes = int(sum(_op_weight_bytes(o) for o in ops if o.type in (OP_CONV, OP_FCONV)))
    achieved = fconv_bytes / (fconv_ms * 1e-3) / 1e9
    tflops = fconv_flops / (fconv_ms * 1e-3) / 1e12
    return {"bound": "hbm", "kernel": "k_conv3s / k_conv_fused / k_conv_fused_pipe / _pair / _rc / k_gca_pool_rc / k_conv4_gn / k_lin4_ln / k_lin4_attn (GroupNorm | LayerNorm | attention core + conv in one launch)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "traffic_note": "not collected in this run (--no-traffic, B > 1 or a multi-rank run): see profiles/r06_unet_eval_b1_pmc.json",
            "batch": B, "launches": n_launch, "ops": len(idx), "avg_launch_us": round(fconv_ms / n_launch * 1e3, 2),
            "algorithmic_bytes_per_launch": fconv_bytes // n_launch, "algorithmic_bytes_per_eval": fconv_bytes,
            "fused_conv_ms_per_eval": round(fconv_ms, 4),
            "timing_note": "one HIP-event pair on the launch stream around 30 replays of a hipGraph that holds ONLY these launches, in plan order",
            "mfma": {"achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_PEAK_TFLOPS, 5),
                     "gflop_per_eval": round(fconv_flops / 1e9, 1)},
            "whole_eval": {"ms": round(eval_ms, 4), "ops": plan.n_body_ops, "weight_bytes": all_conv_bytes,
                           "GBs": round(all_conv_bytes / (eval_ms * 1e-3) / 1e9, 1),
                           "frac": round(all_conv_bytes / (eval_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}


def measure_fconv_counters(timeout_s=300):
    """PMC figures of the fused-conv family for `roofline.traffic`, `roofline.mfma_busy` and `roofline.hbm_gbs_counter` (north_star: "rocprof HBM
    GB/s and MFMA-busy counters"): a child process replays three B = 1 evals (tools/unet_eval_loop.py, plain launches) under rocprofv3, three
    passes of `--kernel-trace --pmc <one group>` -- FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES + SQ_WAIT_ANY + SQ_WAVE_CYCLES + GRBM_GUI_ACTIVE,
    counters in their own runs as MI355X_MICROARCH.md prescribes (tools/unet_pmc.py; FETCH_SIZE x 2: the counter reports half of a wide coalesced
    read on gfx950).  Returns (dict or None, note)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import unet_pmc
        acc = unet_pmc.collect(1, 3, timeout_s=timeout_s)
        d = unet_pmc.summarise(acc, 3, pred=lambda k: any(s_ in k for s_ in unet_pmc.FUSED)).get("selected")
        if not d or d.get("fetch_bytes_per_dispatch") is None:
            return None, "the counter passes recorded no fused-conv dispatch"
        return d, ("per-dispatch means over the fused-conv dispatches (k_conv_fused* / k_conv3s / k_conv4_gn / k_lin4_ln / k_lin4_attn / k_gca_pool_rc) of a child process "
                   "(3 evals, B = 1, plain launches) under rocprofv3 --kernel-trace --pmc, one counter group per pass; FETCH_SIZE KiB x 1024 x 2 (gfx950)")
    except Exception as e:                                            # noqa: BLE001 -- a bench line without counters is still a bench line
        return None, "counter passes failed: %r" % (e,)


def lds_conv_roofline(hp):
    """MFMA-bound companion of `roofline`: the LDS-tiled large-M convs of one SD-VAE decode (k_conv3_halo where the layer is a 3x3 /
    stride 1 with operand-type input, k_conv_glds for the other operand-type layers, k_conv_lds for fp32 inputs), event-timed per op
    on the launch stream; FLOPs are the algorithmic 2*M*N*K of each layer (padding and out-of-image taps not counted)."""
    import ctypes as C
    from sparsefusion_amd import _lib
    from sparsefusion_amd.unet import OP_CONV
    hp.vae.decode(torch.zeros(1, 4, 32, 32, device=hp.dev))
    plan = hp.vae._plan("dec", 1, hp.dev)
    ms = (C.c_float * len(plan.ops))()
    lib = _lib.lib()
    acc = np.zeros(len(plan.ops))
    for it in range(4):
        _lib.check(lib.sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), ms))
        if it:
            acc += np.array(list(ms))
    acc /= 3
    flops, t, n = 0.0, 0.0, 0
    wf, wt, wn = 0.0, 0.0, 0                                       # the layers of >= 128 output tiles (one per two CUs or more)
    for o, m in zip(plan.ops, acc):
        if o.type == OP_CONV and o.i[14] >= 256:                   # tile code 256 + n-fragments = the LDS-tiled kernels
            B, Cin, Ho, Wo, Cout, k = o.i[0], o.i[3], o.i[4], o.i[5], o.i[6], o.i[9]
            fl = 2.0 * B * Ho * Wo * Cout * Cin * k * k
            flops += fl
            t += float(m)
            n += 1
            bnf = (o.i[14] - 256) & 15
            if ((B * Ho * Wo + 127) // 128) * ((Cout + 16 * bnf - 1) // (16 * bnf)) >= 128:
                wf, wt, wn = wf + fl, wt + float(m), wn + 1
    if not n:
        return None
    achieved = flops / (t * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": "k_conv3_halo / k_conv_glds / k_conv_lds (SD-VAE decode, %d LDS-tiled layers)" % n, "achieved": round(achieved, 1), "peak": 2500.0,
           "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4), "gflop": round(flops / 1e9, 1), "ms": round(t, 3)}
    if wn:
        # the same sum over the layers with at least 128 tiles only: the set the r02 / mid-r03 lines measured (the 64-tile 32 x 32
        # layers ran on the weight-streaming kernel then and were not part of this object)
        wa = wf / (wt * 1e-3) / 1e12
        out["layers_of_128_tiles_or_more"] = {"layers": wn, "achieved": round(wa, 1), "frac": round(wa / 2500.0, 4), "gflop": round(wf / 1e9, 1),
                                              "ms": round(wt, 3)}
    return out


def cpu_baseline(max_thres):
    """The CPU oracle (a port: the reference has no CPU path for its CUDA kernels) timed on this host, on a
    bounded sample of the same workload, scaled to one step."""
    from oracle import lpips_ref, ngp_ref, unet_ref, vae_ref
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from unet_common import spec
    cores = min(32, os.cpu_count() or 1)          # torch CPU kernels stop scaling (and oversubscribe) beyond ~32 threads
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    p = ngp_ref.init_params(bound=4, seed=1, table_std=0.5, sigma_bias=-3.0)
    pl = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "aabb" not in k else v) for k, v in p.items()}
    o, d = ngp_ref.circle_rays(32, view=3)                          # 1024 of the 16384 rays
    uc, uf = torch.rand(1024, 64), torch.rand(1024, 64)
    t0 = time.time()
    r = ngp_ref.render_run(pl, o, d, u_coarse=uc, u_fine=uf, bg_color=0.0, training=True)
    (r["image"].mean() + r["weights_sum"].mean()).backward()
    t_render = (time.time() - t0) * 16                              # -> 16384 rays
    sd = unet_ref.init_state(spec("canonical"), seed=0)
    x, cond = torch.randn(1, 4, 32, 32), torch.randn(1, 256, 32, 32)
    ls = unet_ref.log_snr(torch.tensor([0.4]))
    with torch.no_grad():
        unet_ref.unet_forward(sd, x, ls, cond)                      # warm-up
        t0 = time.time()
        for _ in range(3):
            unet_ref.unet_forward(sd, x, ls, cond)
        t_eval = (time.time() - t0) / 3
    vsd = vae_ref.init_state(vae_ref.vae_param_spec(vae_ref.CANONICAL), seed=0)
    with torch.no_grad():
        t0 = time.time()
        vae_ref.encode_mode(vsd, vae_ref.CANONICAL, torch.rand(1, 3, 256, 256) * 2 - 1)
        vae_ref.decode(vsd, vae_ref.CANONICAL, torch.randn(1, 4, 32, 32))
        t_vae = time.time() - t0
    lsd = lpips_ref.init_state(0)
    pr = torch.rand(1, 3, 256, 256, requires_grad=True)
    t0 = time.time()
    lpips_ref.lpips(lsd, pr, torch.rand(1, 3, 256, 256), normalize=True).sum().backward()
    t_lpips = time.time() - t0
    n_evals = min(int(max_thres * 100), 50) + 1
    step_s = 2 * t_render + n_evals * t_eval + t_vae + t_lpips
    return {"value": round(1.0 / step_s, 5), "unit": "views/s", "cores": cores, "kind": "port",
            "ms_per_step": round(step_s * 1e3, 1), "unet_eval_ms": round(t_eval * 1e3, 1),
            "ngp_render_fwd_bwd_ms": round(t_render * 1e3, 1), "vae_enc_dec_ms": round(t_vae * 1e3, 1), "lpips_fwd_bwd_ms": round(t_lpips * 1e3, 1),
            "sample": f"1 NGP render fwd+bwd on 1024/16384 rays (x16) + 3 UNet evals B=1 (x{n_evals}/3) + 1 VAE encode + decode + 1 LPIPS fwd+bwd, oracle fp32, "
                      f"{cores} threads of {os.cpu_count()}; the reference has no CPU path for grid-encode/near-far (port)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--max-thres", type=float, default=None,
                    help="fixed max_thres of every step's PLMS call (the r01-r04 lines: 0.5).  Default (r05): drawn anew on every step, uniformly "
                         "in [0.5, 0.99) -- always the full 50 steps = 51 UNet evals, the upper end of what the reference's own draw "
                         "(distillation.py:303, U[0, .99)) produces, and a fresh schedule / UNet time table per step as in the reference")
    ap.add_argument("--views-per-gpu", type=int, default=1, help="novel views distilled per GPU and step (BASELINE config 4: 4)")
    ap.add_argument("--total-views", type=int, default=0,
                    help="strong scaling: this many novel views per step in total, block-sharded over the ranks (overrides --views-per-gpu)")
    ap.add_argument("--check-replicas", action="store_true", help="(default when N > 1) assert after every step that all ranks hold bit-identical NGP parameters")
    ap.add_argument("--no-check-replicas", action="store_true")
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3, 4),
                    help="BASELINE configs[k]: 1 = 2 input views, cached features (default); 2 = 6 input views, EFT feature render every step; "
                         "3 = 4 novel views per GPU; 4 = fp16-operand UNet, full 50-step PLMS trajectory (max_thres 0.999), and the 512^2 "
                         "evaluation render through render_batched timed beside the step")
    ap.add_argument("--no-also-measured", action="store_true", help="skip the short configs[3] (B = 4) measurement the default line carries")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter pass (child process) that fills roofline.traffic")
    args = ap.parse_args()
    if args.config == 3 and args.views_per_gpu == 1:
        args.views_per_gpu = 4
    if args.config == 4:
        args.max_thres = 0.999                                   # plms.py:60-66: the full trajectory, 50 steps = 51 UNet evals
    thres_range = (0.5, 0.99) if args.max_thres is None else None
    if args.max_thres is None:
        args.max_thres = 0.5                                     # (the value the eval count below is derived from; every draw gives 50 steps)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` spawns its own ranks, as the reference's demo does (demo.py:180 mp.spawn(fit, nprocs=gpus),
        # :22 init_process_group): re-exec under torch.distributed.run on this node, one rank per GPU, rendezvous on 127.0.0.1
        # (the container hostname may not resolve).  Under torchrun (WORLD_SIZE set) this branch is never taken.
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if args.gpus > 1 and args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d of the launcher" % (args.gpus, world))
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")         # "gloo": dry-run of the multi-rank path on fewer GPUs than ranks
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SF_BENCH_FORCE_DIST=1 (with SF_DIST_SINGLE_RANK_COLLECTIVES=1 for sparsefusion_amd.distributed): ONE rank still initialises
    # the process group and issues every collective of the step -- the only way a 1-GPU box can put RCCL under this code path
    # (tests/test_gpu_bench_multirank.py::test_single_rank_rccl)
    multi_rank = world > 1 or os.environ.get("SF_BENCH_FORCE_DIST") == "1"
    if multi_rank:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)         # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    strong = args.total_views > 0
    if strong:
        from sparsefusion_amd.distributed import shard_views
        if args.total_views % world:
            raise SystemExit("--total-views must be a multiple of the number of ranks (equal shards for the latent all-gather)")
        args.views_per_gpu = len(shard_views(args.total_views, rank, world))
    n_in = 6 if args.config == 2 else 2
    hp = HotPath(dev, rank, world, args.max_thres, args.views_per_gpu, n_input_views=n_in, eft_features=args.config == 2,
                 unet_operand="f16" if args.config == 4 else None, thres_range=thres_range)
    hp.multi = multi_rank
    hp.check_replicas = (multi_rank or args.check_replicas) and not args.no_check_replicas
    for _ in range(args.warmup):
        hp.step()

    def barrier():
        if multi_rank:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        hp.step()
    barrier()
    dt = torch.tensor([time.time() - t0], device=dev)
    if multi_rank:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    ms_per_step = float(dt) / args.steps * 1e3
    hp.verify_replicas()                                            # once, OUTSIDE the timed region
    total32 = None
    if args.config == 1 and args.views_per_gpu == 1 and not strong and not args.no_also_measured and 32 % world == 0:
        # BASELINE configs[3] as the strong-scaling experiment it names ("32 novel views per step sharded 4/GPU ... >= 3.5x at 8 GPUs"):
        # 32 novel views per step in total, block-sharded over the ranks (demo.py:59 is the replica split this replaces), at EVERY N --
        # the N = 1 line carries 32 views on one GPU, so value(N) / value(1) of this object is the strong-scaling ratio from driver
        # records alone.  1 warm-up + 2 timed steps behind the headline's timed region, barrier + max over ranks like the headline.
        hp.set_views(32 // world)
        hp._sctx = None
        hp.step()
        barrier()
        t0 = time.time()
        for _ in range(2):
            hp.step()
        barrier()
        dt32 = torch.tensor([time.time() - t0], device=dev)
        if multi_rank:
            torch.distributed.all_reduce(dt32, op=torch.distributed.ReduceOp.MAX)
        ms32 = float(dt32) / 2 * 1e3
        total32 = {"workload": "BASELINE configs[3]: 32 novel views per step in total, block-sharded %d per GPU over %d GPU(s) (UNet at B = %d), "
                               "latents all-gathered, NGP gradients all-reduced" % (32 // world, world, 32 // world),
                   "scaling": "strong", "total_views": 32, "views_per_gpu": 32 // world, "n_gpus": world, "steps": 2, "warmup": 1,
                   "ms_per_step": round(ms32, 3), "value": round(32 / (ms32 * 1e-3), 3), "unit": "views/s"}
        hp.set_views(1)
        hp._sctx = None
    multi = None
    if multi_rank:                                                  # after the timed region: what the collectives cost, on every rank
        import torch.distributed as dist
        hp.time_collectives = True
        for _ in range(2):
            hp.step()
        barrier()
        med = lambda v: float(np.median(v)) if v else None
        multi = {"world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                 "all_gather_latents_us": med(hp.coll_us["all_gather_latents"]), "all_reduce_grads_us": med(hp.coll_us["all_reduce_grads"]),
                 "all_reduce_bytes": int(hp.grads.flat.numel() * hp.grads.flat.element_size()), "collectives_per_step": {"all_gather": 1, "all_reduce": 2},
                 "replicas_identical": bool(getattr(hp, "replicas_ok", None)) if hp.check_replicas else None,
                 "replicas_check": "once after the timed region (r04: inside every timed step)",
                 "timing_note": "host wall time between device synchronisations around each collective, median over 2 extra steps after the timed region"}

    if rank == 0:
        n_evals = 51 if args.max_thres >= .99 else min(int(args.max_thres * 100), 50) + 1
        res = {
            "metric": "novel views/sec, distillation steps (2 NGP renders fwd+bwd + VAE enc/dec + %d-eval PLMS + LPIPS), 256^2 / 32x32 latents, "
                      "2-view synthetic hydrant" % n_evals,
            "value": round(world * args.views_per_gpu / (ms_per_step * 1e-3), 4), "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": ("fp16" if args.config == 4 else "bf16") + " MFMA operands / f32 accumulate (UNet); f32 (NGP render)", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: single MI355X, 6 input views, EFT feature render (32x32 rays x 20 depths x 6 views) of every novel view inside the step"
                                    if args.config == 2 else
                                    "BASELINE configs[4] on %d GPU(s): fp16-operand UNet (libsparsefusion_hip_f16.so), full 50-step PLMS trajectory, 512^2 evaluation render timed beside the step (breakdown_ms.render_batched_512)" % world
                                    if args.config == 4 else
                                    "BASELINE configs[1]: single MI355X" if args.views_per_gpu == 1 and world == 1 else
                                    "BASELINE configs[3]-style view sharding: %d GPU(s) x %d novel views per step" % (world, args.views_per_gpu)) +
                                   ", 256^2 hydrant-like synthetic scene, %d input views, " % n_in +
                                   "32x32 latent UNet (400.68M params, B=%d per GPU) + NGP render 128x128 rays x (64+64) samples; "
                                   "SD-VAE encode 256^2 -> 32x32x4 and decode back (83.65M params) and LPIPS-VGG16 fwd+bwd at 256^2 every step; "
                                   "%s" % (args.views_per_gpu, "max_thres drawn per step in [0.50, 0.99) (50 PLMS steps every step; a new schedule and "
                                                                "UNet time table per step, as distillation.py:303 causes)" if thres_range
                                           else "max_thres=%.2f fixed" % args.max_thres),
                       "views_per_gpu": args.views_per_gpu, "unet_evals_per_step": n_evals, "rays_per_render": 16384,
                       "parallelism": "view-sharded replicas x%d, RCCL all-gather(latents) + all-reduce(NGP grads)" % world},
        }
        # component timings + roofline of the dominant kernel (after the timed region)
        lp_a, lp_b = torch.rand(1, 3, 256, 256, device=dev, requires_grad=True), torch.rand(1, 3, 256, 256, device=dev)
        res["breakdown_ms"] = {
            "ngp_render_fwd": round(time_region(lambda: hp.render(hp.rays_novel[0]), 5), 3),
            "unet_eval_wall": round(time_region(lambda: hp.unet.forward_with_cond_scale(
                torch.zeros(1, 4, 32, 32, device=dev), torch.zeros(1, device=dev), cond_images=hp.features[:1]), 10), 3),
            "unet_eval_in_sampler": round(time_region(lambda: hp.unet.eval_prepared(hp.sampler_ctx(), hp.sampler_x, 1), 20), 3),
            "vae_encode": round(time_region(lambda: hp.vae.encode(torch.zeros(1, 3, 256, 256, device=dev)), 5), 3),
            "vae_decode": round(time_region(lambda: hp.vae.decode(torch.zeros(1, 4, 32, 32, device=dev)), 5), 3),
            "lpips_fwd_bwd": round(time_region(lambda: hp.percep(lp_a, lp_b).sum().backward(), 5), 3),
        }
        if hp.eft is not None:
            res["breakdown_ms"]["eft_feature_render_per_view"] = round(time_region(lambda: hp.render_features(), 5) / args.views_per_gpu, 3)
        if args.config == 4:
            # the evaluation render of configs[4]: 512 x 512 rays through render_batched in max_ray_batch chunks, no gradients
            # (renderer_df.py:681-717, distillation.py:369-388)
            o5, d5 = pinhole_rays(512, 5, 34, dev)
            hp.ngp.eval()
            kw = {k: v for k, v in vars(hp.opt).items() if k != 'max_ray_batch'}
            with torch.no_grad():
                res["breakdown_ms"]["render_batched_512"] = round(time_region(
                    lambda: hp.ngp.render_batched(o5, d5, batched=True, max_ray_batch=128 * 128, perturb=False, bg_color=1, shading='albedo', **kw), 3), 3)
            hp.ngp.train()
        if args.views_per_gpu > 1:
            Bv = args.views_per_gpu
            ctx_b = hp.unet.begin_sampling(hp.features[:Bv], torch.linspace(-3, 3, 4, device=dev))
            x_b = torch.zeros(Bv, 4, 32, 32, device=dev)
            res["breakdown_ms"]["unet_eval_in_sampler_B%d" % Bv] = round(time_region(lambda: hp.unet.eval_prepared(ctx_b, x_b, 1), 20), 3)
        res["roofline"] = unet_roofline(hp, args.views_per_gpu)    # the batch the step ran the UNet at (configs[3]: B = 4: both rooflines)
        if world == 1 and args.views_per_gpu == 1 and not args.no_traffic:
            pmc, res["roofline"]["traffic_note"] = measure_fconv_counters()
            if pmc:
                rf = res["roofline"]
                rf["traffic"] = pmc["fetch_bytes_per_dispatch"]
                rf["traffic_over_algorithmic"] = round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3)
                rf["mfma_busy"] = pmc.get("mfma_busy")                       # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
                rf["hbm_gbs_counter"] = pmc.get("hbm_gbs_counter")           # (FETCH x 2 + WRITE) / mean plain-launch duration of the trace: a lower bound of the in-graph rate
                rf["hbm_frac_counter"] = round(pmc["hbm_gbs_counter"] / HBM_PEAK_GBS, 4) if pmc.get("hbm_gbs_counter") else None
                rf["counters"] = pmc
        we = res["roofline"]["whole_eval"]
        # the eval as the sampler pays it, at top level (r04 review): against the conv / linear weights the replayed body streams
        # (the time-MLP / time-token weights are hoisted into Unet.time_table once per schedule) and
        # against SURVEY 8(d)'s own figure, 400.68 M parameters x 2 B
        res["roofline"]["frac_whole_eval"] = we["frac"]
        res["roofline"]["frac_whole_eval_survey_8d_bytes"] = round(801.4e6 / (we["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res["roofline"]["whole_eval_note"] = ("whole_eval.ms = one hipGraph replay of the %d-op eval body (what each of the sampler's evals costs); "
                                              "frac_whole_eval = whole_eval.weight_bytes (%.1f MB streamed by the body; the time-path weights are read once per "
                                              "schedule by Unet.time_table, not per eval) / ms / 8 TB/s; frac_whole_eval_survey_8d_bytes uses SURVEY 8(d)'s "
                                              "801.4 MB (all 400.68 M parameters)" % (we["ops"], we["weight_bytes"] / 1e6))
        res["roofline_mfma"] = lds_conv_roofline(hp)
        if world == 1 and args.config == 1 and args.views_per_gpu == 1 and not strong and not args.no_also_measured:
            # the per-GPU regime of BASELINE configs[3] (4 novel views per GPU: the UNet at B = 4), on this GPU, in this run:
            # 1 warm-up + 3 timed steps behind the headline's timed region (utils/load_model.py:58-69 is the same model)
            hp.set_views(4)
            hp._sctx = None
            hp.step()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                hp.step()
            torch.cuda.synchronize()
            ms4 = (time.time() - t0) / 3 * 1e3
            ctx4 = hp.unet.begin_sampling(hp.features[:4], torch.linspace(-3, 3, 4, device=dev))
            x4 = torch.zeros(4, 4, 32, 32, device=dev)
            ev4 = time_region(lambda: hp.unet.eval_prepared(ctx4, x4, 1), 20)
            rf4 = unet_roofline(hp, 4)
            res["also_measured"] = {"config3_B4": {
                "workload": "BASELINE configs[3] per-GPU share on 1 GPU: 4 novel views per step, UNet at B = 4", "steps": 3, "warmup": 1,
                "ms_per_step": round(ms4, 3), "value": round(4 / (ms4 * 1e-3), 3), "unit": "views/s", "unet_eval_ms": round(ev4, 4),
                "roofline": {k: rf4[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us", "algorithmic_bytes_per_eval")},
                "roofline_mfma_same_kernels": rf4["mfma"]}}
            hp.set_views(1)
            hp._sctx = None
            # the reference's own draw (distillation.py:303: max_thres ~ U[0, .99) -> min(int(100 max_thres), 50) PLMS steps, 37.6 on
            # average): 8 steps on the same seeded generator
            keep_range, e0 = hp.thres_range, hp.evals_run
            hp.thres_range = (0.0, 0.99)
            hp.step()
            torch.cuda.synchronize()
            e0, t0 = hp.evals_run, time.time()
            for _ in range(8):
                hp.step()
            torch.cuda.synchronize()
            msr = (time.time() - t0) / 8 * 1e3
            res["also_measured"]["reference_max_thres_draw"] = {
                "workload": "configs[1] with the reference's per-step draw max_thres ~ U[0, .99) (distillation.py:303)", "steps": 8, "warmup": 1,
                "ms_per_step": round(msr, 3), "value": round(1 / (msr * 1e-3), 3), "unit": "views/s", "unet_evals_per_step_mean": round((hp.evals_run - e0) / 8, 2)}
            hp.thres_range = keep_range
        if total32 is not None:
            res.setdefault("also_measured", {})["config3_total32"] = total32
        if multi is not None:
            res["multi_gpu"] = multi
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.max_thres)
        print(json.dumps(res))
    if multi_rank:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
