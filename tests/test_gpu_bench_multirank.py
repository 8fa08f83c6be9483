"""Dry run of the multi-rank path of bench.py on ONE GPU: two ranks share cuda:0, the collectives go through gloo
(SF_BENCH_BACKEND=gloo) -- exercises view sharding (--total-views, strong scaling), the latent all-gather, the in-place
all-reduce of the flat NGP gradient buffer, the replica check (once, after the timed region) and the max-over-ranks timing / JSON line.
RCCL with more than one rank only runs on the driver's multi-GPU node; the last two tests put RCCL under the same call sites with ONE rank
(a one-rank group that issues its collectives anyway: SF_BENCH_FORCE_DIST / SF_DIST_SINGLE_RANK_COLLECTIVES)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_strong_scaling_dry_run():
    env = dict(os.environ, SF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-views", "2",
           "--steps", "2", "--warmup", "1", "--max-thres", "0.06", "--no-cpu-baseline", "--check-replicas"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["views_per_gpu"] == 1
    assert res["value"] > 0 and res["steps"] == 2
    # the run verifies itself: the world size the process group saw, the collectives' cost and the replica check are in the line
    mg = res["multi_gpu"]
    assert mg["world_size_seen"] == 2 and mg["backend"] == "gloo" and mg["replicas_identical"] is True
    assert mg["all_gather_latents_us"] > 0 and mg["all_reduce_grads_us"] > 0 and mg["all_reduce_bytes"] == (929336 * 2 + 6532) * 4
    rf = res["roofline"]                                            # measured in this run, per launch, both rooflines
    assert rf["traffic"] is None and rf["launches"] <= rf["ops"] and rf["avg_launch_us"] > 0 and rf["mfma"]["frac"] > 0
    assert abs(rf["algorithmic_bytes_per_launch"] * rf["launches"] - rf["algorithmic_bytes_per_eval"]) < rf["launches"]


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (how the driver invokes the N = 1 bench): bench.py re-execs itself under
    torch.distributed.run, one rank per GPU (the reference spawns its ranks itself too: demo.py:180, :22)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SF_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--max-thres", "0.06",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["multi_gpu"]["world_size_seen"] == 2
    assert res["multi_gpu"]["replicas_identical"] is True and res["value"] > 0
    # r05: the N > 1 line also carries BASELINE configs[3] as the strong-scaling experiment it names -- 32 novel views per step in
    # total, block-sharded over the ranks (16 per rank here) -- next to the weak-scaling headline
    t32 = res["also_measured"]["config3_total32"]
    assert t32["scaling"] == "strong" and t32["total_views"] == 32 and t32["views_per_gpu"] == 16 and t32["n_gpus"] == 2
    assert t32["value"] > 0 and abs(t32["value"] - 32 / (t32["ms_per_step"] * 1e-3)) < 1e-2 * t32["value"]
    assert "once after the timed region" in res["multi_gpu"]["replicas_check"]


def test_config2_eft_feature_render_in_the_step():
    """BASELINE configs[2] as a bench workload: 6 input views, the EFT feature render of the novel view inside every step."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "1", "--warmup", "1", "--max-thres", "0.06",
           "--no-cpu-baseline", "--no-traffic"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "configs[2]" in res["config"]["workload"] and res["breakdown_ms"]["eft_feature_render_per_view"] > 0 and res["value"] > 0


def test_config4_fp16_unet_full_plms_and_512_render():
    """BASELINE configs[4] as a bench workload (one GPU's share): fp16-operand UNet, the full 50-step PLMS trajectory, and the 512^2
    evaluation render through render_batched timed beside the step."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-traffic"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "configs[4]" in res["config"]["workload"] and res["dtype"].startswith("fp16") and res["config"]["unet_evals_per_step"] == 51
    assert res["breakdown_ms"]["render_batched_512"] > 0 and res["value"] > 0


def test_default_line_carries_the_b4_regime():
    """The default (configs[1]) line also measures configs[3]'s per-GPU regime -- 4 novel views per step, UNet at B = 4 -- in the same run."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--max-thres", "0.06", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    b4 = res["also_measured"]["config3_B4"]
    assert b4["ms_per_step"] > 0 and b4["value"] > 0 and b4["unet_eval_ms"] > 0 and 0 < b4["roofline"]["frac"] < 1
    assert res["config"]["views_per_gpu"] == 1 and res["roofline"]["batch"] == 1
    rf = res["roofline"]                # the counter pass of the default line: measured HBM fetch per fused-conv launch, or null with a reason
    assert (rf["traffic"] is None and rf["traffic_note"]) or (rf["traffic"] > 1e6 and 0.5 < rf["traffic_over_algorithmic"] < 10)
    # r05: the whole-eval fraction at top level (against the streamed bytes and against SURVEY 8(d)'s 801.4 MB), the 32-views-on-one-GPU
    # baseline of the strong-scaling experiment, and a run with the reference's own max_thres draw
    assert 0 < rf["frac_whole_eval"] < rf["frac_whole_eval_survey_8d_bytes"] < 1 and rf["whole_eval"]["ms"] > 0
    t32 = res["also_measured"]["config3_total32"]
    assert t32["n_gpus"] == 1 and t32["views_per_gpu"] == 32 and t32["value"] > 0
    rd = res["also_measured"]["reference_max_thres_draw"]
    assert rd["ms_per_step"] > 0 and 20 <= rd["unet_evals_per_step_mean"] <= 51


def test_default_max_thres_is_drawn_per_step():
    """r05: without --max-thres every step draws its own max_thres in [0.5, 0.99) (a new schedule and UNet time table
    per step, as distillation.py:303 causes; always 50 PLMS steps = 51 evals)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--no-also-measured"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "drawn per step" in res["config"]["workload"] and res["config"]["unet_evals_per_step"] == 51 and res["value"] > 0


_HELPERS_ON_RCCL = r"""
import torch, torch.distributed as dist
from sparsefusion_amd import distributed as D
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and not D._no_exchange()
g = torch.Generator(device="cpu").manual_seed(3)
lat = torch.randn(4, 4, 32, 32, generator=g).to(dev)
out = D.all_gather_latents(lat, check=True)
assert out.data_ptr() != lat.data_ptr() and torch.equal(out, lat)             # went through all_gather_into_tensor
net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Linear(64, 16)).to(dev)
bucket = D.FlatGradBucket(net.parameters())
bucket.zero()
net(torch.randn(8, 32, generator=g).to(dev)).square().sum().backward()
ref = bucket.flat.clone()
work = bucket.all_reduce(async_op=True)
assert work is not None
work.wait()
torch.cuda.synchronize()
assert torch.equal(bucket.flat, ref)                                           # mean over one rank
gr = [p.grad.clone() for p in net.parameters()]
D.all_reduce_grads(list(net.parameters()))
assert all(torch.equal(a, p.grad) for a, p in zip(gr, net.parameters()))
w0 = [p.detach().clone() for p in net.parameters()]
D.broadcast_params(net)
assert all(torch.equal(a, p) for a, p in zip(w0, net.parameters()))
assert D.replicas_identical(net) is True                                       # fp64 MAX / MIN all-reduces
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
"""


def _single_rank_env():
    env = {k: v for k, v in os.environ.items() if k not in ("SF_BENCH_BACKEND",)}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90),
               SF_DIST_SINGLE_RANK_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return env


def test_single_rank_rccl_under_the_distributed_helpers():
    """Every call site of sparsefusion_amd.distributed on the "nccl" (= RCCL) backend with device tensors: all_gather_into_tensor, the
    asynchronous in-place all-reduce of the flat gradient bucket, the generic gradient all-reduce, the flat broadcast, the fp64 MAX / MIN
    replica check, barrier -- one rank, so every result must equal its input."""
    out = subprocess.run([sys.executable, "-c", _HELPERS_ON_RCCL], env=_single_rank_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_SINGLE_RANK_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_single_rank_rccl():
    """bench.py's multi-rank path -- process group on the "nccl" backend with device_id, latent all-gather, flat gradient all-reduce twice a
    step on RCCL's stream while the step's hipGraphs and side streams run, barrier + max-over-ranks timing, replica check -- with ONE rank
    on the box's one GPU."""
    env = _single_rank_env()
    env["SF_BENCH_FORCE_DIST"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--max-thres", "0.06",
           "--no-cpu-baseline", "--no-traffic", "--no-also-measured"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    mg = res["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["world_size_seen"] == 1 and mg["replicas_identical"] is True
    assert mg["all_gather_latents_us"] > 0 and mg["all_reduce_grads_us"] > 0 and res["n_gpus"] == 1 and res["value"] > 0
