"""The rasterizer autograd op under DistributedDataParallel, as the reference trains
(/root/reference/src/main.py:93-105: one process per GPU, strategy
``ddp_find_unused_parameters_true``; gradient all-reduce fired by ``manual_backward``,
src/model/model_wrapper.py:440).  A tiny ``nn.Linear -> GaussianAdapter -> DecoderSplattingCUDA``
model is wrapped in DDP over RCCL (backend "nccl"), every rank renders its own batch, and the
all-reduced parameter gradients must equal the mean of the per-rank gradients computed in ONE
process.  The rasterizer itself has no parameters and takes part in no collective;
this pins that its autograd node, streams and workspaces coexist with DDP's hooks and buckets.

Needs >= 2 GPUs: skipped (loudly) on the single-GPU test boxes; the single-GPU half of the same
model (forward + backward, unused parameter, finite gradients) runs everywhere."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

pytestmark = pytest.mark.gpu
R, SIZE = 1024, 32


class TinySplatModel(nn.Module):
    def __init__(self):
        super().__init__()
        from latentsplat_amd import decoder as dec
        from latentsplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
        self.adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 1, 1), 4, rotate_sh=lambda sh, rot: sh)
        self.lin = nn.Linear(16, self.adapter.d_in)
        self.never_used = nn.Linear(4, 4)          # needs find_unused_parameters=True, as in src/main.py:98
        self.decoder = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0])

    def forward(self, tokens, cams):
        from latentsplat_amd import decoder as dec
        raw = self.lin(tokens)                                       # (1,1,R,1,1,d_in)
        g = self.adapter.forward(cams["ctx_extrinsics"], cams["ctx_intrinsics"], cams["coordinates"], cams["depths"],
                                 cams["opacities"], raw, (SIZE, SIZE))
        gauss = dec.Gaussians(g.means.reshape(1, -1, 3), g.covariances.reshape(1, -1, 3, 3), g.opacities.reshape(1, -1),
                              g.color_harmonics.reshape(1, R, 3, -1), g.feature_harmonics.reshape(1, R, 4, -1))
        out = self.decoder.forward(gauss, cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (SIZE, SIZE))
        return (out.color ** 2).mean() + (out.feature_posterior.mean ** 2).mean()


def make_batch(seed, dev):
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(SIZE), torch.arange(SIZE), indexing="ij")
    coords = (torch.stack([xs, ys], -1).reshape(R, 2).float() + 0.5) / SIZE
    K = torch.tensor([[0.8, 0, 0.5], [0, 0.8, 0.5], [0, 0, 1.0]])
    ext = torch.eye(4).repeat(2, 1, 1)
    ext[1, 0, 3] = 0.2
    cams = dict(ctx_extrinsics=torch.eye(4).reshape(1, 1, 1, 1, 1, 4, 4), ctx_intrinsics=K.reshape(1, 1, 1, 1, 1, 3, 3),
                coordinates=coords.reshape(1, 1, R, 1, 1, 2), depths=(1.5 + 3 * torch.rand(1, 1, R, 1, 1, generator=g)),
                opacities=0.2 + 0.6 * torch.rand(1, 1, R, 1, 1, generator=g),
                extrinsics=ext[None], intrinsics=K.repeat(2, 1, 1)[None], near=torch.full((1, 2), 0.5), far=torch.full((1, 2), 40.0))
    tokens = torch.randn(1, 1, R, 1, 1, 16, generator=g)
    return tokens.to(dev), {k: v.to(dev) for k, v in cams.items()}


def _local_grads(model, seed, dev):
    model.zero_grad(set_to_none=True)
    tokens, cams = make_batch(seed, dev)
    loss = model(tokens, cams)
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, float(loss)


def test_model_trains_on_one_gpu(hip_device):
    torch.manual_seed(0)
    model = TinySplatModel().to(hip_device)
    grads, loss = _local_grads(model, 100, hip_device)
    assert loss > 0 and set(grads) == {"lin.weight", "lin.bias"}            # never_used gets no gradient
    assert all(torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in grads.values())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(0)
    model = TinySplatModel().to(dev)
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=[rank], find_unused_parameters=True)
    tokens, cams = make_batch(100 + rank, dev)
    ddp(tokens, cams).backward()
    torch.cuda.synchronize(dev)
    got = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    if rank == 0:   # the same two batches in one process, no DDP
        torch.manual_seed(0)
        ref = TinySplatModel().to(dev)
        per_rank = [_local_grads(ref, 100 + r, dev)[0] for r in range(world)]
        want = {n: sum(g[n] for g in per_rank).cpu() / world for n in per_rank[0]}
        out.put((got, want))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="DDP over RCCL needs >= 2 GPUs; this box has "
                    f"{torch.cuda.device_count()} (the driver's multi-GPU tier runs it on the 8-GPU node)")
def test_ddp_allreduced_grads_equal_single_process_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, want = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert set(got) == set(want) == {"lin.weight", "lin.bias"}
    for n in want:
        scale = max(1e-6, float(want[n].abs().max()))
        assert float((got[n] - want[n]).abs().max()) <= 1e-4 * scale, n


def test_bench_multi_process_path_on_one_gpu():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank:
    RANK / LOCAL_RANK / WORLD_SIZE from the environment, barrier + sync on both sides of the timed region,
    MAX over ranks, one JSON line from rank 0) — on a one-GPU box both ranks share cuda:0 and rendezvous
    over gloo (LSR_BENCH_SHARE_GPU=1), so the numbers mean nothing but every line of the N > 1 path runs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--gaussians", "20000", "--views", "4",
           "--no-cpu-baseline", "--no-latency"]
    env = dict(os.environ, LSR_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints, rank 1 stays silent
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert len(out["per_rank_ms_per_step"]) == 2 and max(out["per_rank_ms_per_step"]) == pytest.approx(out["ms_per_step"], abs=1e-4)
    assert len(lines[0]) < 4096 and "roofline" in out and "fwdbwd" in out       # compact line; the full dictionary goes to gpurun_out/bench_full.json
    # the same job WITHOUT a launcher: `python bench.py --gpus 2` starts its own two ranks ...
    cmd2 = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "20000",
            "--views", "4", "--no-cpu-baseline", "--no-latency", "--no-bwd"]
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r2 = subprocess.run(cmd2, env=env2, cwd=root, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, (r2.stdout + r2.stderr)[-3000:]
    out2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert out2["n_gpus"] == 2
    # ... and a launcher that provides a different world size than --gpus asks for is refused, not mis-reported
    r3 = subprocess.run(cmd2, env=dict(env2, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r3.returncode != 0 and "refusing" in (r3.stdout + r3.stderr)
    assert out["value"] == pytest.approx(2 * 4 * 1e3 / out["ms_per_step"], rel=1e-6)   # whole-job rate: both ranks' views over the slowest rank's time
    assert out["cpu_baseline"] is None                  # rank 0 at N = 1 only
