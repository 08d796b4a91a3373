"""The HIP path under a process group on the GPU box: two ranks share the one test GPU (gloo
rendezvous on 127.0.0.1), so the sharded render drivers and the data-parallel training step run
exactly as they do with one GPU per rank, minus RCCL (BASELINE configs 4 and 5)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def two_rank_results(tmp_path_factory):
    assert torch.cuda.is_available()
    out = tmp_path_factory.mktemp("two_rank")
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), str(out)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return np.load(out / "rank0.npz"), np.load(out / "rank1.npz")


def _nets(dev, synthetic):
    from nerf_simple_amd.utils.nets import Nerf
    net = Nerf(precision="fp16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
    fine = Nerf(precision="fp16").to(dev)
    fine.load_state_dict(synthetic.synthetic_state_dict(7, "structured"))
    return net, fine


def test_sharded_renders_equal_unsharded(two_rank_results, synthetic):
    """render_view_sharded / render_hierarchical_sharded on 2 ranks == the same view rendered by one
    process, bit for bit (jitter keyed by global pixel id or sliced from the caller's table)."""
    from nerf_simple_amd.utils.rendering import render_view, render_hierarchical_view
    from nerf_simple_amd.utils.xyz import spherical_to_pose
    r0, _ = two_rank_results
    dev = torch.device("cuda:0")
    net, fine = _nets(dev, synthetic)
    pose = spherical_to_pose(4, -30, 35)
    cam = [40, 36, synthetic.focal_from_fov(40)]
    with torch.no_grad():
        one = render_view(net, pose, cam, N=64, device_rng=True, seed=11).cpu().numpy()
        u = torch.rand(cam[0] * cam[1], 48, generator=torch.Generator().manual_seed(3)).to(dev)
        one_u = render_view(net, pose, cam, N=48, u=u).cpu().numpy()
        hier = render_hierarchical_view(net, fine, pose, cam, 64, 128, device_rng=True, seed=5).cpu().numpy()
    assert r0["view"].shape == (1440, 4) and np.array_equal(r0["view"], one)
    assert np.array_equal(r0["view_u"], one_u)
    assert np.array_equal(r0["hier"], hier)
    assert np.isfinite(hier).all() and hier[:, :3].min() >= 0 and hier[:, :3].max() <= 1


def test_data_parallel_step_equals_global_batch(two_rank_results, golden, synthetic):
    """GraphedTrainStep(group=...) on 2 ranks x 32 rays: the all-reduced gradient equals the gradient of
    the 64-ray global batch computed by one process (equal shards, per-rank MSE mean; train.py:52-54),
    both ranks end with identical parameters, and the mean of the rank losses is the global loss."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.training import train_step
    r0, r1 = two_rank_results
    dev = torch.device("cuda:0")
    g = golden("train.npz")
    rays, gt, u = (torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in ("rays", "gt", "u"))
    N = int(g["N"])
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    loss = float(train_step(net, opt, rays, gt, N, u=u))
    want = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    got = r0["grads"]
    # the same kernels on the same points; only the float-atomic summation order and the split of the
    # mean over two ranks differ
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()
    assert abs(0.5 * (float(r0["loss"][0]) + float(r1["loss"][0])) - loss) <= 1e-5 * loss
    assert np.array_equal(r0["params"], r1["params"])
    # one all-reduce between the graphs (the default) and the two-bucket, overlapped exchange average to the same
    # gradient (summation order of the split-K atomics aside)
    assert np.abs(r0["grads_one_bucket"] - got).max() <= 2e-5 * np.abs(want).max()
    # the bucketed step: three more replays leave both ranks with identical, finite parameters that moved
    assert np.array_equal(r0["params4"], r1["params4"]) and np.isfinite(r0["params4"]).all()
    assert not np.array_equal(r0["params4"], r0["params"])
    print("2-rank gloo exchange: span / exposed ms", r0["collective_ms"])


def test_data_parallel_exact_step_equals_global_batch(two_rank_results, golden, synthetic):
    """The eager step under the group on the layer-by-layer fp32 path -- a precision='fp32' module and a Nerf(6, 2, 128):
    the all-reduced gradient (one flat bucket) is the global-batch gradient to fp32 round-off, both ranks hold the same
    parameters after the Adam step, the rank losses average to the global loss."""
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.parallel import flat_grad_view
    from nerf_simple_amd.training import train_step
    from nerf_simple_amd.utils.nets import Nerf
    r0, r1 = two_rank_results
    dev = torch.device("cuda:0")
    g = golden("train.npz")
    rays, gt, u = (torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in ("rays", "gt", "u"))
    N = int(g["N"])
    for tag, make in (("exact", lambda: Nerf(precision="fp32")), ("small", lambda: Nerf(6, 2, 128))):
        torch.manual_seed(5)
        m = make().to(dev)
        if tag == "exact":
            m.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        o = FusedAdam(m, lr=5e-4)
        loss = float(train_step(m, o, rays, gt, N, u=u))
        want = flat_grad_view([p for _, p in m.named_parameters()]).cpu().numpy()
        got = r0[f"{tag}_grads"]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (tag, np.abs(got - want).max() / np.abs(want).max())
        assert abs(0.5 * (float(r0[f"{tag}_loss"][0]) + float(r1[f"{tag}_loss"][0])) - loss) <= 1e-6 * loss, tag
        assert np.array_equal(r0[f"{tag}_params"], r1[f"{tag}_params"]), tag
        # Adam's first step is ~ lr * sign(g): identical wherever the gradient is not ~0
        assert np.mean(np.abs(r0[f"{tag}_params"] - o.flat.cpu().numpy()) <= 2e-6) >= 0.97, tag


@pytest.mark.parametrize("mode", ["render", "train"])
def test_bench_self_launch_two_ranks(mode):
    """`python bench.py --gpus 2` starts its own rank processes (no torch.distributed.run needed) and
    rank 0 prints the one JSON line; rehearsed here with both ranks on the one GPU over gloo."""
    env = dict(os.environ, NERF_BENCH_BACKEND="gloo", NERF_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--mode", mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"]["world_size"] == 2 and res["ranks"]["rank_sum_check"] is True
    assert res["value"] > 0 and res["steps"] == 2
    assert res["scaling"] == ("strong" if mode == "render" else "weak")
    assert res["collective_ms"] > 0 and len(res["kernel_ms_per_rank"]) == 2
    assert res["ranks"]["backend"] == "gloo" and res["ranks"]["devices_distinct"] is False      # the rehearsal shares one GPU


@pytest.mark.parametrize("mode", ["render", "train", "train2", "train2_e4m3"])
def test_bench_rccl_single_rank_rehearsal(mode):
    """The multi-GPU step of bench.py over the REAL transport with the one GPU this box has: a process group of one rank
    on backend nccl (= RCCL), every collective issued all the same (NERF_BENCH_FORCE_DIST=1 ->
    parallel.force_collectives): init with device_id, the barrier / MAX all-reduce of the timed loop, the all-gather of
    the pixels after every render, the two asynchronous AVG all-reduces of the gradient buckets beside the replayed
    hipGraphs, all_gather_object in the self-check.  What it cannot show is the wire: more than one rank needs more than
    one GPU."""
    env = dict(os.environ, NERF_BENCH_FORCE_DIST="1", MASTER_PORT=str(free_port()),
               NERF_BENCH_BUCKETS=("2" if mode.startswith("train2") else "1"))
    extra = ["--storage", "e4m3"] if mode.endswith("_e4m3") else []     # the 8-bit storage form through the bucketed launches
    mode = "train" if mode.startswith("train2") else mode
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NERF_BENCH_BACKEND", "NERF_BENCH_SHARE_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--mode", mode, "--no-cpu-baseline", "--no-aux"] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["ranks"]["backend"] == "nccl" and res["ranks"]["rank_sum_check"] is True
    assert res["ranks"]["devices_distinct"] is True and res["collective_ms"] > 0 and len(res["kernel_ms_per_rank"]) == 1
    if mode == "train":
        assert res["collective_ms"] >= res["collective_exposed_ms"] >= 0 and np.isfinite(res["final_loss"])
    print(f"RCCL single-rank rehearsal, {mode}: {res['ms_per_step']:.3f} ms/step, collective {res['collective_ms']:.4f} ms"
          + (f" (exposed {res['collective_exposed_ms']:.4f})" if mode == "train" else ""))


def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` -- rehearsed with two ranks on the one GPU over gloo:
    the ranks take RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (no self-launch), and exactly one JSON
    line reaches stdout."""
    env = dict(os.environ, NERF_BENCH_BACKEND="gloo", NERF_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:5]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"]["world_size"] == 2 and res["scaling"] == "strong"
    assert "cpu_baseline" not in res and "aux" not in res           # N = 1 only
