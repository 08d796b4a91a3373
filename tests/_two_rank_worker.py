"""Worker of tests/test_gpu_multirank.py: one rank of a 2-rank job on ONE GPU (gloo rendezvous, both
ranks on cuda:0), running the HIP path under a process group.  Results go to the directory in argv[1]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    from nerf_simple_amd import _lib, parallel
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_view_sharded, render_hierarchical_sharded
    from nerf_simple_amd.utils.xyz import spherical_to_pose
    _lib.lib()
    res = {}
    pose = spherical_to_pose(4, -30, 35)
    cam = [40, 36, synthetic.focal_from_fov(40)]               # 1440 pixels: 720 per rank
    net = Nerf(precision="fp16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
    fine = Nerf(precision="fp16").to(dev)
    fine.load_state_dict(synthetic.synthetic_state_dict(7, "structured"))
    with torch.no_grad():
        res["view"] = render_view_sharded(net, pose, cam, N=64, device_rng=True, seed=11).cpu().numpy()
        u = torch.rand(cam[0] * cam[1], 48, generator=torch.Generator().manual_seed(3)).to(dev)
        res["view_u"] = render_view_sharded(net, pose, cam, N=48, u=u).cpu().numpy()
        res["hier"] = render_hierarchical_sharded(net, fine, pose, cam, 64, 128, device_rng=True, seed=5).cpu().numpy()
    # data-parallel training step: each rank holds half of golden G6's rays
    g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"))
    rays, gt, uu = (torch.from_numpy(np.ascontiguousarray(g[k])) for k in ("rays", "gt", "u"))
    N = int(g["N"])
    half = rays.shape[0] // world
    sl = slice(rank * half, (rank + 1) * half)
    tnet = Nerf(precision="bf16").to(dev)
    tnet.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    parallel.broadcast_parameters(tnet)
    opt = FusedAdam(tnet, lr=5e-4)
    # the default exchange (one all-reduce between the graphs) on a twin module: its averaged gradient is the reference
    # for the bucketed, overlapped form below
    tnet1 = Nerf(precision="bf16").to(dev)
    tnet1.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    # group=None is the DEFAULT process group, as in train_step and everywhere in parallel.py: the replicas must exchange
    # (round 3's `group is not None and ...` silently skipped the all-reduce here and let the replicas diverge)
    step1 = GraphedTrainStep(tnet1, FusedAdam(tnet1, lr=5e-4), half, N, timing=True)
    assert step1.exchange and not step1.bucketed and step1.graph_a2 is None
    step1.step(rays[sl].to(dev), gt[sl].to(dev), u=uu[sl].to(dev))
    torch.cuda.synchronize()
    res["grads_one_bucket"] = step1.grads.cpu().numpy()
    assert step1.collective_times()[0] > 0
    stepper = GraphedTrainStep(tnet, opt, half, N, group=dist.group.WORLD, timing=True, buckets=2)
    assert stepper.bucketed and stepper.graph_a2 is not None          # two gradient launches, two overlapped exchanges
    loss = stepper.step(rays[sl].to(dev), gt[sl].to(dev), u=uu[sl].to(dev))
    torch.cuda.synchronize()
    res["grads"] = stepper.grads.cpu().numpy()
    res["loss"] = np.array([float(loss)])
    res["params"] = opt.flat.cpu().numpy()
    for _ in range(3):                                                # replays: the exchange pattern repeats cleanly
        stepper.step(rays[sl].to(dev), gt[sl].to(dev), u=uu[sl].to(dev))
    span, exposed = stepper.collective_times()
    assert span >= exposed >= 0.0
    res["params4"] = opt.flat.cpu().numpy()
    res["collective_ms"] = np.array([span, exposed])
    # the eager step under the group, on the exact layer-by-layer path: a precision='fp32' module and one of another size
    from nerf_simple_amd.training import train_step
    from nerf_simple_amd.parallel import flat_grad_view
    for tag, make in (("exact", lambda: Nerf(precision="fp32")), ("small", lambda: Nerf(6, 2, 128))):
        torch.manual_seed(5)                                           # same initial weights on both ranks
        m = make().to(dev)
        if tag == "exact":
            m.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        parallel.broadcast_parameters(m)
        o = FusedAdam(m, lr=5e-4)
        l = train_step(m, o, rays[sl].to(dev), gt[sl].to(dev), N, u=uu[sl].to(dev), group=dist.group.WORLD)
        res[f"{tag}_grads"] = flat_grad_view([p for _, p in m.named_parameters()]).cpu().numpy()
        res[f"{tag}_params"] = o.flat.cpu().numpy()
        res[f"{tag}_loss"] = np.array([float(l)])
    if rank == 0:
        np.savez(os.path.join(out_dir, "rank0.npz"), **res)
    else:
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=res["loss"], params=res["params"], params4=res["params4"],
                 exact_params=res["exact_params"], small_params=res["small_params"],
                 exact_loss=res["exact_loss"], small_loss=res["small_loss"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
