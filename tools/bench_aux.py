#!/usr/bin/env python3
"""Auxiliary measurements of the other BASELINE.json configurations on one GPU
(the headline is bench.py).  Prints one JSON object per line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_simple_amd.utils import synthetic                      # noqa: E402
from nerf_simple_amd.utils.nets import Nerf                      # noqa: E402
from nerf_simple_amd.utils.rendering import render_view, generate_rays  # noqa: E402
from nerf_simple_amd.training import train_step                  # noqa: E402
from nerf_simple_amd.utils.xyz import spherical_to_pose          # noqa: E402

FLOP = 1_186_816
dev = torch.device("cuda:0")
pose = spherical_to_pose(4, -30, 0)
sd = synthetic.synthetic_state_dict(0, "structured")


def timed(fn, warm=1, reps=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def net_of(precision, seed=0):
    n = Nerf(precision=precision).to(dev)
    n.load_state_dict(synthetic.synthetic_state_dict(seed, "structured"))
    return n


which = sys.argv[1:] or ["c2", "c3", "c4", "c5"]
with torch.no_grad():
    if "c2" in which:      # config 2: 400x400, 64 samples, fp32
        net = net_of("fp32")
        cam = [400, 400, synthetic.focal_from_fov(400)]
        dt = timed(lambda: render_view(net, pose, cam, N=64, device_rng=True))
        s = 400 * 400 * 64
        print(json.dumps({"config": "2: 400x400x64 fp32 (exact-f32 MFMA)", "ms": dt * 1e3, "ray_samples_per_s": s / dt,
                          "tflops": s * FLOP / dt / 1e12, "frac_of_157.3TF": s * FLOP / dt / 157.3e12}))
    if "c3" in which:      # config 3 through the one-call image driver
        net = net_of("fp16")
        cam = [800, 800, synthetic.focal_from_fov(800)]
        dt = timed(lambda: render_view(net, pose, cam, N=128, device_rng=True))
        s = 800 * 800 * 128
        print(json.dumps({"config": "3: 800x800x128 fp16 via render_view (ray generation + the fused render kernel)", "ms": dt * 1e3,
                          "ray_samples_per_s": s / dt, "tflops": s * FLOP / dt / 1e12}))
    if "poses" in which:   # the reference's video driver (utils/rendering.py:116-153): 6 poses, 16,000-ray batches, read-back included
        from nerf_simple_amd.utils.rendering import render_poses
        from nerf_simple_amd.utils.xyz import poses_to_render
        net = net_of("fp16")
        cam = [800, 800, synthetic.focal_from_fov(800)]
        views = poses_to_render(4, -30, 6)
        dt = timed(lambda: render_poses(net, views, cam, 16000, N=128, device_rng=True), warm=1, reps=2) / len(views)
        s = 800 * 800 * 128
        print(json.dumps({"config": "3 through render_poses: 6 poses of 800x800x128, 16,000-ray batches, images back on the host",
                          "ms_per_pose": dt * 1e3, "ray_samples_per_s": s / dt}))
    if "c4" in which:      # config 4: 800x800, 64 coarse + 128 fine (192 in the fine pass), one GPU
        from nerf_simple_amd.utils.rendering import render_hierarchical_view
        for prec in ("fp16", "bf16"):
            nc, nf = net_of(prec, 0), net_of(prec, 7)
            cam = [800, 800, synthetic.focal_from_fov(800)]
            dt = timed(lambda: render_hierarchical_view(nc, nf, pose, cam, 64, 128, device_rng=True, seed=3))
            s = 800 * 800 * (64 + 192)
            print(json.dumps({"config": f"4: 800x800 hierarchical 64 + (64+128) {prec}, 1 GPU, ONE library call "
                                        "(raygen, coarse, sample_pdf, fine)", "ms": dt * 1e3,
                              "mlp_evals_per_s": s / dt, "tflops": s * FLOP / dt / 1e12}))
if "pcie" in which:        # reference-compatible jitter: the torch.rand(B,N) stream of the CPU generator (host_rng.py)
    from nerf_simple_amd.utils.rendering import render_nerf
    with torch.no_grad():
        net = net_of("fp16")
        rays = generate_rays(pose, [800, 800, synthetic.focal_from_fov(800)], dev)
        for B in (16000, 640000):
            r = rays[:B].contiguous()
            dt = timed(lambda: render_nerf(r, net, 128, outputs=("rgb", "disp", "acc")), warm=1, reps=3)
            print(json.dumps({"config": f"3 (parity mode): render_nerf B={B} N=128, torch CPU-generator jitter stream continued on the device",
                              "ms": dt * 1e3, "ray_samples_per_s": B * 128 / dt}))
if "pcie" in which:        # the same mode through the image driver: batches of 64,000 rays, jitter drawn ahead
    from nerf_simple_amd.utils.rendering import _render_batched
    with torch.no_grad():
        net = net_of("fp16")
        rays = generate_rays(pose, [800, 800, synthetic.focal_from_fov(800)], dev)
        for bs in (64000, 640000):
            dt = timed(lambda: _render_batched(rays, net, bs, 128, 2, 6, None, False), warm=1, reps=3)
            print(json.dumps({"config": f"3 (parity mode): 800x800x128 image driver, batch_size {bs}, reference jitter drawn "
                              "on the device ahead of the batches", "ms": dt * 1e3, "ray_samples_per_s": 640000 * 128 / dt}))
if "c5g" in which or "c5" in which:   # config 5 through the captured hipGraphs (training.GraphedTrainStep)
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    for N in (64, 128):
        net = net_of("bf16")
        opt = FusedAdam(net, lr=5e-4)
        rays = generate_rays(pose, [800, 800, synthetic.focal_from_fov(800)], dev)[:4096].contiguous()
        gt = torch.rand(4096, 3, device=dev)
        u = torch.rand(4096, N, device=dev)
        stepper = GraphedTrainStep(net, opt, 4096, N)
        dt = timed(lambda: stepper.step(rays, gt, u=u), warm=20, reps=500)
        s = 4096 * N
        print(json.dumps({"config": f"5: train step 4096 rays x {N} bf16, hipGraph replay (fwd+bwd+FusedAdam+repack)",
                          "ms": dt * 1e3, "ray_samples_per_s": s / dt, "tflops_fwd_bwd(3x)": 3 * s * FLOP / dt / 1e12}))
if "c5" in which:          # config 5: training steps, 4096 rays x {64,128}, bf16
    from nerf_simple_amd.optim import FusedAdam
    for N, fused_opt in ((64, True), (128, True), (64, False)):
        net = net_of("bf16")
        opt = FusedAdam(net, lr=5e-4) if fused_opt else torch.optim.Adam(net.parameters(), lr=5e-4)
        rays = generate_rays(pose, [800, 800, synthetic.focal_from_fov(800)], dev)[:4096].contiguous()
        gt = torch.rand(4096, 3, device=dev)
        dt = timed(lambda: train_step(net, opt, rays, gt, N, device_rng=True, seed=1), warm=3, reps=10)
        s = 4096 * N
        print(json.dumps({"config": f"5: train step 4096 rays x {N} bf16 (fwd+bwd+" +
                          ("FusedAdam+repack)" if fused_opt else "torch Adam)"), "ms": dt * 1e3,
                          "ray_samples_per_s": s / dt, "tflops_fwd_bwd(3x)": 3 * s * FLOP / dt / 1e12}))
