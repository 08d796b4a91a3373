#!/usr/bin/env python3
"""A/B the bf16 MLP kernel variants in ONE process on ONE device (interleaved
rounds, HIP-event timing of the MLP kernel alone on the 800x800x128 workload).
Usage: python tools/ab_bench.py [rounds]"""
import os
import sys
import statistics

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_simple_amd import _lib                      # noqa: E402
from nerf_simple_amd.utils import synthetic            # noqa: E402
from nerf_simple_amd.utils.nets import Nerf            # noqa: E402
from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["32", "16"]
dev = torch.device("cuda:0")
net = Nerf(precision="bf16").to(dev)
net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
rays = camera_rays([pose], [800, 800, synthetic.focal_from_fov(800)]).float().contiguous().to(dev)   # the C ABI takes contiguous fp32
B, N = rays.shape[0], 128
raw = torch.empty(B, N, 4, device=dev)
ts = torch.empty(B, N, device=dev)
tb = torch.linspace(2, 6, N + 1).to(dev)
packed = net.packed_weights()
lib = _lib.lib()


def run():
    _lib.check(lib.nerf_amd_mlp_forward_rays(_lib.ptr(rays), None, _lib.ptr(tb), _lib.ptr(packed), 1, 2, 1234, 0,
                                             _lib.ptr(raw), _lib.ptr(ts), B, N, _lib.stream_ptr(dev)), "mlp")


times = {v: [] for v in variants}
outs = {}
for v in variants:
    os.environ["NERF_AMD_BF16_TILE"] = v
    run()
    torch.cuda.synchronize()
    outs[v] = raw.clone()
for r in range(rounds):
    for v in variants:
        os.environ["NERF_AMD_BF16_TILE"] = v
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1))
for v in variants:
    t = times[v]
    tf = B * N * 1186816 / (statistics.median(t) * 1e-3) / 1e12
    print(f"tile {v}: median {statistics.median(t):.3f} ms  min {min(t):.3f}  max {max(t):.3f}  -> {tf:.0f} TFLOP/s "
          f"({tf / 25:.1f}% of 2.5 PF)")
if len(variants) == 2:
    a, b = (outs[v] for v in variants)
    print("max |raw_a - raw_b| =", float((a - b).abs().max()), " scale", float(a.abs().max()))
