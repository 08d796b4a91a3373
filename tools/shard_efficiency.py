#!/usr/bin/env python3
"""Strong-scaling proxy on ONE GPU: the render kernel on 1/N of the 800x800x128 image (what each of N ranks runs,
bench.py --gpus N), N = 1, 2, 4, 8 -- how much of the per-rank work is fixed cost (launch, ramp, partial last tile).
The exchange (one all-gather of 16 B per ray) is not in it.

    python tools/shard_efficiency.py [--precision fp16]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_simple_amd import _lib, parallel                    # noqa: E402
from nerf_simple_amd.utils import synthetic                   # noqa: E402
from nerf_simple_amd.utils.nets import Nerf                   # noqa: E402
from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16")
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.lib()
code = _lib.precision_code(args.precision)
net = Nerf(precision=args.precision).to(dev)
net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
packed = net.packed_weights(code)
pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
rays_all = camera_rays([pose], [800, 800, synthetic.focal_from_fov(800)]).to(dev).contiguous()
tb = torch.linspace(2, 6, 129).to(dev)
full = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in {0, world - 1}:
        lo, hi = parallel.shard_range(rays_all.shape[0], rank, world)
        rays = rays_all[lo:hi].contiguous()
        px = torch.empty((hi - lo, 4), dtype=torch.float32, device=dev)

        def run():
            _lib.check(lib.nerf_amd_render_pixels_forward(_lib.ptr(rays), None, _lib.ptr(tb), _lib.ptr(packed), code,
                                                          _lib.FLAG_DEVICE_RNG, 1234, lo, _lib.ptr(px), None, hi - lo, 128,
                                                          _lib.stream_ptr(dev)), "render")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 * world
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        worst = max(worst, e0.elapsed_time(e1) / reps)
    full = full or worst
    print(json.dumps({"ranks": world, "rays_per_rank": rays_all.shape[0] // world, "ms_per_rank": round(worst, 4),
                      "compute_scaling_efficiency": round(full / (world * worst), 4)}))
