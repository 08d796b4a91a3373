"""One-process A/B: the graphed training step fed a fixed batch vs selecting a fresh batch from a 16 M-row table inside graph A."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_simple_amd.utils import synthetic
from nerf_simple_amd.utils.nets import Nerf
from nerf_simple_amd.optim import FusedAdam
from nerf_simple_amd.training import GraphedTrainStep

dev = torch.device("cuda:0")
rg = bench.synthetic_ray_table(dev)
out = {}
for N in (64, 128):
    def make(**kw):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        return GraphedTrainStep(net, FusedAdam(net, lr=5e-4), 4096, N, device_rng=True, seed=7, **kw)
    a, b = make(), make(rays_from=rg)
    rays, gt, _ = rg.select_batch("train", 4096, device_rng=True, seed=1)
    res = {"fixed": [], "select": []}
    for rep in range(4):
        for name, fn in (("fixed", lambda: a.step(rays, gt)), ("select", lambda: b.step())):
            res[name].append(bench.event_timed(fn, 1500, 50, dev))
    out[f"N{N}"] = {k: [round(x, 4) for x in v] for k, v in res.items()}
    out[f"N{N}"]["delta_us"] = round(1e3 * (min(res["select"]) - min(res["fixed"])), 2)
print(json.dumps(out))
