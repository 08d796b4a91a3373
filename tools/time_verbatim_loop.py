"""Wall time per iteration of the reference's training loop body (train.py:45-57) as written, import root swapped, tables on the
GPU: rg.select -> train_imgs[ray_ids] -> zero_grad -> render_nerf -> MSELoss -> backward -> Adam.step -> lr decay; against the
same iteration as captured hipGraphs (GraphedTrainStep(rays_from=rg), reference RNG stream and counter RNG)."""
import os, sys, time, json
import numpy as np
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_simple_amd.utils import synthetic
from nerf_simple_amd.utils.nets import Nerf
from nerf_simple_amd.utils.rendering import render_nerf
from nerf_simple_amd.optim import FusedAdam
from nerf_simple_amd.training import GraphedTrainStep, train_step

dev = torch.device("cuda:0")
rg = bench.synthetic_ray_table(dev)
train_imgs = rg.colours["train"]
params = {"Nf": int(sys.argv[1]) if len(sys.argv) > 1 else 64, "batch_size": 4096}
decay = np.exp(np.log(5e-5 / 5e-4) / 10000)
out = {"N": params["Nf"]}
import warnings
warnings.simplefilter("ignore")


def timed(fn, iters, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def fresh():
    net = Nerf(precision="bf16").cuda()
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    return net


net = fresh()
criterion = nn.MSELoss()
optimizer = torch.optim.Adam(net.parameters(), lr=5e-4)


def verbatim():
    rays, ray_ids = rg.select(mode='train', N=params['batch_size'])
    gt_colors = train_imgs[ray_ids, :].float().cuda()
    optimizer.zero_grad()
    rgb, depth, alpha, acc, w = render_nerf(rays.cuda(), net, params['Nf'])
    loss = criterion(rgb, gt_colors)
    loss.backward()
    optimizer.step()
    for p in optimizer.param_groups:
        p['lr'] = p['lr'] * decay


out["verbatim_torch_adam_ms"] = timed(verbatim, 300)
net2 = fresh()
opt2 = FusedAdam(net2, lr=5e-4)


def eager_fused():
    rays, gt, _ = rg.select_batch("train", params["batch_size"])
    train_step(net2, opt2, rays, gt, params["Nf"], decay=decay)


out["eager_train_step_fused_adam_ms"] = timed(eager_fused, 300)
net3 = fresh()
g3 = GraphedTrainStep(net3, FusedAdam(net3, lr=5e-4), params["batch_size"], params["Nf"], rays_from=rg)
out["graphed_reference_stream_ms"] = timed(lambda: g3.step(decay=decay), 1000)
net4 = fresh()
g4 = GraphedTrainStep(net4, FusedAdam(net4, lr=5e-4), params["batch_size"], params["Nf"], rays_from=rg, device_rng=True, seed=3)
out["graphed_counter_rng_ms"] = timed(lambda: g4.step(decay=decay), 1000)
# the reference's own first line on this host, for scale: torch.randperm(16 M) on the CPU
t0 = time.perf_counter()
for _ in range(3):
    torch.randperm(16_000_000)
out["host_randperm_16M_ms"] = (time.perf_counter() - t0) / 3 * 1e3
print(json.dumps(out))
