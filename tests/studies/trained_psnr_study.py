#!/usr/bin/env python3
"""GPU study: BASELINE's PSNR criterion |PSNR(GPU,T) - PSNR(CPU,T)| on weights that come out of TRAINING, as a
function of how well the model fits its target (the synthetic protocol of the tests renders a 2 % weight perturbation
of the same network, i.e. a model 25-30 dB from its target).

Trains the default-initialised network on the two-view dataset of tests/golden/dataset.npz with the graphed bf16 step,
stops at several iteration counts, and renders training view 0 (the best-fitted pixels) and a held-out view with the fp16
and bf16 kernels and with the CPU oracle in fp32 from the same state dict; T = the teacher's fp32 render.

    python tests/studies/trained_psnr_study.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nerf_oracle as O                                    # noqa: E402
from nerf_simple_amd.utils import synthetic                # noqa: E402
from nerf_simple_amd.utils.nets import Nerf                # noqa: E402
from nerf_simple_amd.optim import FusedAdam                # noqa: E402
from nerf_simple_amd.training import GraphedTrainStep, lr_decay_factor   # noqa: E402
from nerf_simple_amd.utils.rendering import render_nerf    # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(min(16, os.cpu_count() or 1))
d = np.load(os.path.join(ROOT, "tests", "golden", "dataset.npz"))
hw = int(d["hw"])
f = synthetic.focal_from_fov(hw)
rays_tab = torch.cat([O.camera_rays(torch.from_numpy(O.spherical_to_pose(4, -30, float(phi))).float(), [hw, hw, f])
                      for phi in d["views"]]).contiguous()
gt_tab = torch.from_numpy(d["gt"])
teacher = synthetic.synthetic_state_dict(0, "structured")
B, N = 1024, 128
STOPS = (500, 2000, 8000, 20000)
net = Nerf().to(dev)
net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
opt = FusedAdam(net, lr=5e-4)
stepper = GraphedTrainStep(net, opt, B, N)
decay = lr_decay_factor(5e-4, 5e-5, STOPS[-1])
gen = torch.Generator().manual_seed(5)
rays_dev, gt_dev = rays_tab.to(dev), gt_tab.to(dev)
views = {"training view 0": rays_tab[:hw * hw],
         "held-out view (azimuth 20)": O.camera_rays(torch.from_numpy(O.spherical_to_pose(4, -30, 20.0)).float(), [hw, hw, f]).contiguous()}
u_eval = torch.rand(hw * hw, N, generator=torch.Generator().manual_seed(9))
with torch.no_grad():
    T = {k: torch.clip(O.render_nerf(v, teacher, N, u=u_eval)[0], 0, 1) for k, v in views.items()}
# the training view is judged against the pixels the model was trained on (the dataset's own render of the teacher):
# a re-render of the high-gain teacher with other jitter differs from them by 22 dB on its own
T["training view 0"] = gt_tab[:hw * hw]
it = 0
for stop in STOPS:
    while it < stop:
        ids = torch.randperm(rays_tab.shape[0], generator=gen)[:B].to(dev)
        u = torch.rand(B, N, generator=gen).to(dev)
        loss = stepper.step(rays_dev[ids], gt_dev[ids], u=u, decay=decay)
        it += 1
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    print(f"== after {it} iterations, training loss {float(loss):.6f}")
    with torch.no_grad():
        for name, rays in views.items():
            cpu = torch.clip(O.render_nerf(rays, sd, N, u=u_eval)[0], 0, 1)
            p_cpu = float(O.img_psnr(T[name], cpu))
            out = []
            for prec in ("fp16", "bf16"):
                img = torch.clip(render_nerf(rays.to(dev), net, N, u=u_eval.to(dev), precision=prec)[0], 0, 1).cpu()
                out.append(f"{prec} {float(O.img_psnr(T[name], img)) - p_cpu:+.4f} dB (PSNR(GPU,CPU) {float(O.img_psnr(cpu, img)):.1f})")
            print(f"   {name:28s} PSNR(CPU,T) {p_cpu:6.2f} dB   delta " + "   ".join(out), flush=True)
