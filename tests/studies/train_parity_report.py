#!/usr/bin/env python3
"""Report behind the training-parity bounds of tests/test_gpu_trajectory.py (run on the GPU box):

    python tests/studies/train_parity_report.py [g6b] [g6c] [g8] [--repeat R]

  g6b / g6c  per parameter tensor: relative L2 error e_k of the fused bf16 gradient against the CPU oracle's fp32
             autograd, the reference's own minibatch noise s_k (fixture ``mbstd``), and e_k / s_k
  g8         the 60-iteration trajectory, eager and graphed: loss per step against the reference run of the same
             seed, the reference's spread between seeds, parameter displacement errors, validation MSE
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nerf_oracle as O                                    # noqa: E402
from nerf_simple_amd.utils import synthetic                # noqa: E402
from nerf_simple_amd.utils.nets import Nerf                # noqa: E402
from nerf_simple_amd.optim import FusedAdam                # noqa: E402
from nerf_simple_amd.training import train_step, GraphedTrainStep   # noqa: E402
from nerf_simple_amd.utils.rendering import render_nerf    # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_l2(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def dataset_tables():
    d = np.load(os.path.join(GOLDEN, "dataset.npz"))
    hw = int(d["hw"])
    rays = torch.cat([O.camera_rays(torch.from_numpy(O.spherical_to_pose(4, -30, float(phi))).float(),
                                    [hw, hw, synthetic.focal_from_fov(hw)]) for phi in d["views"]]).contiguous()
    return rays, t(d["gt"])


def new_net():
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    return net


def one_step_report(tag, rays, gt, u, N, mbstd=None):
    sd = synthetic.synthetic_state_dict(0, "default")
    t0 = time.time()
    want_loss, want = O.train_step_grads(sd, rays, u, gt, N)
    print(f"== {tag}: {rays.shape[0]} rays x {N}; oracle fp32 autograd {time.time() - t0:.1f} s, loss {float(want_loss):.7f}")
    for graphed in (False, True):
        net = new_net()
        opt = FusedAdam(net, lr=5e-4)
        if graphed:
            st = GraphedTrainStep(net, opt, rays.shape[0], N)
            loss = float(st.step(rays.to(dev), gt.to(dev), u=u.to(dev)))
        else:
            loss = float(train_step(net, opt, rays.to(dev), gt.to(dev), N, u=u.to(dev)))
        grads = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}
        print(f"  {'graphed' if graphed else 'eager  '}: loss {loss:.7f} (rel {abs(loss - float(want_loss)) / float(want_loss):.2e})")
        worst = 0.0
        for k in want:
            e = rel_l2(grads[k].numpy(), want[k].numpy())
            cos = float((grads[k].double().flatten() @ want[k].double().flatten()) /
                        (grads[k].double().norm() * want[k].double().norm()))
            nr = float(grads[k].double().norm() / want[k].double().norm())
            line = f"     {k:28s} e {e:.3e}  1-cos {1 - cos:.2e}  norm ratio {nr:.5f}"
            if mbstd is not None:
                s = float(mbstd[f"mbstd/{k}"])
                line += f"  s {s:.3e}  e/s {e / s:.3f}"
                worst = max(worst, e / s)
            print(line)
        allg = torch.cat([grads[k].reshape(-1) for k in want]).numpy()
        allw = torch.cat([want[k].reshape(-1) for k in want]).numpy()
        print(f"     all entries as one vector: e {rel_l2(allg, allw):.3e}" + (f"; worst e/s {worst:.3f}" if mbstd is not None else ""))


def g6b():
    g = np.load(os.path.join(GOLDEN, "train_n128.npz"))
    one_step_report("G6b", t(g["rays"]), t(g["gt"]), t(g["u"]), int(g["N"]))
    g = np.load(os.path.join(GOLDEN, "train.npz"))
    one_step_report("G6", t(g["rays"]), t(g["gt"]), t(g["u"]), int(g["N"]))


def g6c():
    g = np.load(os.path.join(GOLDEN, "train_cfg.npz"))
    rays_tab, gt_tab = dataset_tables()
    B, N = int(g["B"]), int(g["N"])
    torch.manual_seed(int(g["seed"]))
    ids = torch.randperm(rays_tab.size(0))[:B]
    u = torch.rand(B, N)
    one_step_report("G6c", rays_tab[ids], gt_tab[ids], u, N, mbstd=g)
    # the same rays at the first G8 batch size, against G8's minibatch noise at 256 rays
    g8 = np.load(os.path.join(GOLDEN, "trajectory.npz"))
    torch.manual_seed(int(g8["seeds"][0]))
    ids = torch.randperm(rays_tab.size(0))[:int(g8["B"])]
    u = torch.rand(int(g8["B"]), int(g8["N"]))
    one_step_report("G8 step 0", rays_tab[ids], gt_tab[ids], u, int(g8["N"]), mbstd=g8)


def g8(repeat=1):
    g = np.load(os.path.join(GOLDEN, "trajectory.npz"))
    rays_tab, gt_tab = dataset_tables()
    B, N, K = int(g["B"]), int(g["N"]), int(g["K"])
    decay = float(g["decay"])
    seeds = [int(s) for s in g["seeds"]]
    ckpts = [int(c) for c in g["checkpoints"]]
    val_rays, val_gt = rays_tab[::int(g["val_stride"])].contiguous().to(dev), gt_tab[::int(g["val_stride"])].contiguous().to(dev)
    torch.manual_seed(int(g["val_seed"]))
    u_val = torch.rand(val_rays.shape[0], N).to(dev)
    sd0 = synthetic.synthetic_state_dict(0, "default")
    ref = {s: g[f"loss/{s}"] for s in seeds}
    refv = {s: g[f"val/{s}"] for s in seeds}
    spread = np.std(np.stack([refv[s] for s in seeds]), axis=0, ddof=1)
    print(f"== G8: {K} iterations of {B} rays x {N}; reference val MSE per seed at steps 0,{ckpts}:")
    for s in seeds:
        print(f"     seed {s}: {refv[s]}")
    print(f"     sample std between seeds {spread}  (relative {spread / np.mean(np.stack([refv[s] for s in seeds]), axis=0)})")

    def val_mse(net):
        with torch.no_grad():
            rgb = render_nerf(val_rays, net, N, u=u_val, precision="fp32")[0]
        return float(torch.mean((rgb - val_gt) ** 2))

    for rep in range(repeat):
        for mode in ("eager", "graphed"):
            for seed in seeds[:2] if rep == 0 else seeds[:1]:
                net = new_net()
                opt = FusedAdam(net, lr=5e-4)
                stepper = GraphedTrainStep(net, opt, B, N) if mode == "graphed" else None
                losses, vals, snaps = [], [val_mse(net)], {}
                torch.manual_seed(seed)
                for i in range(K):
                    ids = torch.randperm(rays_tab.size(0))[:B]
                    rays, gt = rays_tab[ids].to(dev), gt_tab[ids].to(dev)
                    if stepper is not None:
                        loss = stepper.step(rays, gt, decay=decay)
                    else:
                        loss = train_step(net, opt, rays, gt, N, decay=decay)
                    losses.append(float(loss))
                    if i + 1 in ckpts:
                        vals.append(val_mse(net))
                        snaps[i + 1] = {k: p.detach().float().cpu().clone() for k, p in net.named_parameters()}
                nxt = torch.rand(4)
                losses = np.asarray(losses)
                rl = np.abs(losses - ref[seed]) / ref[seed]
                print(f"  {mode} seed {seed} rep {rep}: loss rel dev max {rl.max():.3e} (step {rl.argmax()}), mean {rl.mean():.3e}; "
                      f"first 3 {rl[:3]}, last 3 {rl[-3:]}")
                print(f"     val MSE {vals}  vs ref {refv[seed]}  rel {[abs(a - b) / b for a, b in zip(vals, refv[seed])]}")
                print(f"     in dB: {[10 * np.log10(a / b) for a, b in zip(vals, refv[seed])]}")
                if seed == seeds[0]:
                    print(f"     rng stream position after the run matches the reference: {np.array_equal(nxt.numpy(), g['rng_next'])}")
                    for step in ckpts:
                        num = den = 0.0
                        worst = (0.0, "")
                        for k, p in snaps[step].items():
                            if f"step{step}/{k}" in g.files:
                                want, got, p0 = g[f"step{step}/{k}"], p.numpy(), sd0[k].numpy()
                            else:
                                want, got, p0 = g[f"step{step}c/{k}"], p.numpy()[:16, :16], sd0[k].numpy()[:16, :16]
                            dn, dd = np.linalg.norm(got - want), np.linalg.norm(want - p0)
                            num += dn ** 2
                            den += dd ** 2
                            if dn / dd > worst[0]:
                                worst = (dn / dd, k)
                        print(f"     step {step}: parameter error / displacement from init: all stored slices {np.sqrt(num / den):.3e}, "
                              f"worst tensor {worst[0]:.3e} ({worst[1]})")
    # the reference's own displacement spread between seeds is not in the fixture; the loss-curve spread is:
    allr = np.stack([ref[s] for s in seeds])
    print(f"  reference loss curves: relative std between seeds, mean over steps {np.mean(np.std(allr, axis=0, ddof=1) / allr.mean(0)):.3e} "
          f"(different batches per seed)")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rep = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 1
    args = [a for a in args if not a.isdigit()]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    if not args or "g6b" in args:
        g6b()
    if not args or "g6c" in args:
        g6c()
    if not args or "g8" in args:
        g8(rep)
