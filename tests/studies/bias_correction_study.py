#!/usr/bin/env python3
"""CPU study (test infrastructure, uses the oracle): can a pack-time bias correction win bf16 operands back?

The PSNR loss of bf16 operands is the SYSTEMATIC part of the weight rounding.  dW = W - bf16(W)
is known at pack time; with the mean input activation of each layer, E[x_l], from one fp32 calibration forward, the
mean shift of every pre-activation can be folded into the layer's fp32 bias:  b' = b + dW . E[x_l].

Emulates the kernel's numerics as tests/studies/precision_study.py does (operands rounded to bf16, fp32 accumulate,
fp32 bias) and reports |PSNR(x,T) - PSNR(CPU,T)| on the three views of tests/test_gpu_parity.py's criterion test,
for several calibration sets.  Kill criterion of the experiment: 0.05 dB on all three views.

    python tests/studies/bias_correction_study.py
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O                                   # noqa: E402
from nerf_simple_amd.utils import synthetic               # noqa: E402
import numpy as np                                        # noqa: E402

LINEARS = ["layers_0.0", "layers_0.2", "layers_0.4", "layers_0.6", "layers_0.8", "skip_conn_layer.0", "layers_1.0",
           "layers_1.2", "sigma_fc.0", "layers_2", "color_fc.0", "color_fc.2"]


def bf(x):
    return x.to(torch.bfloat16).float()


def layer_inputs(sd, v):
    """fp32 forward; returns the input of each of the 12 linears."""
    x, d = O.positional_encoder(v)
    ins = {}
    h = x
    for i in (0, 2, 4, 6, 8):
        ins[f"layers_0.{i}"] = h
        h = F.relu(F.linear(h, sd[f"layers_0.{i}.weight"], sd[f"layers_0.{i}.bias"]))
    hx = torch.cat([h, x], 1)
    ins["skip_conn_layer.0"] = hx
    h = F.relu(F.linear(hx, sd["skip_conn_layer.0.weight"], sd["skip_conn_layer.0.bias"]))
    for i in (0, 2):
        ins[f"layers_1.{i}"] = h
        h = F.relu(F.linear(h, sd[f"layers_1.{i}.weight"], sd[f"layers_1.{i}.bias"]))
    ins["sigma_fc.0"] = h
    ins["layers_2"] = h
    h9 = F.linear(h, sd["layers_2.weight"], sd["layers_2.bias"])
    hd = torch.cat([h9, d], 1)
    ins["color_fc.0"] = hd
    c = F.relu(F.linear(hd, sd["color_fc.0.weight"], sd["color_fc.0.bias"]))
    ins["color_fc.2"] = c
    return ins


def corrected(sd, calib_pts, weights=None):
    """State dict whose biases carry dW . E[x] (the weights themselves unchanged: the kernel rounds them)."""
    out = {k: v.clone() for k, v in sd.items()}
    ins = layer_inputs(sd, calib_pts)
    for name in LINEARS:
        w = sd[name + ".weight"]
        dw = w - bf(w)
        x = ins[name]
        mean = x.mean(0) if weights is None else (x * weights[:, None]).sum(0) / weights.sum()
        out[name + ".bias"] = sd[name + ".bias"] + dw @ mean
    return out


def forward_bf16(sd, v):
    """All operands bf16, fp32 accumulate, fp32 bias (precision_study.py 'all bf16')."""
    def lin(x, name):
        return F.linear(bf(x), bf(sd[name + ".weight"]), sd[name + ".bias"])
    x, d = O.positional_encoder(v)
    h = x
    for i in (0, 2, 4, 6, 8):
        h = F.relu(lin(h, f"layers_0.{i}"))
    h = F.relu(lin(torch.cat([h, x], 1), "skip_conn_layer.0"))
    for i in (0, 2):
        h = F.relu(lin(h, f"layers_1.{i}"))
    sigma = lin(h, "sigma_fc.0")
    h9 = lin(h, "layers_2")
    c = F.relu(lin(torch.cat([h9, d], 1), "color_fc.0"))
    return torch.cat([lin(c, "color_fc.2"), sigma], 1)


def render(sd, rays, u, fwd, N=32):
    ts = O.sample_ts(u)
    q, dn = O.query_points(rays, ts)
    out = fwd(sd, q).reshape(rays.shape[0], N, 4)
    rgb, _, _, _, w = O.volume_render(out, ts, dn)
    return torch.clip(rgb, 0, 1), q, w.reshape(-1)


def main():
    torch.set_num_threads(8)
    u = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "image_u.npz"))["u"])
    f = synthetic.focal_from_fov(100)
    for kind in ("structured", "default"):
        sd = synthetic.synthetic_state_dict(0, kind)
        teacher = synthetic.perturbed_state_dict(sd, seed=1, rel=0.02)
        views = {}
        with torch.no_grad():
            for phi in (0.0, 120.0, 240.0):
                pose = torch.from_numpy(O.spherical_to_pose(4, -30, phi)).float()
                rays = O.camera_rays(pose, [100, 100, f])
                T, _ = O.render_image(teacher, rays, 2500, N=32, u=u)
                cpu, q, w = render(sd, rays, u, O.nerf_forward)
                views[phi] = (rays, T, float(O.img_psnr(T, cpu)), q, w)
            # calibration sets
            cube = synthetic.points_in_scene(8192, seed=5)                       # uniform in the scene cube
            pose = torch.from_numpy(O.spherical_to_pose(4, -30, 60.0)).float()   # a view that is not evaluated
            r60 = O.camera_rays(pose, [64, 64, f])
            _, q60, w60 = render(sd, r60, torch.rand(r60.shape[0], 32, generator=torch.Generator().manual_seed(9)), O.nerf_forward)
            sets = {"none (plain bf16)": None,
                    "8192 points uniform in the cube": corrected(sd, cube),
                    "samples of a 64x64 view at phi 60": corrected(sd, q60),
                    "the same, weighted by compositing weight w": corrected(sd, q60, w60 + 1e-6),
                    "ORACLE calibration: the evaluated view's own samples": "own"}
            print(f"== {kind}")
            for name, csd in sets.items():
                ds = []
                for phi, (rays, T, p_cpu, q, w) in views.items():
                    use = sd if csd is None else (corrected(sd, q) if csd == "own" else csd)
                    img, _, _ = render(use, rays, u, forward_bf16)
                    ds.append(float(O.img_psnr(T, img)) - p_cpu)
                print(f"  {name:55s} dPSNR " + "  ".join(f"{d:+.4f}" for d in ds) + f"   worst {max(abs(d) for d in ds):.4f} dB", flush=True)


if __name__ == "__main__":
    main()
