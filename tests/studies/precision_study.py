#!/usr/bin/env python3
"""CPU study (test infrastructure, uses the oracle): which layers of the fused MLP need more than
bf16's 8-bit mantissa for |PSNR(GPU,T) - PSNR(CPU,T)| <= 0.05 dB on the structured weights.

Emulates the kernel's numerics with torch-CPU ops: per internal layer a dtype for (weights, input
activations) -- 'bf16', 'fp16', 'fp32', or 'bf16x2' (hi+lo split of both operands, 3 products) --
fp32 accumulation and fp32 bias, encoder outputs rounded like the kernel stages them.

    python tests/studies/precision_study.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O                                   # noqa: E402
from nerf_simple_amd.utils import synthetic               # noqa: E402

LAYERS = ["L0", "L1", "L2", "L3", "L4", "L5", "L6", "L7", "L8", "SIG", "L9", "L10"]


def rnd(x, kind):
    if kind == "bf16":
        return x.to(torch.bfloat16).float()
    if kind == "fp16":
        return x.to(torch.float16).float()
    return x


def lin(x, w, b, kind):
    """x @ w.T + b with both operands rounded to `kind`, fp32 accumulate."""
    if kind == "bf16x2":
        xh = rnd(x, "bf16"); xl = rnd(x - xh, "bf16")
        wh = rnd(w, "bf16"); wl = rnd(w - wh, "bf16")
        return F.linear(xh, wh) + F.linear(xl, wh) + F.linear(xh, wl) + b
    if kind == "bf16w2":     # weights split only (activation stays bf16)
        xh = rnd(x, "bf16")
        wh = rnd(w, "bf16"); wl = rnd(w - wh, "bf16")
        return F.linear(xh, wh) + F.linear(xh, wl) + b
    return F.linear(rnd(x, kind), rnd(w, kind), b)


def forward(sd, v, cfg):
    x, d = O.positional_encoder(v)
    h = x
    for n, i in zip(("L0", "L1", "L2", "L3", "L4"), (0, 2, 4, 6, 8)):
        h = F.relu(lin(h, sd[f"layers_0.{i}.weight"], sd[f"layers_0.{i}.bias"], cfg[n]))
    h = F.relu(lin(torch.cat([h, x], 1), sd["skip_conn_layer.0.weight"], sd["skip_conn_layer.0.bias"], cfg["L5"]))
    for n, i in zip(("L6", "L7"), (0, 2)):
        h = F.relu(lin(h, sd[f"layers_1.{i}.weight"], sd[f"layers_1.{i}.bias"], cfg[n]))
    sigma = lin(h, sd["sigma_fc.0.weight"], sd["sigma_fc.0.bias"], cfg["SIG"])
    h9 = lin(h, sd["layers_2.weight"], sd["layers_2.bias"], cfg["L8"])
    c = F.relu(lin(torch.cat([h9, d], 1), sd["color_fc.0.weight"], sd["color_fc.0.bias"], cfg["L9"]))
    rgb = lin(c, sd["color_fc.2.weight"], sd["color_fc.2.bias"], cfg["L10"])
    return torch.cat([rgb, sigma], 1)


def render(sd, rays, u, cfg, N=32):
    ts = O.sample_ts(u)
    q, dn = O.query_points(rays, ts)
    out = forward(sd, q, cfg).reshape(rays.shape[0], N, 4)
    rgb = O.volume_render(out, ts, dn)[0]
    return torch.clip(rgb, 0, 1)


def main():
    torch.set_num_threads(8)
    kind = sys.argv[1] if len(sys.argv) > 1 else "structured"
    g = np.load(os.path.join(ROOT, "tests", "golden", f"image_{kind}.npz"))
    u = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "image_u.npz"))["u"])
    sd = synthetic.synthetic_state_dict(0, kind)
    f = synthetic.focal_from_fov(100)
    pose = torch.from_numpy(g["pose"])
    rays = O.camera_rays(pose, [100, 100, f])
    teacher = synthetic.perturbed_state_dict(sd, seed=1, rel=0.02)
    with torch.no_grad():
        T, _ = O.render_image(teacher, rays, 2500, N=32, u=u)
        cpu = torch.from_numpy(g["rgb"])
        p_cpu = float(O.img_psnr(T, cpu))
        print(f"PSNR(CPU,T) = {p_cpu:.3f} dB")

        def report(name, cfg):
            img = render(sd, rays, u, cfg)
            p = float(O.img_psnr(T, img))
            print(f"{name:40s} dPSNR {p - p_cpu:+.4f} dB   PSNR(x,CPU) {float(O.img_psnr(cpu, img)):.2f} dB", flush=True)

        base = {k: "bf16" for k in LAYERS}
        report("all fp32 (sanity)", {k: "fp32" for k in LAYERS})
        report("all bf16", base)
        report("all fp16", {k: "fp16" for k in LAYERS})
        for k in LAYERS:                                  # one layer exact at a time
            report(f"bf16, {k} fp32", dict(base, **{k: "fp32"}))
        for k in LAYERS:                                  # one layer bf16, rest exact
            report(f"fp32, {k} bf16", dict({q: "fp32" for q in LAYERS}, **{k: "bf16"}))
        # cumulative from the tail
        for alt in ("fp16", "bf16x2", "bf16w2"):
            for tail in (["SIG"], ["SIG", "L8"], ["SIG", "L8", "L9", "L10"], ["SIG", "L7", "L8", "L9", "L10"],
                         ["SIG", "L6", "L7", "L8", "L9", "L10"], ["SIG", "L5", "L6", "L7", "L8", "L9", "L10"]):
                report(f"bf16, {'+'.join(tail)} {alt}", dict(base, **{k: alt for k in tail}))


if __name__ == "__main__":
    main()
