#!/usr/bin/env python3
"""Random network sizes and point counts through the layer-by-layer fp32 path (utils/generic_mlp.py on
nerf_amd_linear_f32) against the CPU oracle in float64: forward and every parameter gradient, beyond the handful of sizes the
suite runs every round.  GPU box:  python tests/studies/fuzz_generic.py [cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as oracle                              # noqa: E402
from nerf_simple_amd.utils.nets import Nerf               # noqa: E402

REL = 2e-5                                                # as tests/test_gpu_generic_sizes.py


def oracle_preactivations(sd, v, Lp, Ld):
    """Pre-activations of every ReLU'd layer of the reference forward (utils/nets.py:34-43), by layer name."""
    import torch.nn.functional as F
    x, d = oracle.positional_encoder(v, Lp, Ld)
    pre, h = {}, x
    for i in (0, 2, 4, 6, 8):
        pre[f"layers_0.{i}"] = F.linear(h, sd[f"layers_0.{i}.weight"], sd[f"layers_0.{i}.bias"])
        h = F.relu(pre[f"layers_0.{i}"])
    pre["skip_conn_layer.0"] = F.linear(torch.cat([h, x], 1), sd["skip_conn_layer.0.weight"], sd["skip_conn_layer.0.bias"])
    h = F.relu(pre["skip_conn_layer.0"])
    for i in (0, 2):
        pre[f"layers_1.{i}"] = F.linear(h, sd[f"layers_1.{i}.weight"], sd[f"layers_1.{i}.bias"])
        h = F.relu(pre[f"layers_1.{i}"])
    f = F.linear(h, sd["layers_2.weight"], sd["layers_2.bias"])
    pre["color_fc.0"] = F.linear(torch.cat([f, d], 1), sd["color_fc.0.weight"], sd["color_fc.0.bias"])
    return {k: z.detach() for k, z in pre.items()}
dev = torch.device("cuda:0")
rng = np.random.Generator(np.random.PCG64(11))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for c in range(cases):
    Lp, Ld, H = int(rng.integers(1, 13)), int(rng.integers(1, 7)), int(rng.integers(2, 400))
    P = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 4097, 20011]))
    torch.manual_seed(c)
    net = Nerf(Lp, Ld, H)
    with torch.no_grad():
        net.sigma_fc[0].weight.mul_(4.0)
        net.color_fc[2].weight.mul_(4.0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    g = torch.Generator().manual_seed(100 + c)
    v = torch.cat([torch.rand(P, 3, generator=g) * 6 - 3, torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1)], 1)
    g_out = torch.randn(P, 4, generator=g)
    out = net(v.to(dev))
    out.backward(g_out.to(dev))
    # the checker runs in float64: fp32 on either side can put a pre-activation on the other side of a ReLU's kink
    ref = {k: t.double().clone().requires_grad_(True) for k, t in sd.items()}
    want = oracle.nerf_forward(ref, v.double(), Lp, Ld)
    want.backward(g_out.double())
    e_fwd = float((out.detach().cpu().double() - want.detach()).abs().max()) / max(float(want.detach().abs().max()), 1e-30)
    errs = {k: (p.grad.cpu().double() - ref[k].grad).abs() / max(float(ref[k].grad.abs().max()), 1e-30) for k, p in net.named_parameters()}
    e_grad = max(float(e.max()) for e in errs.values())
    ok = e_fwd <= REL and e_grad <= REL
    note = ""
    if not ok and e_fwd <= REL:
        # A ReLU kink: one unit of one layer has, at one point, a pre-activation within round-off of zero, and its mask bit
        # comes out on the other side.  Its row of that layer's weight / bias gradient then differs, and so does everything
        # below (the point's dX changes for every unit of the earlier layers); the layers above are untouched.  Accept the
        # case if that is the whole picture: the deepest layer with an error has it in ONE row, whose unit's smallest
        # |pre-activation| over the points (float64) is below 1e-6.
        order = ["layers_0.0", "layers_0.2", "layers_0.4", "layers_0.6", "layers_0.8", "skip_conn_layer.0", "layers_1.0",
                 "layers_1.2", "layers_2", "sigma_fc.0", "color_fc.0", "color_fc.2"]
        bad_rows = {}
        for k, e in errs.items():
            where = (e.reshape(e.shape[0], -1).max(dim=1).values > REL).nonzero().flatten().tolist()
            if where:
                bad_rows.setdefault(k.rsplit(".", 1)[0], set()).update(where)
        deepest = max(bad_rows, key=order.index)
        if len(bad_rows[deepest]) == 1 and deepest in ("layers_0.0", "layers_0.2", "layers_0.4", "layers_0.6", "layers_0.8",
                                                         "skip_conn_layer.0", "layers_1.0", "layers_1.2", "color_fc.0"):
            unit = next(iter(bad_rows[deepest]))
            zmin = float(oracle_preactivations(ref, v.double(), Lp, Ld)[deepest][:, unit].abs().min())
            if zmin < 1e-6:
                ok, note = True, f"  [ReLU kink at {deepest} unit {unit}: |z| = {zmin:.1e} at one point; rows below it differ too]"
    bad += not ok
    print(f"{'ok  ' if ok else 'FAIL'} Lp={Lp:2d} Ld={Ld} H={H:3d} P={P:5d}  forward {e_fwd:.1e}  worst gradient {e_grad:.1e}{note}", flush=True)
print("fuzz done, failures:", bad)
