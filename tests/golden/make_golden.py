#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  It holds no reference code: it imports the reference's
``utils`` package by path with two process-local shims:
  * an empty ``cv2`` module (cv2 is imported but unused on the hot path),
  * ``torch.Tensor.cuda`` -> identity (``.cuda()`` is hard-coded at
    utils/rendering.py:30,68; there is no GPU here),
and writes small .npz files of inputs and expected outputs.

Weights are NOT stored: both sides regenerate them from
nerf_simple_amd.utils.synthetic.synthetic_state_dict(seed, kind).

Fixtures:
  G1 encode.npz        positional_encoder / gamma on 256 scene-scale points
  G2 mlp_<kind>.npz    Nerf.forward on 512 points + h5/h8/h9 intermediates
  G3 composite.npz     volume_render: analytic KAT, NaN edge, softplus edges, random
  G4 render_<kind>.npz render_nerf end-to-end, 256 rays x N in {32,64,128,192}
  G5 image_<kind>.npz  config-1 full 100x100 image, N=32 (clipped rgb, disp); image_u.npz = the jitter
  G6 train.npz         64 rays x 64 samples: loss, grads, params after one Adam step
  G7 camera.npz        rays_single_cam / spherical_to_pose / poses_to_render
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.path.insert(0, ROOT)
from nerf_simple_amd.utils import synthetic  # noqa: E402

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
warnings.simplefilter("ignore")
import utils.xyz as rxyz            # noqa: E402
import utils.nets as rnets          # noqa: E402
import utils.rendering as rrend     # noqa: E402


def ref_net(sd):
    net = rnets.Nerf()
    net.load_state_dict(sd, strict=True)
    return net.eval()


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def g1_encode():
    v = synthetic.points_in_scene(256, seed=11)
    with torch.no_grad():
        posx, posd = rxyz.positional_encoder(v)
        g7 = rxyz.gamma(v[:, 0:1], L=7)
    save("encode.npz", v=np_(v), posx=np_(posx), posd=np_(posd), gamma7_x=np_(g7))


def g2_mlp(kind):
    sd = synthetic.synthetic_state_dict(0, kind)
    net = ref_net(sd)
    v = synthetic.points_in_scene(512, seed=12)
    with torch.no_grad():
        x, d = rxyz.positional_encoder(v)
        h5 = net.layers_0(x)
        h = net.skip_conn_layer(torch.cat([h5, x], axis=1))
        h8 = net.layers_1(h)
        h9 = net.layers_2(h8)
        out = net.forward(v)
    # intermediates for the first 128 points only (layer-by-layer bisecting)
    save(f"mlp_{kind}.npz", v=np_(v), out=np_(out),
         h5=np_(h5[:128]), h8=np_(h8[:128]), h9=np_(h9[:128]))


def _vr(raw, ts, dirs):
    with torch.no_grad():
        return [np_(o) for o in rrend.volume_render(raw, ts, dirs)]


def g3_composite():
    out = {}
    # analytic KAT: ts=[2,3,4,5], sigma=0, rgb=(.2,.4,.6), dir=(0,0,1)
    ts = torch.tensor([[2., 3., 4., 5.]])
    raw = torch.tensor([[[.2, .4, .6, 0.]] * 4])
    dirs = torch.tensor([[0., 0., 1.]])
    for k, v in zip(("rgb", "disp", "alpha", "acc", "w"), _vr(raw, ts, dirs)):
        out[f"kat_{k}"] = v
    out.update(kat_raw=np_(raw), kat_ts=np_(ts), kat_dirs=np_(dirs))
    # edge: sigma=-200 everywhere -> rgb 0, disp NaN
    raw = torch.tensor([[[.2, .4, .6, -200.]] * 4])
    for k, v in zip(("rgb", "disp", "alpha", "acc", "w"), _vr(raw, ts, dirs)):
        out[f"nan_{k}"] = v
    out.update(nan_raw=np_(raw))
    # softplus threshold edges
    sig = torch.tensor([19.9, 20.0, 20.1, -19.9, 0.5, 30.0])
    ts6 = torch.linspace(2, 6, 7)[:-1].reshape(1, 6) + 0.01
    raw = torch.cat([torch.full((1, 6, 3), 0.3), sig.reshape(1, 6, 1) * 0.05], dim=-1)
    raw[..., 3] = sig
    for k, v in zip(("rgb", "disp", "alpha", "acc", "w"), _vr(raw, ts6, dirs)):
        out[f"sp_{k}"] = v
    out.update(sp_raw=np_(raw), sp_ts=np_(ts6))
    # random, all N of the configs (+ a non-power-of-two N)
    g = torch.Generator().manual_seed(3)
    for N in (32, 64, 128, 192):
        B = 64
        raw = torch.randn(B, N, 4, generator=g)
        raw[..., 3] = raw[..., 3] * 3.0
        u = torch.rand(B, N, generator=g)
        t_bins = torch.linspace(2, 6, N + 1)
        ts = (t_bins[1] - t_bins[0]) * u + t_bins[:-1]
        d = torch.randn(B, 3, generator=g)
        d = d / torch.norm(d, dim=1, keepdim=True)
        for k, v in zip(("rgb", "disp", "alpha", "acc", "w"), _vr(raw, ts, d)):
            out[f"rnd{N}_{k}"] = v
        out.update({f"rnd{N}_raw": np_(raw), f"rnd{N}_ts": np_(ts), f"rnd{N}_dirs": np_(d)})
    save("composite.npz", **out)


def _cam_rays(H, W, phi=0.0):
    f = synthetic.focal_from_fov(W)
    pose = torch.from_numpy(rxyz.spherical_to_pose(4, -30, phi)).float()
    d = rxyz.rays_single_cam([H, W, f])
    rd = torch.matmul(pose[:3, :3], d)
    o = pose[:3, 3:].expand(3, H * W)
    return torch.cat((o, rd), dim=0).permute(1, 0).reshape(-1, 6), pose, f


def g4_render(kind):
    sd = synthetic.synthetic_state_dict(0, kind)
    net = ref_net(sd)
    rays_img, _, _ = _cam_rays(100, 100)
    idx = torch.randperm(10000, generator=torch.Generator().manual_seed(4))[:256]
    rays = rays_img[idx].contiguous()
    out = {"rays": np_(rays)}
    for N in (32, 64, 128, 192):
        seed = 100 + N
        torch.manual_seed(seed)
        u = torch.rand(256, N)          # what the reference will draw
        torch.manual_seed(seed)
        with torch.no_grad():
            rgb, disp, alpha, acc, w = rrend.render_nerf(rays, net, N)
        out.update({f"N{N}_seed": seed, f"N{N}_u": np_(u), f"N{N}_rgb": np_(rgb),
                    f"N{N}_disp": np_(disp), f"N{N}_alpha": np_(alpha),
                    f"N{N}_acc": np_(acc), f"N{N}_w": np_(w)})
    save(f"render_{kind}.npz", **out)


def g5_image(kind):
    """Config 1: 100x100, N=32, through the reference's render_poses loop body
    (utils/rendering.py:139-151; N is hard-coded to 128 there, so the loop is
    driven here with N=32 via render_nerf exactly as that body does)."""
    sd = synthetic.synthetic_state_dict(0, kind)
    net = ref_net(sd)
    rays, pose, f = _cam_rays(100, 100)
    bs = 2500
    torch.manual_seed(1234)
    us, rgbs, disps = [], [], []
    with torch.no_grad():
        for i in range(rays.size(0) // bs):
            st = torch.get_rng_state()
            us.append(torch.rand(bs, 32))
            torch.set_rng_state(st)
            rgb, disp, _, _, _ = rrend.render_nerf(rays[i * bs:(i + 1) * bs], net, N=32)
            rgbs.append(torch.clip(rgb, torch.tensor(0.), torch.tensor(1.)))
            disps.append(disp)
    # the jitter is the same for both weight sets: stored once
    if kind == "default":
        save("image_u.npz", u=np_(torch.cat(us)), seed=1234, batch_size=bs)
    save(f"image_{kind}.npz", pose=np_(pose), f=np.float64(f), batch_size=bs,
         rgb=np_(torch.cat(rgbs)), disp=np_(torch.cat(disps)))


def g6_train():
    sd = synthetic.synthetic_state_dict(0, "default")
    net = rnets.Nerf()
    net.load_state_dict(sd, strict=True)
    rays_img, _, _ = _cam_rays(100, 100)
    idx = torch.randperm(10000, generator=torch.Generator().manual_seed(6))[:64]
    rays = rays_img[idx].contiguous()
    gt = torch.rand(64, 3, generator=torch.Generator().manual_seed(7))
    N = 64
    torch.manual_seed(66)
    u = torch.rand(64, N)
    torch.manual_seed(66)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    opt.zero_grad()
    rgb, _, _, _, _ = rrend.render_nerf(rays, net, N)
    loss = torch.nn.MSELoss()(rgb, gt)
    loss.backward()
    out = {"rays": np_(rays), "gt": np_(gt), "u": np_(u), "N": N,
           "loss": np_(loss), "rgb": np_(rgb)}
    for k, p in net.named_parameters():
        g = p.grad
        out[f"gnorm/{k}"] = np_(g.norm())
        if g.numel() <= 4096:
            out[f"grad/{k}"] = np_(g)
        else:
            out[f"gradc/{k}"] = np_(g[:16, :16])
    opt.step()
    for k, p in net.named_parameters():
        if p.numel() <= 4096:
            out[f"post/{k}"] = np_(p)
        else:
            out[f"postc/{k}"] = np_(p[:16, :16])
    save("train.npz", **out)


def g7_camera():
    f = synthetic.focal_from_fov(100)
    d = rxyz.rays_single_cam([100, 100, f])
    d2 = rxyz.rays_single_cam([6, 10, 7.5])
    pose = rxyz.spherical_to_pose(4, -30, 40)
    poses = torch.stack(rxyz.poses_to_render(4, -30, 5))
    rays, _, _ = _cam_rays(100, 100, phi=40.0)
    save("camera.npz", f=np.float64(f), dirs100=np_(d), dirs_6x10=np_(d2),
         pose_4_m30_40=pose, poses5=np_(poses), rays100_phi40=np_(rays))


if __name__ == "__main__":
    torch.set_num_threads(8)
    g1_encode()
    g3_composite()
    g7_camera()
    for kind in ("default", "structured"):
        g2_mlp(kind)
        g4_render(kind)
        g5_image(kind)
    g6_train()
