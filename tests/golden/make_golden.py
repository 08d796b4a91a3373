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
  G6b train_n128.npz   the same capture at the reference's own sample count Nf = 128 (configs/lego.yaml:6)
  G6c train_cfg.npz    the reference's real step shape: 4096 rays x 128 samples (configs/lego.yaml:6,12) drawn from
                       the G8 dataset, plus the reference's own minibatch gradient noise (4 independent batches)
  G7 camera.npz        rays_single_cam / spherical_to_pose / poses_to_render
  G8 trajectory.npz    the training loop body of train.py:45-57 run for 60 iterations (randperm ray selection,
                       render_nerf at Nf = 128, MSELoss, Adam, lr *= decay) on a synthetic two-view dataset: loss per
                       step, parameters at steps 1 / 10 / 60, validation MSE; the same for three more seeds (the
                       reference's own run-to-run spread); dataset.npz = the dataset's target colours
  G9 sizes.npz         the reference's Nerf(Lp, Ld, H) at three other sizes (its constructor takes any): forward and
                       backward on 300 points; the initial weights are rebuilt from the seed (checksums stored)

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py g6c g8     # only the named fixtures
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
    return t.detach().cpu().numpy().copy()          # a copy: .numpy() alone aliases live parameters (G8 stores mid-run)


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


def _store_grads(out, net):
    for k, p in net.named_parameters():
        g = p.grad
        out[f"gnorm/{k}"] = np_(g.norm())
        if g.numel() <= 4096:
            out[f"grad/{k}"] = np_(g)
        else:
            out[f"gradc/{k}"] = np_(g[:16, :16])


def _store_params(out, net, tag="post"):
    for k, p in net.named_parameters():
        if p.numel() <= 4096:
            out[f"{tag}/{k}"] = np_(p)
        else:
            out[f"{tag}c/{k}"] = np_(p[:16, :16])


def g6b_train_n128():
    """G6 at Nf = 128, the sample count the reference trains with (train.py:51, configs/lego.yaml:6)."""
    sd = synthetic.synthetic_state_dict(0, "default")
    net = rnets.Nerf()
    net.load_state_dict(sd, strict=True)
    rays_img, _, _ = _cam_rays(100, 100)
    idx = torch.randperm(10000, generator=torch.Generator().manual_seed(16))[:64]
    rays = rays_img[idx].contiguous()
    gt = torch.rand(64, 3, generator=torch.Generator().manual_seed(17))
    N = 128
    torch.manual_seed(166)
    u = torch.rand(64, N)
    torch.manual_seed(166)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    opt.zero_grad()
    rgb, _, _, _, _ = rrend.render_nerf(rays, net, N)
    loss = torch.nn.MSELoss()(rgb, gt)
    loss.backward()
    out = {"rays": np_(rays), "gt": np_(gt), "u": np_(u), "N": N, "loss": np_(loss), "rgb": np_(rgb)}
    _store_grads(out, net)
    opt.step()
    _store_params(out, net)
    save("train_n128.npz", **out)


# ---- the synthetic dataset of G6c / G8 -------------------------------------------------------------
DATASET_VIEWS = (0.0, 40.0)         # azimuths of the two 64x64 training views
DATASET_HW = 64
DATASET_SEED = 800


def _dataset():
    """Ray table [8192,6] of two 64x64 views (as utils/dataload.py:114-129 builds one per split) and its target
    colours: the reference's own render of the 'structured' weights (the teacher), N = 128, clipped like
    render_poses clips (utils/rendering.py:146)."""
    teacher = ref_net(synthetic.synthetic_state_dict(0, "structured"))
    rays = torch.cat([_cam_rays(DATASET_HW, DATASET_HW, phi)[0] for phi in DATASET_VIEWS]).contiguous()
    torch.manual_seed(DATASET_SEED)
    with torch.no_grad():
        gts = []
        for v in range(len(DATASET_VIEWS)):
            n = DATASET_HW * DATASET_HW
            rgb = rrend.render_nerf(rays[v * n:(v + 1) * n], teacher, 128)[0]
            gts.append(torch.clip(rgb, torch.tensor(0.), torch.tensor(1.)))
    return rays, torch.cat(gts)


def _minibatch_std(grad_sets):
    """Relative sampling deviation of the reference's minibatch gradient, per tensor: sample standard deviation of
    the M batch gradients around their mean, over the norm of the mean."""
    out = {}
    M = len(grad_sets)
    for k in grad_sets[0]:
        stack = torch.stack([g[k] for g in grad_sets]).double()
        mean = stack.mean(0)
        dev = ((stack - mean) ** 2).sum() / (M - 1)
        out[k] = float(torch.sqrt(dev) / mean.norm())
    return out


def g6c_train_config(rays_tab, gt_tab):
    """One step at the reference's real shape (batch_size 4096, Nf 128: configs/lego.yaml:6,12) on the G8
    dataset, ray selection as RayGenerator.select does it (utils/dataload.py:150-153); then the same step on
    three more independent selections: the spread of the four gradients is the reference's own minibatch noise."""
    sd = synthetic.synthetic_state_dict(0, "default")
    B, N = 4096, 128
    out, grad_sets = {}, []
    for j, seed in enumerate((300, 301, 302, 303)):
        net = rnets.Nerf()
        net.load_state_dict(sd, strict=True)
        opt = torch.optim.Adam(net.parameters(), lr=5e-4)
        torch.manual_seed(seed)
        ray_ids = torch.randperm(rays_tab.size(0))[:B]
        rays, gt = rays_tab[ray_ids, :], gt_tab[ray_ids, :]
        opt.zero_grad()
        rgb, _, _, _, _ = rrend.render_nerf(rays, net, N)
        loss = torch.nn.MSELoss()(rgb, gt)
        loss.backward()
        grad_sets.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})
        if j == 0:
            out.update({"seed": seed, "B": B, "N": N, "ray_ids": np_(ray_ids), "loss": np_(loss), "rgb": np_(rgb)})
            _store_grads(out, net)
            opt.step()
            _store_params(out, net)
        out[f"loss_batch{j}"] = np_(loss)
    for k, v in _minibatch_std(grad_sets).items():
        out[f"mbstd/{k}"] = np.float64(v)
    save("train_cfg.npz", **out)


G8_SEEDS = (88, 89, 90, 91)
G8_B, G8_N, G8_K = 256, 128, 60
G8_LR_INIT, G8_LR_FINAL = 5e-4, 1e-4
G8_CHECKPOINTS = (1, 10, 60)
G8_VAL_SEED = 801


def g8_trajectory(rays_tab, gt_tab):
    """train.py:45-57 for K iterations, per seed: ray_ids = randperm(n)[:B] -> render_nerf(rays, net, Nf) ->
    MSELoss -> backward -> Adam.step -> lr *= decay.  torch's CPU generator is seeded once per run; the ray
    selection and the jitter of every step come from that one stream, as in the reference."""
    sd = synthetic.synthetic_state_dict(0, "default")
    decay = np.exp(np.log(G8_LR_FINAL / G8_LR_INIT) / G8_K)           # train.py:36-39
    val_rays, val_gt = rays_tab[::16].contiguous(), gt_tab[::16].contiguous()
    out = {"seeds": np.asarray(G8_SEEDS), "B": G8_B, "N": G8_N, "K": G8_K, "lr_init": G8_LR_INIT,
           "lr_final": G8_LR_FINAL, "decay": np.float64(decay), "checkpoints": np.asarray(G8_CHECKPOINTS),
           "val_seed": G8_VAL_SEED, "val_stride": 16}

    def val_mse(net):
        st = torch.get_rng_state()
        torch.manual_seed(G8_VAL_SEED)                                # u_val = the first torch.rand(512, N) of this seed
        with torch.no_grad():
            rgb = rrend.render_nerf(val_rays, net, G8_N)[0]
        torch.set_rng_state(st)
        return np_(torch.mean((rgb - val_gt) ** 2))

    grad_sets, snaps = [], {}
    for seed in G8_SEEDS:
        net = rnets.Nerf()
        net.load_state_dict(sd, strict=True)
        criterion = torch.nn.MSELoss()
        optimizer = torch.optim.Adam(net.parameters(), lr=5e-4)
        losses, lrs, vals = [], [], [val_mse(net)]
        torch.manual_seed(seed)
        for i in range(G8_K):
            ray_ids = torch.randperm(rays_tab.size(0))[:G8_B]
            rays, gt = rays_tab[ray_ids, :], gt_tab[ray_ids, :]
            optimizer.zero_grad()
            rgb, _, _, _, _ = rrend.render_nerf(rays, net, G8_N)
            loss = criterion(rgb, gt)
            loss.backward()
            if i == 0:
                grad_sets.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})
                if seed == G8_SEEDS[0]:
                    out["ray_ids0"] = np_(ray_ids)
                    _store_grads(out, net)
            lrs.append(optimizer.param_groups[0]["lr"])
            optimizer.step()
            for p in optimizer.param_groups:
                p["lr"] = p["lr"] * decay
            losses.append(np_(loss))
            if i + 1 in G8_CHECKPOINTS:
                vals.append(val_mse(net))
                snaps[(seed, i + 1)] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
                if seed == G8_SEEDS[0]:
                    _store_params(out, net, tag=f"step{i + 1}")
        out[f"loss/{seed}"] = np.asarray(losses, dtype=np.float32)
        out[f"val/{seed}"] = np.asarray(vals, dtype=np.float32)       # at steps 0 and G8_CHECKPOINTS
        if seed == G8_SEEDS[0]:
            out["lr"] = np.asarray(lrs, dtype=np.float64)
            out["rng_next"] = np_(torch.rand(4))                      # the stream position after the run
        print(f"  G8 seed {seed}: loss {float(losses[0]):.5f} -> {float(losses[-1]):.5f}, val {[float(v) for v in vals]}")
    for k, v in _minibatch_std(grad_sets).items():
        out[f"mbstd/{k}"] = np.float64(v)
    # the reference's own run-to-run spread in parameter space: distance between the runs of two seeds over the
    # distance the first one has travelled from the initial weights, all 595,844 parameters
    p0 = torch.cat([v.reshape(-1) for v in sd.values()])
    for c in G8_CHECKPOINTS:
        a = snaps[(G8_SEEDS[0], c)]
        out[f"seedspread/step{c}"] = np.float64(min(float((snaps[(s_, c)] - a).norm() / (a - p0).norm()) for s_ in G8_SEEDS[1:]))
    save("trajectory.npz", **out)


G9_SIZES = ((6, 2, 128), (3, 1, 40), (2, 3, 37))


def g9_sizes():
    """G9: the reference module at sizes other than its default (utils/nets.py:9-32 takes any): Nerf(Lp, Ld, H) built
    under a fixed seed of torch's CPU generator (so the test can rebuild the same initial weights: per-tensor checksums
    are stored, not the weights), heads scaled x4 for a non-trivial output, forward on 300 scene points, backward of a
    fixed upstream gradient: outputs, per-tensor gradient norms, 16 x 16 corners / full biases."""
    out = {"sizes": np.asarray(G9_SIZES)}
    for i, (Lp, Ld, H) in enumerate(G9_SIZES):
        torch.manual_seed(900 + i)
        net = rnets.Nerf(Lp, Ld, H)
        with torch.no_grad():
            net.sigma_fc[0].weight.mul_(4.0)
            net.color_fc[2].weight.mul_(4.0)
        v = synthetic.points_in_scene(300, seed=13 + i)
        g_out = torch.randn(300, 4, generator=torch.Generator().manual_seed(70 + i))
        y = net.forward(v)
        y.backward(g_out)
        tag = f"{Lp}_{Ld}_{H}"
        out[f"{tag}/v"], out[f"{tag}/g_out"], out[f"{tag}/out"] = np_(v), np_(g_out), np_(y)
        for k, p_ in net.named_parameters():
            w = p_.detach()
            out[f"{tag}/init/{k}"] = np.asarray([float(w.double().sum()), float(w.double().abs().sum()), float(w.reshape(-1)[0]),
                                                 float(w.reshape(-1)[-1])])
            g = p_.grad.detach()
            out[f"{tag}/gnorm/{k}"] = np.float64(g.double().norm())
            out[f"{tag}/grad/{k}"] = np_(g if g.dim() == 1 else g[:16, :16])
    save("sizes.npz", **out)


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
    which = set(a.lower() for a in sys.argv[1:])

    def want(name):
        return not which or name in which

    if want("g1"):
        g1_encode()
    if want("g3"):
        g3_composite()
    if want("g7"):
        g7_camera()
    for kind in ("default", "structured"):
        if want("g2"):
            g2_mlp(kind)
        if want("g4"):
            g4_render(kind)
        if want("g5"):
            g5_image(kind)
    if want("g9"):
        g9_sizes()
    if want("g6"):
        g6_train()
    if want("g6b"):
        g6b_train_n128()
    if want("g6c") or want("g8"):
        rays_tab, gt_tab = _dataset()
        save("dataset.npz", gt=np_(gt_tab), views=np.asarray(DATASET_VIEWS), hw=DATASET_HW, seed=DATASET_SEED)
        if want("g6c"):
            g6c_train_config(rays_tab, gt_tab)
        if want("g8"):
            g8_trajectory(rays_tab, gt_tab)
