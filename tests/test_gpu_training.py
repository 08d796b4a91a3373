"""GPU parity tests of the training step: the hand-written bf16
training kernels (fused forward saving activations, compositor forward / backward, dX chain, dW
split-K GEMMs, Adam) against golden G6 (captured from the reference's own autograd) and against
the CPU oracle's fp32 autograd on the same (rays, u, gt, weights).

The REQUIREMENT on the training path -- gradients inside half the reference's own minibatch sampling deviation at the
reference's real step shape, a 60-iteration trajectory inside fractions of the reference's own seed-to-seed spread --
lives in tests/test_gpu_trajectory.py (fixtures G6b, G6c, G8).  The bounds in THIS file are
regression guards fitted to the observed bf16 error (<= 3x observed) for shapes that have no fixture of their own:
ragged batches, the structured weights, points mode.

Stated bounds: per parameter tensor, relative L2 = ||g_gpu - g_ref|| / ||g_ref||.  The error is
bf16 rounding of operands, saved activations and activation gradients through 12 layers; it is
noise-like per point, so it averages down with the number of points P in the batch (observed on
MI355X, worst tensor = layers_0.0.weight, the end of the backward chain):
    default weights      P = 36,864: 1.4e-2   P = 4,096 (G6): 3.3e-2   P = 2,405: 4.2e-2   P = 300: 8.0e-2
    structured weights   P = 36,864: 5.9e-2                            P = 2,405: 1.1e-1
REL_L2[kind] below is the bound at P >= 36,864; smaller batches scale it by (36,864 / P)^0.4, which
keeps every bound within 3x of the observed error.  The loss comes from an fp32 compositor on
bf16-MLP outputs: LOSS_RTOL."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REL_L2 = {"default": 3.5e-2, "structured": 1.5e-1}
LOSS_RTOL = {"default": 1e-3, "structured": 2e-2}      # observed 1.6e-4 / 7.6e-3


def rel_l2_bound(kind, P):
    return REL_L2[kind] * max(1.0, 36864.0 / P) ** 0.4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from nerf_simple_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_composite_backward_vs_autograd(dev, oracle):
    from nerf_simple_amd.utils.rendering import volume_render
    gen = torch.Generator().manual_seed(21)
    for B, N in ((5, 64), (3, 128), (4, 192), (2, 33), (2, 300)):
        raw = torch.randn(B, N, 4, generator=gen)
        raw[..., 3] *= 2.0
        ts = torch.sort(torch.rand(B, N, generator=gen) * 4 + 2, dim=1).values
        d = torch.randn(B, 3, generator=gen)
        coef = [torch.randn(s, generator=gen) for s in ((B, 3), (B,), (B, N), (B,), (B, N))]

        def loss_of(outs):
            return sum((c.to(o.device) * o).sum() for c, o in zip(coef, outs))

        r_cpu = raw.clone().requires_grad_(True)
        loss_of(oracle.volume_render(r_cpu, ts, d)).backward()
        r_gpu = raw.to(dev).requires_grad_(True)
        outs = volume_render(r_gpu, ts.to(dev), d.to(dev))
        loss_of(outs).backward()
        want, got = r_cpu.grad.numpy(), r_gpu.grad.cpu().numpy()
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 1e-4 * scale, (B, N, np.abs(got - want).max(), scale)
    # rgb-only upstream gradient (the training case): other g_* are None/zero
    raw = torch.randn(6, 64, 4, generator=gen)
    ts = torch.sort(torch.rand(6, 64, generator=gen) * 4 + 2, dim=1).values
    d = torch.randn(6, 3, generator=gen)
    r_cpu = raw.clone().requires_grad_(True)
    oracle.volume_render(r_cpu, ts, d)[0].pow(2).sum().backward()
    r_gpu = raw.to(dev).requires_grad_(True)
    volume_render(r_gpu, ts.to(dev), d.to(dev))[0].pow(2).sum().backward()
    np.testing.assert_allclose(r_gpu.grad.cpu().numpy(), r_cpu.grad.numpy(), rtol=1e-3, atol=1e-6)


def rel_l2(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def _fused_grads(dev, synthetic, kind, rays, gt, u, N, precision="bf16"):
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.training import train_step
    net = Nerf(precision=precision).to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, kind))
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    loss = train_step(net, opt, rays.to(dev), gt.to(dev), N, u=u.to(dev))
    return float(loss), {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}


def _compare_with_oracle(oracle, synthetic, kind, rays, gt, u, N, loss, grads, tag):
    sd = synthetic.synthetic_state_dict(0, kind)
    want_loss, want = oracle.train_step_grads(sd, rays, u, gt, N)
    assert abs(loss - float(want_loss)) <= LOSS_RTOL[kind] * abs(float(want_loss)), (loss, float(want_loss))
    bound = rel_l2_bound(kind, rays.shape[0] * N)
    worst = {}
    for k in want:
        worst[k] = rel_l2(grads[k].numpy(), want[k].numpy())
    print(f"{tag} {kind}: loss {loss:.6g} vs {float(want_loss):.6g}; rel L2 per tensor: max {max(worst.values()):.3e} "
          f"({max(worst, key=worst.get)})")
    for k, v in worst.items():
        print(f"    {k:28s} {v:.3e}")
    assert max(worst.values()) <= bound, (bound, worst)
    allg = torch.cat([grads[k].reshape(-1) for k in want]).numpy()
    allw = torch.cat([want[k].reshape(-1) for k in want]).numpy()
    print(f"    all 595,844 entries as one vector: {rel_l2(allg, allw):.3e}")
    assert rel_l2(allg, allw) <= 0.5 * bound           # dominated by the large late-layer tensors
    return want


def test_fused_train_compositor_and_pack(dev, synthetic):
    """The training-step conveniences against the entry points they merge: compositing + MSE gradient +
    compositing backward in one launch == volume_render_rays -> mse_loss -> volume_render_rays_backward
    (rgb bit-equal, d_raw to rounding), and the one-launch training re-pack == the two separate packs."""
    from nerf_simple_amd import _lib
    lib = _lib.lib()
    st = _lib.stream_ptr(dev)
    gen = torch.Generator().manual_seed(77)
    for B, N in ((64, 64), (33, 100), (5, 300)):
        raw = torch.randn(B, N, 4, generator=gen).to(dev)
        ts = torch.sort(torch.rand(B, N, generator=gen) * 4 + 2, dim=1).values.to(dev)
        rays = torch.randn(B, 6, generator=gen).to(dev)
        gt = torch.rand(B, 3, generator=gen).to(dev)
        rgb, disp, acc = torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
        loss, g_rgb, d_ref = torch.zeros((), device=dev), torch.empty(B, 3, device=dev), torch.empty(B, N, 4, device=dev)
        ck, ptr = _lib.check, _lib.ptr
        ck(lib.nerf_amd_volume_render_rays(ptr(raw), ptr(ts), ptr(rays), ptr(rgb), ptr(disp), None, ptr(acc), None, B, N, st), "v")
        ck(lib.nerf_amd_mse_loss(ptr(rgb), ptr(gt), ptr(loss), ptr(g_rgb), B * 3, st), "m")
        ck(lib.nerf_amd_volume_render_rays_backward(ptr(raw), ptr(ts), ptr(rays), ptr(g_rgb), None, None, None, None,
                                                    ptr(d_ref), B, N, st), "b")
        rgb2, d2 = torch.empty(B, 3, device=dev), torch.empty(B, N, 4, device=dev)
        ck(lib.nerf_amd_volume_render_mse_backward(ptr(raw), ptr(ts), ptr(rays), ptr(gt), ptr(rgb2), ptr(d2), B, N, st), "f")
        torch.cuda.synchronize()
        assert torch.equal(rgb2, rgb)
        assert float((d2 - d_ref).abs().max()) <= 2e-6 * float(d_ref.abs().max()), (B, N)
        assert abs(float(loss) - float(((rgb - gt) ** 2).mean())) <= 1e-6 * float(loss)
    flat = synthetic.flatten_state_dict(synthetic.synthetic_state_dict(3, "structured")).to(dev)
    a1 = torch.zeros(int(lib.nerf_amd_packed_bytes(_lib.BF16)), dtype=torch.uint8, device=dev)
    b1 = torch.zeros(int(lib.nerf_amd_packed_bytes(_lib.BF16_BWD)), dtype=torch.uint8, device=dev)
    a2, b2 = torch.ones_like(a1), torch.ones_like(b1)
    # the one-launch form re-packs an image that nerf_amd_pack_weights has initialised (it never clears the sticky
    # weight-range word): start from the image of OTHER weights, as a training step does
    other = synthetic.flatten_state_dict(synthetic.synthetic_state_dict(4, "default")).to(dev)
    _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(other), _lib.ptr(a2), _lib.BF16, st), "p")
    _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(a1), _lib.BF16, st), "p")
    _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(b1), _lib.BF16_BWD, st), "p")
    _lib.check(lib.nerf_amd_pack_weights_train(_lib.ptr(flat), _lib.ptr(a2), _lib.ptr(b2), st), "p")
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(b1, b2)


def test_train_step_golden_fused(dev, golden, synthetic, oracle):
    """G6 (64 rays x 64 samples, default weights, MSELoss, Adam(lr=5e-4): reference train.py:51-55)
    through the FUSED bf16 kernels: loss, every gradient the fixture holds, gradient norms and the
    first Adam update, under the stated bf16 bounds; then all 24 full gradients against the oracle."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.training import train_step
    g = golden("train.npz")
    rays, gt, u, N = t(g["rays"]), t(g["gt"]), t(g["u"]), int(g["N"])
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    loss = float(train_step(net, opt, rays.to(dev), gt.to(dev), N, u=u.to(dev)))
    assert abs(loss - float(g["loss"])) <= LOSS_RTOL["default"] * abs(float(g["loss"]))
    for k, p in net.named_parameters():
        grad = p.grad.cpu().numpy()
        assert abs(np.linalg.norm(grad) / g[f"gnorm/{k}"] - 1) <= rel_l2_bound("default", 4096), k
        if f"grad/{k}" in g.files:
            want, got = g[f"grad/{k}"], grad
        else:
            want, got = g[f"gradc/{k}"], grad[:16, :16]
        assert rel_l2(got, want) <= 2 * rel_l2_bound("default", 4096), (k, rel_l2(got, want))   # 16x16 corners: noisier than a whole tensor
        post = p.detach().cpu().numpy()
        wantp = g[f"post/{k}"] if f"post/{k}" in g.files else g[f"postc/{k}"]
        gotp = post if f"post/{k}" in g.files else post[:16, :16]
        # first Adam step = lr * g / (|g| + eps): ~lr * sign(g) wherever |g| >> eps, so the update
        # survives bf16 gradient noise except where the gradient changes sign
        assert np.abs(gotp - wantp).max() <= 2 * 5e-4, k
        solid = np.abs(want) > max(0.2 * np.abs(want).max(), 1e-6)     # entries bf16 noise cannot flip
        if solid.any():
            assert np.abs(gotp - wantp)[solid].max() <= 1e-5, k
    l2, grads = _fused_grads(dev, synthetic, "default", rays, gt, u, N)
    assert l2 == loss                                            # same kernels, same inputs: reproducible
    _compare_with_oracle(oracle, synthetic, "default", rays, gt, u, N, loss, grads, "G6")


@pytest.mark.parametrize("kind", ["default", "structured"])
@pytest.mark.parametrize("shape", [(576, 64), (577, 64), (37, 65)])
def test_fused_training_vs_oracle(dev, synthetic, oracle, kind, shape):
    """The fused training path against the CPU oracle's fp32 autograd (train.py:51-54) at a full-tile
    size (576 x 64 = 144 tiles of 256 points) and at genuinely ragged ones: 577 x 64 (P % 256 = 64)
    and 37 x 65 (P = 2405: P % 256 = 101, P % 32 = 5 -- partial tile, partial dW slab, partial mask
    tile, out-of-range activation stores)."""
    B, N = shape
    gen = torch.Generator().manual_seed(B * 1000 + N)
    pose = torch.from_numpy(oracle.spherical_to_pose(4, -30, 0)).float()
    side = int(np.ceil(np.sqrt(B)))
    rays = oracle.camera_rays(pose, [side, side, synthetic.focal_from_fov(side)])[:B].contiguous()
    gt = torch.rand(B, 3, generator=gen)
    u = torch.rand(B, N, generator=gen)
    loss, grads = _fused_grads(dev, synthetic, kind, rays, gt, u, N)
    _compare_with_oracle(oracle, synthetic, kind, rays, gt, u, N, loss, grads, f"{B}x{N}")


def test_ragged_training_ignores_garbage_beyond_P(dev, synthetic, oracle):
    """Same ragged case through the C ABI with the activation and dY buffers pre-filled with NaN:
    nothing from points >= P (never written by the forward / dX kernels) may reach dW or db, and the
    kernels are deterministic at a ragged size too."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    lib = _lib.lib()
    B, N = 37, 65
    P = B * N
    pose = torch.from_numpy(oracle.spherical_to_pose(4, -30, 0)).float()
    rays = oracle.camera_rays(pose, [7, 7, synthetic.focal_from_fov(7)])[:B].contiguous().to(dev)
    u = torch.rand(B, N, generator=torch.Generator().manual_seed(1)).to(dev)
    net = Nerf().to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    packed, image = net.packed_weights(_lib.BF16), net.packed_weights(_lib.BF16_BWD)
    tbins = torch.linspace(2, 6, N + 1).to(dev)
    st = _lib.stream_ptr(dev)
    nb = int(lib.nerf_amd_train_activation_bytes(P))
    g = torch.randn(P, 4, generator=torch.Generator().manual_seed(5)).to(dev) * 1e-3
    outs = []
    for fill in (0xFF, 0x00, 0xFF):                     # 0xFFFF is a bf16 NaN
        acts = torch.full((nb,), fill, dtype=torch.uint8, device=dev)
        dys = torch.full((nb,), fill, dtype=torch.uint8, device=dev)
        raw = torch.empty(B, N, 4, device=dev)
        ts = torch.empty(B, N, device=dev)
        posx = torch.empty(P, 64, dtype=torch.bfloat16, device=dev)
        posd = torch.empty(P, 32, dtype=torch.bfloat16, device=dev)
        scratch = torch.empty(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), dtype=torch.uint8, device=dev)
        flat = torch.empty(int(lib.nerf_amd_param_count()), device=dev)
        ck = _lib.check
        ck(lib.nerf_amd_mlp_forward_train(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), _lib.ptr(packed), 0, 0, 0,
                                          _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(acts), B, N, st), "forward_train")
        ck(lib.nerf_amd_sample_encode_bf16(_lib.ptr(rays), _lib.ptr(ts), None, _lib.FLAG_TS_GIVEN, 0, 0, _lib.ptr(posx),
                                           _lib.ptr(posd), None, B, N, st), "encode")
        ck(lib.nerf_amd_mlp_backward(_lib.ptr(g), _lib.ptr(image), _lib.ptr(acts), _lib.ptr(dys), P, st), "backward")
        ck(lib.nerf_amd_param_gradients(_lib.ptr(g), _lib.ptr(acts), _lib.ptr(dys), _lib.ptr(posx), _lib.ptr(posd),
                                        _lib.ptr(scratch), _lib.ptr(flat), P, st), "param_gradients")
        torch.cuda.synchronize()
        outs.append((raw.clone(), flat.clone()))
    for raw, flat in outs:
        assert torch.isfinite(raw).all() and torch.isfinite(flat).all()
        assert torch.equal(raw, outs[0][0])
        assert float((flat - outs[0][1]).abs().max()) <= 1e-5 * float(outs[0][1].abs().max())


def test_nerf_forward_autograd(dev, oracle, synthetic):
    """Nerf.forward(v) with gradients (utils/nets.py:34-43 under autograd) = the fused training
    forward in points mode + the HIP backward: outputs and all 24 gradients vs the oracle."""
    from nerf_simple_amd.utils.nets import Nerf
    sd = synthetic.synthetic_state_dict(0, "default")
    v = synthetic.points_in_scene(300, seed=4)                    # 300 points: a ragged tile
    params = {k: p.clone().requires_grad_(True) for k, p in sd.items()}
    want_out = oracle.nerf_forward(params, v)
    want_out.pow(2).sum().backward()
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(sd)
    out = net(v.to(dev))
    assert out.requires_grad and out.shape == (300, 4)
    with torch.no_grad():
        assert torch.equal(out.detach(), net(v.to(dev)))          # the training forward computes what inference computes
    out.pow(2).sum().backward()
    worst = max(rel_l2(p.grad.cpu().numpy(), params[k].grad.numpy()) for k, p in net.named_parameters())
    print("points-mode rel L2 max:", worst)
    assert worst <= rel_l2_bound("default", 300)


def test_training_precision_contract(dev, synthetic, golden, oracle):
    """precision selects the inference kernel only: an 'fp16' module trains through the same bf16
    kernels as a 'bf16' one (identical gradients).  An 'fp32' module trains EXACTLY (layer by layer on the fp32 GEMM
    kernel, utils/generic_mlp.py), i.e. as exactly as the reference's own fp32 does: against the SAME step evaluated in
    float64 (the oracle on doubled inputs) every gradient tensor of this path is as close as the reference's own
    gradients are (fixture G6, captured from the reference: its early layers sit 1.3e-3 from the float64 values, its
    heads 1e-7 -- fp32 round-off through twelve layers of back-propagation), and the loss is the reference's to 1e-6.
    The tight end-to-end check of sampling, network, compositor and their backward that bf16 noise does not allow."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.training import train_step
    g = golden("train.npz")
    rays, gt, u, N = t(g["rays"]), t(g["gt"]), t(g["u"]), int(g["N"])
    la, ga = _fused_grads(dev, synthetic, "default", rays, gt, u, N, "bf16")
    lb, gb = _fused_grads(dev, synthetic, "default", rays, gt, u, N, "fp16")
    assert la == lb
    for k in ga:       # dW accumulates with float atomics: equal to summation order
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-5 * max(float(ga[k].abs().max()), 1e-12), k
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    net = Nerf(precision="fp32").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    opt = FusedAdam(net, lr=5e-4)
    loss = train_step(net, opt, rays.to(dev), gt.to(dev), N, u=u.to(dev))
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    sd64 = {k: v.double() for k, v in synthetic.synthetic_state_dict(0, "default").items()}
    loss64, grads64 = oracle.train_step_grads(sd64, rays.double(), u.double(), gt.double(), N)
    assert abs(float(loss) - float(loss64)) <= 1e-6 * float(loss64)
    report = {}
    for k, p in net.named_parameters():
        grad, post, exact = p.grad.cpu().numpy(), p.detach().cpu().numpy(), grads64[k].numpy()
        assert abs(np.linalg.norm(grad) / np.linalg.norm(exact) - 1) <= 2e-4, k        # observed <= 6e-5 (the reference: the same)
        if f"grad/{k}" in g.files:
            ref_g, ref_p = g[f"grad/{k}"], g[f"post/{k}"]
        else:                                    # the fixture keeps the 16 x 16 corner of the matrices
            ref_g, ref_p, grad, post, exact = g[f"gradc/{k}"], g[f"postc/{k}"], grad[:16, :16], post[:16, :16], exact[:16, :16]
        ours, theirs = rel_l2(grad, exact), rel_l2(ref_g, exact)
        report[k] = (ours, theirs)
        assert ours <= 3 * theirs + 3e-6, (k, ours, theirs)
        # Adam's first step moves every entry by ~ lr * sign(g): entries whose gradient is ~0 may land elsewhere
        assert np.abs(post - ref_p).max() <= 2.1 * 5e-4, k
        assert np.mean(np.abs(post - ref_p) <= 2e-6) >= 0.9, k
    pts = synthetic.points_in_scene(8, seed=1).to(dev)
    assert net(pts).requires_grad                                       # fp32 forward with gradients: the exact path
    with torch.no_grad():
        assert net(pts).shape == (8, 4)                                 # ... and without: the fp32 MFMA kernel
    with pytest.raises(RuntimeError, match="the fused training step is bf16"):
        GraphedTrainStep(net, opt, rays.shape[0], N)


def test_bf16_training_reduces_loss(dev, synthetic, oracle):
    """Config-5-shaped steps: a few Adam steps reduce the loss against a fixed target."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.training import train_step
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    sd = synthetic.synthetic_state_dict(0, "default")
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [32, 32, synthetic.focal_from_fov(32)]).to(dev)
    gt = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(2)).to(dev) * 0.2
    u = torch.rand(rays.shape[0], 64, generator=torch.Generator().manual_seed(3)).to(dev)
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(sd)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    losses = [float(train_step(net, opt, rays, gt, 64, u=u)) for _ in range(9)]
    assert losses[-1] < losses[0], losses


def test_training_forward_relu_mask_bits(dev, synthetic):
    """The ReLU mask bits the training forward appends to the saved activations (read by the
    backward dX chain instead of the activations) decode, by the layout documented in
    csrc/nerf_layout.h, to exactly `saved bf16 activation != 0` -- for every ReLU layer, with a
    ragged last tile (600 points = 2 full tiles + 88 points).  The activations themselves are
    decoded from their point-blocked layout and checked against the torch forward (bf16 budget)."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()
    net = Nerf().to(dev)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.synthetic_state_dict(5, "structured").items()})
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [5, 5, synthetic.focal_from_fov(5)]).float().contiguous().to(dev)   # 25 rays (the C ABI takes fp32)
    B, N = rays.shape[0], 24
    P = B * N                                                                        # 600
    u = torch.rand(B, N, generator=torch.Generator().manual_seed(4)).to(dev)
    tbins = torch.linspace(2, 6, N + 1).to(dev)
    raw = torch.empty(B, N, 4, device=dev)
    ts = torch.empty(B, N, device=dev)
    nbytes = int(lib.nerf_amd_train_activation_bytes(P))
    acts = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    packed = net.packed_weights(_lib.BF16)
    _lib.check(lib.nerf_amd_mlp_forward_train(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), _lib.ptr(packed), 0, 0, 0,
                                              _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(acts), B, N, _lib.stream_ptr(dev)),
               "nerf_amd_mlp_forward_train")
    torch.cuda.synchronize()
    host = acts.cpu().numpy()
    ntiles = (P + 255) // 256
    region = 10 * ntiles * 256 * 512        # point-blocked bf16 activations: 10 layers x tiles x 128 KiB
    assert nbytes == region + 10 * ntiles * 8192
    masks = host[region:].view(np.uint32).reshape(10, ntiles, 4, 512)               # [layer, tile, dword, thread]
    # thread (wave, lane), column block cb, pair Q, word j, half e  ->  (point, feature, dword, bit)
    tid = np.arange(512)
    wave, lane = tid >> 6, tid & 63
    checked = 0
    for L in (0, 1, 2, 3, 4, 5, 6, 7, 9):
        width = 128 if L == 9 else 256
        # layer L, tile t: [feature chunk f/8 (32)][point in tile (256)][8 bf16]  ->  a[p, f]
        blk = host[L * ntiles * 131072: (L + 1) * ntiles * 131072].view(np.uint16).reshape(ntiles, 32, 256, 8)
        a = blk.transpose(0, 2, 1, 3).reshape(ntiles * 256, 256)[:P, :width]
        for tile in range(ntiles):
            for cb in range(2):
                pt = tile * 256 + wave * 32 + cb * 16 + (lane & 15)
                ok = pt < P
                for Q in range(width // 32):
                    for j in range(4):
                        for e in range(2):
                            feat = 32 * Q + 16 * (j >> 1) + 4 * (lane >> 4) + 2 * (j & 1) + e
                            bit = (masks[L, tile, cb * 2 + (Q >> 2)] >> ((Q & 3) * 4 + j + 16 * e)) & 1
                            want = a[np.where(ok, pt, 0), feat] != 0
                            assert np.array_equal(bit[ok].astype(bool), want[ok]), (L, tile, cb, Q, j, e)
                            checked += int(ok.sum())
        assert 0.02 < (a != 0).mean() < 0.98, L          # the masks are not trivial
    assert checked == P * (8 * 256 + 128)
    # the decoded layer-0 activations are relu(layers_0.0(gamma(x))) of the same sample points
    from nerf_simple_amd.utils.xyz import positional_encoder
    o, dd = rays[:, :3], rays[:, 3:]
    pts = (o[:, None, :] + ts[..., None] * dd[:, None, :]).reshape(-1, 3)
    posx, _ = positional_encoder(torch.cat([pts, dd[:, None, :].expand(B, N, 3).reshape(-1, 3)], dim=-1).contiguous())
    sd = net.state_dict()
    want0 = torch.relu(posx.float() @ sd["layers_0.0.weight"].float().T + sd["layers_0.0.bias"].float()).cpu()
    blk0 = host[:ntiles * 131072].view(np.uint16).reshape(ntiles, 32, 256, 8).transpose(0, 2, 1, 3).reshape(-1, 256)[:P]
    got0 = torch.from_numpy(blk0.astype(np.int32) << 16).view(torch.float32)
    err = (got0 - want0).abs()
    badmask = err > 2e-2 * float(want0.abs().max())
    if bool(badmask.any()):
        rows, cols = torch.nonzero(badmask, as_tuple=True)
        print("layer-0 activation mismatches:", int(badmask.sum()), "rows", sorted(set(rows.tolist()))[:40],
              "cols", sorted(set(cols.tolist()))[:40])
    assert not bool(badmask.any())


def test_sample_encode_bf16_matches_fp32_encoder(dev, synthetic):
    """The training front end (one thread per point, hardware sin / cos in revolutions, bf16 rows
    padded to 64 / 32 columns for the dW kernel) against the fp32 parity encoder
    nerf_amd_sample_encode on the same rays and sample positions: within one bf16 rounding
    (values are in [-1, 1] apart from the raw coordinates), pad columns exactly zero."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [9, 7, synthetic.focal_from_fov(9)]).float().contiguous().to(dev)
    B, N = rays.shape[0], 40                                          # 2520 points: a ragged last block
    u = torch.rand(B, N, generator=torch.Generator().manual_seed(8)).to(dev)
    tbins = torch.linspace(2, 6, N + 1).to(dev)
    P = B * N
    posx = torch.empty(P, 63, device=dev); posd = torch.empty(P, 27, device=dev); ts = torch.empty(B, N, device=dev)
    _lib.check(lib.nerf_amd_sample_encode(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), 0, 0, 0,
                                          _lib.ptr(posx), _lib.ptr(posd), _lib.ptr(ts), B, N, _lib.stream_ptr(dev)),
               "nerf_amd_sample_encode")
    px = torch.full((P, 64), 7.0, dtype=torch.bfloat16, device=dev)
    pd = torch.full((P, 32), 7.0, dtype=torch.bfloat16, device=dev)
    ts2 = torch.empty(B, N, device=dev)
    _lib.check(lib.nerf_amd_sample_encode_bf16(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), 0, 0, 0,
                                               _lib.ptr(px), _lib.ptr(pd), _lib.ptr(ts2), B, N, _lib.stream_ptr(dev)),
               "nerf_amd_sample_encode_bf16")
    torch.cuda.synchronize()
    assert torch.equal(ts, ts2)
    assert float(px[:, 63].abs().max()) == 0 and float(pd[:, 27:].abs().max()) == 0
    for got, want in ((px[:, :63].float(), posx), (pd[:, :27].float(), posd)):
        tol = 2.0 ** -8 * want.abs().clamp(min=1.0) + 2e-6          # half an ulp of bf16 is 2^-9 relative
        assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())


def test_training_kernels_are_deterministic(dev, synthetic):
    """forward_train and mlp_backward have no atomics: repeated on identical inputs, every output
    byte must repeat (a difference is a race or an unpadded hardware hazard -- this is the test
    that exposes the store-data hazard described at store_granule in csrc/nerf_device.h);
    param_gradients (float atomics) must repeat to rounding."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()
    net = Nerf().to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [32, 32, synthetic.focal_from_fov(32)]).float().contiguous().to(dev)
    B, N = rays.shape[0], 64
    P = B * N                                                         # 65536 points: every CU gets a tile
    u = torch.rand(B, N, generator=torch.Generator().manual_seed(4)).to(dev)
    tbins = torch.linspace(2, 6, N + 1).to(dev)
    packed, image = net.packed_weights(_lib.BF16), net.packed_weights(_lib.BF16_BWD)
    st = _lib.stream_ptr(dev)
    nb = int(lib.nerf_amd_train_activation_bytes(P))
    g = torch.randn(P, 4, generator=torch.Generator().manual_seed(5)).to(dev) * 1e-3
    posx = torch.empty(P, 64, dtype=torch.bfloat16, device=dev)
    posd = torch.empty(P, 32, dtype=torch.bfloat16, device=dev)
    ts = torch.empty(B, N, device=dev)
    _lib.check(lib.nerf_amd_sample_encode_bf16(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), 0, 0, 0, _lib.ptr(posx),
                                               _lib.ptr(posd), _lib.ptr(ts), B, N, st), "nerf_amd_sample_encode_bf16")
    scratch = torch.empty(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), dtype=torch.uint8, device=dev)

    def run(acts_in=None):
        raw = torch.zeros(B, N, 4, device=dev)
        t2 = torch.zeros(B, N, device=dev)
        acts = torch.zeros(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.nerf_amd_mlp_forward_train(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), _lib.ptr(packed), 0, 0, 0,
                                                  _lib.ptr(raw), _lib.ptr(t2), _lib.ptr(acts), B, N, st), "forward_train")
        src = acts if acts_in is None else acts_in
        dys = torch.zeros(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.nerf_amd_mlp_backward(_lib.ptr(g), _lib.ptr(image), _lib.ptr(src), _lib.ptr(dys), P, st), "backward")
        flat = torch.empty(int(lib.nerf_amd_param_count()), device=dev)
        _lib.check(lib.nerf_amd_param_gradients(_lib.ptr(g), _lib.ptr(src), _lib.ptr(dys), _lib.ptr(posx), _lib.ptr(posd),
                                                _lib.ptr(scratch), _lib.ptr(flat), P, st), "param_gradients")
        return raw, acts, dys, flat

    raw0, acts0, dys0, flat0 = run()
    assert torch.isfinite(flat0).all() and float(flat0.abs().max()) > 0
    for _ in range(25):
        raw, acts, dys, flat = run(acts0)
        assert torch.equal(raw, raw0) and torch.equal(acts, acts0)
        assert torch.equal(dys, dys0)
        assert float((flat - flat0).abs().max()) <= 1e-5 * float(flat0.abs().max())


def test_graphed_train_step_matches_eager(dev, golden, synthetic):
    """training.GraphedTrainStep (the step captured as hipGraphs: forward .. parameter gradients,
    then Adam with device-resident hyper-parameters + re-pack) against the eager train_step with
    FusedAdam on the same rays / jitter / targets, 6 decayed steps: same losses (the dW atomics'
    order is the only difference), same parameters to Adam's eps-corner statistics, and the packed
    images follow (inference after training sees the new weights)."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import train_step, lr_decay_factor, GraphedTrainStep
    g = golden("train.npz")
    rays, gt, u, N = t(g["rays"]).to(dev), t(g["gt"]).to(dev), t(g["u"]).to(dev), int(g["N"])
    decay = lr_decay_factor(5e-4, 4e-4, 10)
    us = [torch.rand(u.shape, generator=torch.Generator().manual_seed(100 + i)).to(dev) for i in range(6)]
    runs = []
    for graphed in (False, True):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        opt = FusedAdam(net, lr=5e-4)
        if graphed:
            stepper = GraphedTrainStep(net, opt, rays.shape[0], N)
            losses = [float(stepper.step(rays, gt, u=us[i], decay=decay)) for i in range(6)]
            assert opt.step_count == 6
            # without u the step draws torch.rand(B, N) from the CPU generator like the reference
            saved = torch.get_rng_state()
            try:
                torch.manual_seed(77)
                st = torch.get_rng_state()
                want_u = torch.rand(rays.shape[0], N)
                want_next = torch.rand(3)
                torch.set_rng_state(st)
                probe_net_state = opt.flat.clone()
                stepper.step(rays, gt)
                assert torch.equal(stepper.u.cpu(), want_u) and torch.equal(torch.rand(3), want_next)
                opt.flat.copy_(probe_net_state)          # undo that extra update for the comparison below
                net.repack_from_flat(opt.flat)
                opt.step_count -= 1
            finally:
                torch.set_rng_state(saved)
        else:
            losses = [float(train_step(net, opt, rays, gt, N, u=us[i], decay=decay)) for i in range(6)]
        assert abs(opt.param_groups[0]["lr"] - 5e-4 * decay ** 6) < 1e-12
        with torch.no_grad():
            probe = net(rays[:8].new_zeros(8, 6) + 0.1).cpu()
        runs.append((losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu(), probe))
    (la, pa, qa), (lb, pb, qb) = runs
    assert la[-1] < la[0] and lb[-1] < lb[0]
    np.testing.assert_allclose(la, lb, rtol=2e-3)
    d = (pa - pb).abs()
    assert float(d.max()) <= 6 * 5e-4 and float(d.mean()) <= 1e-5 and float((d > 1e-5).float().mean()) <= 0.06
    assert float((qa - qb).abs().max()) <= 2e-2 * max(1.0, float(qa.abs().max()))


def test_graphed_train_step_without_host_sync(dev, golden, synthetic):
    """Six graphed steps with explicit u, a decaying learning rate and NO host sync in between (the
    config-5 loop): the Adam scalars of step k must not be overwritten by step k+1's while step k is
    still queued.  A long GPU-side delay in front of the first step makes the host run far ahead."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import lr_decay_factor, GraphedTrainStep
    g = golden("train.npz")
    rays, gt, N = t(g["rays"]).to(dev), t(g["gt"]).to(dev), int(g["N"])
    decay = lr_decay_factor(5e-4, 1e-4, 6)
    us = [torch.rand(rays.shape[0], N, generator=torch.Generator().manual_seed(200 + i)).to(dev) for i in range(6)]
    finals = []
    for synced in (True, False):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        opt = FusedAdam(net, lr=5e-4)
        stepper = GraphedTrainStep(net, opt, rays.shape[0], N)
        torch.cuda.synchronize()
        if not synced:
            torch.cuda._sleep(int(2e9))                      # ~1 s of GPU time queued ahead of the steps
        for i in range(6):
            stepper.step(rays, gt, u=us[i], decay=decay)
            if synced:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        finals.append(opt.flat.clone().cpu())
    d = (finals[0] - finals[1]).abs()
    # Identical schedules: only the dW atomics' summation order differs between the runs, which flips
    # the sign of Adam's ~lr-sized move for the few entries whose gradient is noise (|g| ~ eps) -- at
    # most 2 lr per step for those.  A step run with its successor's scalars (lr x0.76, bias
    # corrections 0.19 for 0.1) would instead shift EVERY entry by ~lr/2: mean ~1e-4.
    print("no-sync vs synced: max", float(d.max()), "mean", float(d.mean()), "frac > 1e-5", float((d > 1e-5).float().mean()))
    assert float(d.max()) <= 12 * 5e-4 and float(d.mean()) <= 1e-5 and float((d > 1e-5).float().mean()) <= 0.06


def test_fused_adam_matches_torch(dev, golden, synthetic):
    """N3: optim.FusedAdam == torch.optim.Adam (reference train.py:43,55-57) -- on golden G6 for the
    first step, and over several decayed steps of the fused bf16 path against torch's optimizer."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import train_step, lr_decay_factor
    g = golden("train.npz")
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    opt = FusedAdam(net, lr=5e-4)
    train_step(net, opt, t(g["rays"]).to(dev), t(g["gt"]).to(dev), int(g["N"]), u=t(g["u"]).to(dev))
    for k, p in net.named_parameters():
        want_g = g[f"grad/{k}"] if f"grad/{k}" in g.files else g[f"gradc/{k}"]
        wantp = g[f"post/{k}"] if f"post/{k}" in g.files else g[f"postc/{k}"]
        post = p.detach().cpu().numpy()
        gotp = post if f"post/{k}" in g.files else post[:16, :16]
        solid = np.abs(want_g) > max(0.2 * np.abs(want_g).max(), 1e-6)     # entries bf16 noise cannot flip
        if solid.any():
            assert np.abs(gotp - wantp)[solid].max() <= 1e-5, k
        assert np.abs(gotp - wantp).max() <= 1e-3, k

    # the update rule itself, free of the dW atomics' run-to-run summation order: both optimizers
    # are fed the SAME gradients for 5 decayed steps (|g| spans eps .. 1: the 1/(sqrt(v)+eps) corner)
    nets = []
    for make in (lambda n: torch.optim.Adam(n.parameters(), lr=5e-4), lambda n: FusedAdam(n, lr=5e-4)):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        opt = make(net)
        gen = torch.Generator().manual_seed(11)
        for _ in range(5):
            for prm in net.parameters():
                mag = torch.pow(10.0, torch.rand(prm.shape, generator=gen) * 8 - 8)
                prm.grad = (torch.randn(prm.shape, generator=gen) * mag).to(dev)
            opt.step()
            for pg in opt.param_groups:
                pg["lr"] *= 0.9
        nets.append(torch.cat([prm.detach().reshape(-1) for prm in net.parameters()]).cpu())
    assert float((nets[0] - nets[1]).abs().max()) <= 2e-7

    # several steps, lr decay through param_groups as the reference's loop does
    rays = t(g["rays"]).to(dev)
    gt = t(g["gt"]).to(dev)
    u = t(g["u"]).to(dev)
    decay = lr_decay_factor(5e-4, 4e-4, 10)
    finals = []
    for make in (lambda n: torch.optim.Adam(n.parameters(), lr=5e-4), lambda n: FusedAdam(n, lr=5e-4)):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        opt = make(net)
        losses = [float(train_step(net, opt, rays, gt, int(g["N"]), u=u, decay=decay)) for _ in range(6)]
        assert abs(opt.param_groups[0]["lr"] - 5e-4 * decay ** 6) < 1e-12
        finals.append((losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()))
        # the packed images follow the fused update: inference right after training sees new weights
        with torch.no_grad():
            out = net(rays[:8].new_zeros(8, 6) + 0.1)
        assert torch.isfinite(out).all()
    (la, pa), (lb, pb) = finals
    assert la[-1] < la[0] and lb[-1] < lb[0]
    np.testing.assert_allclose(la, lb, rtol=2e-3)
    # identical update rule; the dW atomics' summation order is the only difference between the
    # runs, which Adam amplifies only where |g| ~ eps (an entry can move by at most lr per step)
    # (measured: mean 1.2e-6, 2.3 % of the entries beyond 1e-5; the bound leaves room for other orders)
    d = (pa - pb).abs()
    assert float(d.max()) <= 6 * 5e-4 and float(d.mean()) <= 1e-5 and float((d > 1e-5).float().mean()) <= 0.06


def test_graphed_step_with_device_jitter(dev, golden, synthetic):
    """GraphedTrainStep(device_rng=True): every replay draws fresh stratified jitter inside the kernels (counter RNG,
    seed + step count read from device memory through NERF_AMD_SEED_IN_MEMORY -- the captured launches carry no per-step
    argument).  Step k's gradient equals the eager train_step(device_rng=True, seed=seed + k, ray_id0=...) gradient at
    the same weights (lr = 0 keeps them fixed), consecutive steps see different sample positions, and another ray-id
    offset (another data-parallel rank) another jitter."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import train_step, GraphedTrainStep
    g = golden("train.npz")
    rays, gt, N = t(g["rays"]).to(dev), t(g["gt"]).to(dev), int(g["N"])
    B = rays.shape[0]

    def fresh():
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        return net

    net = fresh()
    stepper = GraphedTrainStep(net, FusedAdam(net, lr=0.0), B, N, device_rng=True, seed=40, ray_id0=3 * B)
    with pytest.raises(RuntimeError, match="device_rng"):
        stepper.step(rays, gt, u=torch.zeros(B, N, device=dev))
    seen = []
    for k in (1, 2, 3):
        loss = float(stepper.step(rays, gt))
        torch.cuda.synchronize()
        seen.append((loss, stepper.ts.clone(), stepper.grads.clone()))
        ref = fresh()
        opt = torch.optim.SGD(ref.parameters(), lr=0.0)
        want_loss = float(train_step(ref, opt, rays, gt, N, device_rng=True, seed=40 + k, ray_id0=3 * B))
        want = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        assert abs(loss - want_loss) <= 1e-6 * abs(want_loss), (k, loss, want_loss)
        assert float((stepper.grads - want).abs().max()) <= 2e-5 * float(want.abs().max()), k     # atomics' order only
    assert not torch.equal(seen[0][1], seen[1][1]) and not torch.equal(seen[1][1], seen[2][1])    # fresh positions per step
    ts = seen[0][1]
    edges = torch.linspace(2, 6, N + 1).to(dev)
    assert bool((ts >= edges[:-1]).all()) and bool((ts <= edges[1:]).all())                        # stratified: one per bin
    other = GraphedTrainStep(fresh_net := fresh(), FusedAdam(fresh_net, lr=0.0), B, N, device_rng=True, seed=40, ray_id0=0)
    other.step(rays, gt)
    torch.cuda.synchronize()
    assert not torch.equal(other.ts, seen[0][1])


def test_diverged_training_is_loud(dev, golden, synthetic):
    """A NaN that reaches the weights (a diverged run): the reference's loss turns NaN.  The kernels' integer ReLU can
    turn NaNs into finite garbage instead, so GraphedTrainStep watches the forward's range flag: a later step raises
    FloatingPointError.  A healthy run never trips it."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    g = golden("train.npz")
    rays, gt, u, N = t(g["rays"]).to(dev), t(g["gt"]).to(dev), t(g["u"]).to(dev), int(g["N"])
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    opt = FusedAdam(net, lr=5e-4)
    stepper = GraphedTrainStep(net, opt, rays.shape[0], N, check_every=1)
    for _ in range(5):
        stepper.step(rays, gt, u=u)
    torch.cuda.synchronize()
    stepper.step(rays, gt, u=u)                               # healthy: nothing raised
    with torch.no_grad():
        opt.flat[70000] = float("nan")                        # one weight of layers_0.2 goes bad behind everyone's back
    net.repack_from_flat(opt.flat)
    with pytest.raises(FloatingPointError, match="non-finite values inside the network"):
        for _ in range(4):
            stepper.step(rays, gt, u=u)
            torch.cuda.synchronize()


def test_diverged_eager_training_is_loud(dev, golden, synthetic):
    """The eager step (train_step, Nerf.forward with gradients) watches the same status words as the graphed one: a NaN
    weight that the integer ReLU would turn into a dead unit -- finite loss, broken network -- raises at a later call;
    a healthy run never does, and copies of the module do not carry the watch."""
    import copy
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.training import train_step
    g = golden("train.npz")
    rays, gt, u, N = t(g["rays"]).to(dev), t(g["gt"]).to(dev), t(g["u"]).to(dev), int(g["N"])
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    for _ in range(4):
        loss = train_step(net, opt, rays, gt, N, u=u)
        torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    assert "_watch" in net.__dict__ and "_watch" not in copy.deepcopy(net).__dict__
    with torch.no_grad():
        net.layers_0[2].weight[209, 112] = float("nan")
    with pytest.raises(FloatingPointError, match="non-finite values inside the network"):
        for _ in range(3):
            train_step(net, opt, rays, gt, N, u=u)
            torch.cuda.synchronize()


def test_pack_train_decides_the_weight_range_word(dev, synthetic):
    """nerf_amd_pack_weights_train states the weight-range word for exactly the weights it packs (a clear kernel in front of
    the packing kernel): a NaN weight sets it, weights repaired through the flat vector clear it at the next re-pack -- it
    used to stay set until nerf_amd_pack_weights (advisor finding, round 3) -- and no other word of the block is left set."""
    from nerf_simple_amd import _lib
    lib = _lib.lib()
    st = _lib.stream_ptr(dev)
    flat = synthetic.flatten_state_dict(synthetic.synthetic_state_dict(3, "default")).to(dev)
    a = torch.zeros(int(lib.nerf_amd_packed_bytes(_lib.BF16)), dtype=torch.uint8, device=dev)
    b = torch.zeros(int(lib.nerf_amd_packed_bytes(_lib.BF16_BWD)), dtype=torch.uint8, device=dev)
    off = int(lib.nerf_amd_packed_status_offset(_lib.BF16))

    def words():
        torch.cuda.synchronize()
        return a[off:off + 64].view(torch.int32).cpu().tolist()

    _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(a), _lib.BF16, st), "p")
    for bad_at in (None, 5, 300_000, None, 595_840, None):          # weights (fp32 biases are not operands: the forward's flag sees them)
        w = flat.clone()
        if bad_at is not None:
            w[bad_at] = float("nan") if bad_at != 300_000 else float("inf")
        a[off:off + 4].view(torch.int32)[0] = 7                              # word 0 (the forward's flag) is cleared by the re-pack
        for _ in range(2):                                                   # twice: the scratch words start from zero each time
            _lib.check(lib.nerf_amd_pack_weights_train(_lib.ptr(w), _lib.ptr(a), _lib.ptr(b), st), "pt")
            got = words()
            assert got[0] == 0 and got[1] == (0 if bad_at is None else 1), (bad_at, got)
            assert not any(got[2:]), (bad_at, got)
