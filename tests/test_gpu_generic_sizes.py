"""Nerf(Lp, Ld, H) of sizes other than the default (reference utils/nets.py:9-32 takes any): the layer-by-layer fp32
path (utils/generic_mlp.py on nerf_amd_linear_f32) against the CPU oracle -- forward, render, gradients, a training
step -- and the GEMM kernel itself against torch on ragged shapes and every stride pattern the path uses.

Tolerance: fp32 products and fp32 accumulation on both sides, different summation order (16-wide k steps here, MKL
blocking there, float atomics in the split reductions): REL = 2e-5 of the tensor's largest magnitude, 3x the observed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 2e-5
SIZES = [(6, 2, 128), (10, 4, 64), (3, 1, 40), (2, 3, 37), (12, 5, 300), (1, 1, 2)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with a GPU: pytest -m gpu"
    from nerf_simple_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel_err(got, want):
    want = want.double()
    return float((got.double().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-30)


def make(Lp, Ld, H, dev, seed=0):
    from nerf_simple_amd.utils.nets import Nerf
    torch.manual_seed(seed)
    net = Nerf(Lp, Ld, H)
    with torch.no_grad():                      # nn.Linear's default init renders almost black: give the heads some gain
        net.sigma_fc[0].weight.mul_(4.0)
        net.color_fc[2].weight.mul_(4.0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    return net.to(dev), sd


def points(P, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(P, 3, generator=g) * 6 - 3
    d = torch.randn(P, 3, generator=g)
    return torch.cat([x, d / d.norm(dim=1, keepdim=True)], 1)


def test_linear_kernel_against_torch(dev):
    """nerf_amd_linear_f32 on ragged shapes: plain, transposed-A, weight slices, masks, bias / ReLU / accumulate, the
    ones-vector column sum, the split reduction, empty sizes and the argument errors."""
    from nerf_simple_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    st = _lib.stream_ptr(dev)

    def call(A, sa_i, sa_k, mask, B, sb_k, sb_j, bias, C, ldc, M, N, K, flags, a_off=0, b_off=0, c_off=0):
        return lib.nerf_amd_linear_f32(A.data_ptr() + 4 * a_off, sa_i, sa_k, None if mask is None else mask.data_ptr() + 4 * a_off,
                                       B.data_ptr() + 4 * b_off, sb_k, sb_j, _lib.ptr(bias), C.data_ptr() + 4 * c_off, ldc,
                                       M, N, K, flags, st)

    for M, N, K in ((1, 1, 1), (5, 3, 7), (64, 64, 16), (65, 63, 17), (257, 130, 300), (1000, 40, 9)):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K + 5, generator=g)                       # the layer reads columns 2 .. 2+K of a wider matrix
        b = torch.randn(N, generator=g)
        c0 = torch.randn(M, N + 3, generator=g)                      # and writes columns 1 .. 1+N of a wider output
        for flags in (0, 1, 2, 3):
            C = c0.clone().to(dev)
            assert call(x.to(dev), K, 1, None, w.to(dev), 1, K + 5, b.to(dev), C, N + 3, M, N, K, flags, b_off=2, c_off=1) == 0
            want = x @ w[:, 2:2 + K].T + b + (c0[:, 1:1 + N] if flags & 2 else 0)
            if flags & 1:
                want = want.clamp_min(0)
            assert rel_err(C[:, 1:1 + N], want) <= REL, (M, N, K, flags)
            assert torch.equal(C[:, 0].cpu(), c0[:, 0]) and torch.equal(C[:, 1 + N:].cpu(), c0[:, 1 + N:])   # neighbours untouched
    # the backward's three products with a ReLU mask, P = 20000 points (split reduction with atomics for dW / db)
    P, n_out, k_in = 20000, 37, 50
    dy = torch.randn(P, n_out, generator=g)
    act = torch.randn(P, n_out, generator=g).clamp_min(0)
    x = torch.randn(P, k_in, generator=g)
    w = torch.randn(n_out, k_in, generator=g)
    dym = dy * (act > 0)
    gw = torch.zeros(n_out, k_in, device=dev)
    gb = torch.zeros(n_out, device=dev)
    dx = torch.empty(P, k_in, device=dev)
    one = torch.ones(1, device=dev)
    dyd, actd, xd, wd = dy.to(dev), act.to(dev), x.to(dev), w.to(dev)
    assert call(dyd, 1, n_out, actd, xd, k_in, 1, None, gw, k_in, n_out, k_in, P, 2) == 0
    assert call(dyd, 1, n_out, actd, one, 0, 0, None, gb, 1, n_out, 1, P, 2) == 0
    assert call(dyd, n_out, 1, actd, wd, k_in, 1, None, dx, k_in, P, k_in, n_out, 0) == 0
    assert rel_err(gw, dym.double().T @ x.double()) <= REL
    assert rel_err(gb, dym.double().sum(0)) <= REL
    assert rel_err(dx, dym @ w) <= REL
    # empty sizes are no-ops, K = 0 writes the bias
    C = torch.full((4, 4), 7.0, device=dev)
    assert call(xd, 1, 1, None, wd, 1, 1, None, C, 4, 0, 4, 3, 0) == 0 and call(xd, 1, 1, None, wd, 1, 1, None, C, 4, 4, 0, 3, 0) == 0
    assert float(C.min()) == 7.0
    bias = torch.arange(4.0, device=dev)
    assert call(xd, 1, 1, None, wd, 1, 1, bias, C, 4, 4, 4, 0, 0) == 0
    assert torch.equal(C, bias.expand(4, 4))
    # argument errors (NERF_AMD_EINVAL = -1): negative sizes, ldc < N, unknown flags, NULL operands
    assert call(xd, 1, 1, None, wd, 1, 1, None, C, 4, -1, 4, 3, 0) == -1
    assert call(xd, 1, 1, None, wd, 1, 1, None, C, 3, 4, 4, 3, 0) == -1
    assert call(xd, 1, 1, None, wd, 1, 1, None, C, 4, 4, 4, 3, 4) == -1
    assert lib.nerf_amd_linear_f32(None, 1, 1, None, wd.data_ptr(), 1, 1, None, C.data_ptr(), 4, 4, 4, 3, 0, st) == -1


@pytest.mark.parametrize("Lp,Ld,H", SIZES)
def test_forward_any_size_vs_oracle(dev, oracle, Lp, Ld, H):
    """Nerf(Lp, Ld, H).forward == the reference module's forward (oracle restatement with the same sizes), ragged P."""
    net, sd = make(Lp, Ld, H, dev)
    for P in (1, 257, 1000):
        v = points(P, seed=P)
        with torch.no_grad():
            got = net(v.to(dev))
            want = oracle.nerf_forward(sd, v, Lp, Ld)
        assert got.shape == (P, 4) and not got.requires_grad
        assert rel_err(got, want) <= REL, (P, rel_err(got, want))
    assert net(torch.empty(0, 6, device=dev)).shape == (0, 4)


def test_inference_in_chunks_is_the_same(dev, monkeypatch):
    """Without gradients an image's worth of points goes through the layers chunk by chunk (bounded activations in HBM):
    identical to one pass, ragged last chunk included."""
    from nerf_simple_amd.utils import generic_mlp
    net, _ = make(6, 2, 128, dev)
    v = points(1000, seed=3).to(dev)
    with torch.no_grad():
        whole = net(v)
        monkeypatch.setattr(generic_mlp, "INFERENCE_CHUNK", 300)
        parts = net(v)
    assert torch.equal(whole, parts)
    assert net(v).requires_grad                      # with gradients: one pass, activations kept for the backward


@pytest.mark.parametrize("Lp,Ld,H", SIZES[:5])
def test_gradients_any_size_vs_oracle_autograd(dev, oracle, Lp, Ld, H):
    """Every parameter gradient of the hand-written backward against torch autograd through the oracle's forward."""
    net, sd = make(Lp, Ld, H, dev, seed=3)
    P = 1500
    v = points(P, seed=9)
    g_out = torch.randn(P, 4, generator=torch.Generator().manual_seed(2))
    out = net(v.to(dev))
    assert out.requires_grad
    out.backward(g_out.to(dev))
    ref = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    oracle.nerf_forward(ref, v, Lp, Ld).backward(g_out)
    worst = {}
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.shape == ref[k].grad.shape, k
        worst[k] = rel_err(p.grad, ref[k].grad)
    assert max(worst.values()) <= REL, worst
    from nerf_simple_amd.parallel import flat_grad_view
    flat = flat_grad_view([p for _, p in net.named_parameters()])       # one bucket for the all-reduce / FusedAdam
    assert flat is not None and flat.numel() == sum(p.numel() for p in net.parameters())


def test_render_and_training_step_any_size(dev, oracle):
    """render_nerf / render_view / train_step with a non-default Nerf: the reference's own composition (sampling ->
    net.forward -> volume_render) against the oracle's, the loss and every gradient of a training step against
    autograd, FusedAdam and torch.optim.Adam on the module."""
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import train_step
    from nerf_simple_amd.utils.rendering import generate_rays, render_nerf, render_view
    from nerf_simple_amd.utils.xyz import spherical_to_pose
    Lp, Ld, H, N = 6, 2, 128, 24
    net, sd = make(Lp, Ld, H, dev, seed=5)
    pose = spherical_to_pose(4, -30, 20)
    cam = [12, 12, 13.0]
    rays = generate_rays(pose, cam, dev)
    B = rays.shape[0]
    u = torch.rand(B, N, generator=torch.Generator().manual_seed(4))

    def oracle_render(state):
        ts = oracle.sample_ts(u, 2, 6)
        q, dn = oracle.query_points(rays.cpu(), ts)
        return oracle.volume_render(oracle.nerf_forward(state, q, Lp, Ld).reshape(B, N, 4), ts, dn)

    with torch.no_grad():
        got = render_nerf(rays, net, N, u=u.to(dev))
        want = oracle_render(sd)
        px = render_view(net, pose, cam, N=N, u=u.to(dev))
    for name, g_, w_ in zip(("rgb", "disp", "alpha", "acc", "w"), got, want):
        assert rel_err(g_, w_) <= 5 * REL, name
    assert torch.equal(px[:, :3], got[0].clamp(0, 1)) and torch.equal(px[:, 3], got[1])
    # one training step: loss and gradients, then the two optimizers agree on the updated parameters
    gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(6))
    ref = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    loss_ref = ((oracle_render(ref)[0] - gt) ** 2).mean()
    loss_ref.backward()
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    loss = train_step(net, opt, rays, gt.to(dev), N, u=u.to(dev))
    assert abs(float(loss) - float(loss_ref.detach())) <= 1e-5 * float(loss_ref.detach())
    for k, p in net.named_parameters():
        assert rel_err(p.grad, ref[k].grad) <= 5 * REL, k
    after_torch = {k: p.detach().clone() for k, p in net.named_parameters()}
    net2, _ = make(Lp, Ld, H, dev, seed=5)
    fused = FusedAdam(net2, lr=5e-4)
    loss2 = train_step(net2, fused, rays, gt.to(dev), N, u=u.to(dev))
    assert abs(float(loss2) - float(loss)) <= 1e-6 * float(loss)
    for k, p in net2.named_parameters():
        assert float((p.detach() - after_torch[k]).abs().max()) <= 2e-6, k
    # what the fused-only entry points say for such a module
    from nerf_simple_amd.training import GraphedTrainStep
    with pytest.raises(RuntimeError, match="no packed weight image"):
        GraphedTrainStep(net2, fused, B, N)


def test_other_sizes_golden_from_the_reference(dev, golden):
    """G9 (captured by running the reference's own Nerf(Lp, Ld, H) at three sizes): the layer-by-layer path reproduces
    its outputs and its gradients -- the module rebuilt from the fixture's seed, initial weights checked bit for bit."""
    from test_oracle_golden import g9_module
    g = golden("sizes.npz")
    for i in range(len(g["sizes"])):
        net, tag, _ = g9_module(g, i)
        net = net.to(dev)
        v, g_out = torch.from_numpy(g[f"{tag}/v"]).to(dev), torch.from_numpy(g[f"{tag}/g_out"]).to(dev)
        out = net(v)
        assert rel_err(out.detach(), torch.from_numpy(g[f"{tag}/out"])) <= REL, tag
        out.backward(g_out)
        for k, p in net.named_parameters():
            grad = p.grad.cpu().numpy()
            assert abs(np.linalg.norm(grad.astype(np.float64)) / float(g[f"{tag}/gnorm/{k}"]) - 1) <= 5 * REL, (tag, k)
            ref = g[f"{tag}/grad/{k}"]
            got = grad if grad.ndim == 1 else grad[:16, :16]
            assert np.abs(got - ref).max() <= REL * max(float(np.abs(grad).max()), 1e-30), (tag, k)
