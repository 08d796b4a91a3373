"""GPU tests of the boundary's edges: the fp16 range guard, the reference's degenerate
N = 1 result, a reference-format checkpoint through utils/checkpoint on the GPU, a foreign ``net`` object."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from nerf_simple_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def scene_rays(oracle, synthetic, side=12):
    pose = torch.from_numpy(oracle.spherical_to_pose(4, -30, 0)).float()
    return oracle.camera_rays(pose, [side, side, synthetic.focal_from_fov(side)]).contiguous()


def high_gain_state_dict(synthetic, gain):
    """The structured weights with the first hidden layer scaled up by ``gain`` and the second layer's weights scaled
    down by it: ReLU is positively homogeneous, so the network computes the same function in exact arithmetic, but its
    first hidden activations are ``gain`` times larger -- far beyond fp16's 65504 for gain = 1e5, while every weight
    stays finite in fp16.  The fp32 reference and bf16 operands (fp32's exponent range) are indifferent to it."""
    sd = {k: v.clone() for k, v in synthetic.synthetic_state_dict(0, "structured").items()}
    sd["layers_0.0.weight"] *= gain
    sd["layers_0.0.bias"] *= gain
    sd["layers_0.2.weight"] /= gain
    return sd


def test_fp16_overflow_is_never_silent(dev, oracle, synthetic):
    """Weights whose hidden activations exceed 65504: the default (fp16) render must not hand out NaN pixels silently.
    The first render with such weights warns, demotes the module to bf16 operands for these weights and returns the
    bf16 render; the raw C ABI call shows what would have happened (non-finite pixels + the sticky status bit)."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    lib = _lib.lib()
    rays = scene_rays(oracle, synthetic).to(dev)
    B, N = rays.shape[0], 64
    u = torch.rand(B, N, generator=torch.Generator().manual_seed(3)).to(dev)
    sd = high_gain_state_dict(synthetic, 1e5)
    with torch.no_grad():
        want = oracle.render_nerf(rays.cpu(), sd, N, u=u.cpu())
    assert torch.isfinite(want[0]).all()                   # the fp32 reference has no problem with these weights
    net = Nerf().to(dev)                                   # default precision: fp16
    assert net.precision == "fp16"
    net.load_state_dict(sd)
    # the kernel itself, through the C ABI: non-finite pixels and the status bit
    from nerf_simple_amd.utils.nets import packed_status
    packed = net.packed_weights(_lib.FP16)
    assert packed_status(packed, _lib.FP16) == 0
    rgb, disp, acc = torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    tb = torch.linspace(2, 6, N + 1).to(dev)
    _lib.check(lib.nerf_amd_render_forward(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tb), _lib.ptr(packed), _lib.FP16, 0, 0, 0,
                                           _lib.ptr(rgb), _lib.ptr(disp), None, _lib.ptr(acc), None, None, B, N,
                                           _lib.stream_ptr(dev)), "render")
    # The pixels are wrong but need not even be NaN: the kernel's integer ReLU turns a NaN with the sign bit set into 0,
    # and a layer whose rows are all NaN comes out as zeros -- finite garbage from there on.  The status word is set
    # all the same, from the accumulators (mlp_bf16_16.hip epilogue_piece).
    assert float((rgb.cpu() - want[0]).abs().max()) > 0.2 or not torch.isfinite(rgb).all()
    assert packed_status(packed, _lib.FP16) == _lib.STATUS_NONFINITE
    # bf16 operands on the same weights: the flag stays clear
    pb = net.packed_weights(_lib.BF16)
    _lib.check(lib.nerf_amd_render_forward(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tb), _lib.ptr(pb), _lib.BF16, 0, 0, 0,
                                           _lib.ptr(rgb), _lib.ptr(disp), None, _lib.ptr(acc), None, None, B, N,
                                           _lib.stream_ptr(dev)), "render")
    assert packed_status(pb, _lib.BF16) == 0 and torch.isfinite(rgb).all()
    # the host wrapper: warning + bf16 result, never the NaN pixels
    net2 = Nerf().to(dev)
    net2.load_state_dict(sd)
    with torch.no_grad():
        with pytest.warns(UserWarning, match="fp16 MFMA operands left their range"):
            got = render_nerf(rays, net2, N, u=u)
        bf = render_nerf(rays, net2, N, u=u, precision="bf16")
        with warnings.catch_warnings():
            warnings.simplefilter("error")                 # demoted already: no second warning, same bf16 kernels
            warnings.filterwarnings("ignore", message="input not in range")     # (the reference's own, on lego-scale points)
            again = render_nerf(rays, net2, N, u=u)
            out = net2(rays.new_zeros(8, 6) + 0.1)         # Nerf.forward is guarded by the same state
    for a, b, c in zip(got, bf, again):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)
        assert np.array_equal(a.cpu().numpy(), c.cpu().numpy(), equal_nan=True)
    # finite like the fp32 reference: everything but the disparity of rays that accumulate nothing (acc == 0 -> 0/0)
    for k in (0, 2, 3, 4):
        assert torch.isfinite(got[k]).all() and torch.isfinite(want[k]).all()
    assert torch.equal(torch.isnan(got[1]), got[3] == 0) and torch.equal(torch.isnan(want[1]), want[3] == 0)
    assert torch.isfinite(out).all()
    err = float((got[0].cpu() - want[0]).abs().max())
    print("high-gain weights, bf16 fallback: max |rgb err| vs the fp32 oracle", err)
    assert err <= 4.5e-2 * max(1.0, float(want[0].abs().max()))      # the bf16 tolerance of tests/test_gpu_parity.py
    # new weights: the guard starts over, and sane weights stay in fp16 without a warning
    net2.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        warnings.filterwarnings("ignore", message="input not in range")
        a = render_nerf(rays, net2, N, u=u)
        b = render_nerf(rays, net2, N, u=u, precision="fp16")
        c = render_nerf(rays, net2, N, u=u, precision="bf16")
    assert torch.equal(a[0], b[0]) and not torch.equal(a[0], c[0])


def test_fp16_weight_out_of_range_flagged_at_pack(dev, synthetic):
    """A single weight beyond 65504 (finite in fp32) is flagged by the packer itself."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    lib = _lib.lib()
    sd = {k: v.clone() for k, v in synthetic.synthetic_state_dict(0, "default").items()}
    sd["layers_1.0.weight"][3, 5] = 7.0e4
    net = Nerf().to(dev)
    net.load_state_dict(sd)
    from nerf_simple_amd.utils.nets import packed_status
    for code, want in ((_lib.FP16, _lib.STATUS_WEIGHT_RANGE), (_lib.BF16, 0)):
        assert packed_status(net.packed_weights(code), code) == want
    v = synthetic.points_in_scene(64, seed=2).to(dev)
    with torch.no_grad():
        with pytest.warns(UserWarning, match="a weight beyond 65504"):
            out = net(v)
        assert torch.equal(out, net(v, precision="bf16"))


def test_single_sample_matches_the_reference(dev, oracle, synthetic):
    """N = 1: the reference's delta construction leaves the sample axis empty (utils/rendering.py:60-61), so
    render_nerf returns rgb = 0, disparity = NaN, acc = 0 and alpha / w of shape [B,0], still drawing its
    torch.rand(B,1) -- checked against the oracle (whose torch ops do exactly that) for every precision, the
    standalone compositor, and the training path (all gradients exactly zero)."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf, volume_render
    rays = scene_rays(oracle, synthetic, 5)
    B = rays.shape[0]
    sd = synthetic.synthetic_state_dict(0, "structured")
    saved = torch.get_rng_state()
    try:
        torch.manual_seed(5)
        with torch.no_grad():
            want = oracle.render_nerf(rays, sd, 1)
        want_next = torch.rand(3)
        assert want[2].shape == (B, 0) and want[4].shape == (B, 0) and torch.isnan(want[1]).all()
        for precision in ("fp32", "fp16", "bf16"):
            net = Nerf(precision=precision).to(dev)
            net.load_state_dict(sd)
            torch.manual_seed(5)
            with torch.no_grad():
                got = render_nerf(rays.to(dev), net, 1)
            assert torch.equal(torch.rand(3), want_next), precision       # the same draw from the CPU generator
            for a, b in zip(got, want):
                assert a.shape == b.shape, precision
                assert np.array_equal(a.cpu().numpy(), b.numpy(), equal_nan=True), precision
    finally:
        torch.set_rng_state(saved)
    raw = torch.randn(B, 1, 4)
    ts = torch.rand(B, 1) + 2
    d = torch.randn(B, 3)
    want = oracle.volume_render(raw, ts, d)
    got = volume_render(raw.to(dev), ts.to(dev), d.to(dev))
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a.cpu().numpy(), b.numpy(), equal_nan=True)
    # training at N = 1: the reference's loss does not depend on the network, its gradients are exact zeros
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(sd)
    rgb = render_nerf(rays.to(dev), net, 1, u=torch.rand(B, 1).to(dev))[0]
    assert rgb.requires_grad
    rgb.pow(2).sum().backward()
    for k, p in net.named_parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0, k


def test_reference_checkpoint_renders_g5(dev, golden, oracle, synthetic, tmp_path):
    """N4 on the GPU: a .pth in the reference's format (torch.save(net.state_dict()), train.py:84-91) written by the
    reference-side recipe, loaded through utils/checkpoint.load_checkpoint (strict, weights_only), renders golden G5
    (the 100x100x32 image) with the fp32 kernel."""
    from nerf_simple_amd.utils import checkpoint
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_poses
    g = golden("image_structured.npz")
    path = str(tmp_path / "1666742866.6157136.pth")
    torch.save({k: v.clone() for k, v in synthetic.synthetic_state_dict(0, "structured").items()}, path)
    net = checkpoint.load_checkpoint(Nerf(precision="fp32").to(dev), path)
    assert next(net.parameters()).is_cuda
    u = t(golden("image_u.npz")["u"]).to(dev)
    pose = t(g["pose"])
    rgb, disp = render_poses(net, [pose], [100, 100, float(g["f"])], int(g["batch_size"]), N=32, u=u)
    assert np.abs(rgb[0].reshape(-1, 3) - g["rgb"]).max() <= 1e-4
    assert np.abs(disp[0].reshape(-1) - g["disp"]).max() <= 1e-4 * max(1.0, float(np.abs(g["disp"]).max()))
    # and back: a checkpoint written here holds the reference's 24 keys with the trained values
    out = checkpoint.save_checkpoint(net, str(tmp_path / "out.pth"))
    sd = torch.load(out, weights_only=True)
    assert list(sd.keys()) == [k for k, _ in synthetic.PARAM_SPECS]
    assert all(torch.equal(sd[k], v) for k, v in synthetic.synthetic_state_dict(0, "structured").items())


def test_foreign_net_path_is_the_same_arithmetic(dev, golden, synthetic):
    """render_nerf with an arbitrary ``net`` object (utils/rendering.py:41 calls net.forward on whatever it is given):
    sampling and query points come from nerf_amd_query_points, the net runs as given, nerf_amd_volume_render_rays
    composites.  With the fp32 kernel behind the foreign object the result equals the fused one-launch render bit
    for bit -- explicit jitter and the counter RNG alike -- and gradients reach a foreign torch module."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    g = golden("render_structured.npz")
    net = Nerf(precision="fp32").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
    seen = {}

    class Wrapped:
        def forward(self, q):
            seen["shape"] = tuple(q.shape)
            return net.forward_inference(q)

    rays = t(g["rays"]).to(dev)
    u = t(g["N64_u"]).to(dev)
    with torch.no_grad():
        for kw in (dict(u=u), dict(device_rng=True, seed=11, ray_id0=1000)):
            a = render_nerf(rays, Wrapped(), 64, **kw)
            b = render_nerf(rays, net, 64, **kw)
            assert seen["shape"] == (rays.shape[0] * 64, 6)
            for x, y in zip(a, b):
                assert torch.equal(x, y)
    tiny = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4)).to(dev)
    rgb = render_nerf(rays[:32], tiny, 16, u=torch.rand(32, 16, device=dev))[0]
    rgb.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in tiny.parameters())
    assert float(tiny[0].weight.grad.abs().max()) > 0


TOL_ODD = {"fp32": 1e-4, "fp16": 7e-3, "bf16": 4.5e-2}       # the (precision, structured) tolerances of tests/test_gpu_parity.py


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_render_ragged_shapes_vs_oracle(dev, oracle, synthetic, precision):
    """render_nerf against the oracle (not against another kernel path) at shapes nothing else is sized like: a single
    ray, two samples, rays that straddle tiles (N = 65, 300), the longest ray of the one-launch path (768), the first
    one of the two-launch path (769), near and far planes other than the defaults -- all five outputs -- and the empty
    batch."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    sd = synthetic.synthetic_state_dict(0, "structured")
    net = Nerf(precision=precision).to(dev)
    net.load_state_dict(sd)
    table = scene_rays(oracle, synthetic, 32)
    tol = TOL_ODD[precision]
    gen = torch.Generator().manual_seed(77)
    for B, N, tn, tf in ((1, 2, 2, 6), (3, 5, 2, 6), (37, 65, 2, 6), (5, 300, 2, 6), (2, 768, 2, 6), (3, 769, 2, 6),
                         (1000, 31, 2, 6), (64, 64, 0.5, 9.5)):
        rays = table[torch.randperm(table.shape[0], generator=gen)[:B]].contiguous()
        u = torch.rand(B, N, generator=gen)
        with torch.no_grad():
            want = oracle.render_nerf(rays, sd, N, tn, tf, u=u)
            got = render_nerf(rays.to(dev), net, N, tn, tf, u=u.to(dev))
        for name, a, b in zip(("rgb", "disp", "alpha", "acc", "w"), got, want):
            assert a.shape == b.shape, (B, N, name)
            a, b = a.cpu().numpy(), b.numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b)), (B, N, name)
            ok = ~np.isnan(b)
            err = np.abs(a[ok] - b[ok]).max() / max(1.0, np.abs(b[ok]).max()) if ok.any() else 0.0
            assert err <= tol, (precision, B, N, tn, tf, name, err)
    # B = 0: the reference returns empty tensors of the right shapes (every op of utils/rendering.py:13-85 accepts
    # an empty batch) and still "draws" torch.rand(0, N): nothing
    empty = torch.empty(0, 6)
    with torch.no_grad():
        want = oracle.render_nerf(empty, sd, 8)
        got = render_nerf(empty.to(dev), net, 8)
    for a, b in zip(got, want):
        assert tuple(a.shape) == tuple(b.shape)


def test_integration_stub_from_the_document(dev, golden, synthetic):
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would add (pack + render_nerf over the
    C ABI, keeping the reference's own Nerf for the weights).  This test runs THAT text -- the code block as printed, with
    only the library's path filled in -- on a plain nn.Module holding the reference's 24 parameters, against golden G4."""
    import os
    import re
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# utils/_nerf_amd\.py.*?)```", text, flags=re.S).group(1)
    block = block.replace('ctypes.CDLL("libnerf_amd.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md:stub", "exec"), ns)
    g = golden("render_structured.npz")
    net = Nerf().to(dev)                                   # stands in for the reference's module: same parameters, same order
    net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
    packed = ns["pack"](net)
    rays = t(g["rays"]).to(dev)
    saved = torch.get_rng_state()
    try:
        torch.manual_seed(int(g["N64_seed"]))              # the stub draws torch.rand(B, N) on the CPU like the reference
        out = ns["render_nerf"](rays, packed, 64)
    finally:
        torch.set_rng_state(saved)
    torch.cuda.synchronize()
    assert not ns["out_of_range"](packed)
    for name, o in zip(("rgb", "disp", "alpha", "acc", "w"), out):
        want = g[f"N64_{name}"]
        err = np.abs(o.cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
        assert err <= 7e-3, (name, err)                    # the fp16 tolerance of tests/test_gpu_parity.py


def test_composite_special_values_like_the_reference(dev, oracle):
    """volume_render on values a diverging or untrained network can emit -- sigma at and around the softplus threshold,
    huge, infinite, NaN; colours huge, infinite, NaN; coincident sample positions (delta = 0) -- against the oracle's
    torch ops: NaNs in the same places (torch.max's NaN propagation in the disparity, 0 * inf in the products), finite
    values to the compositor's tolerance, infinities equal."""
    from nerf_simple_amd.utils.rendering import volume_render
    sig_vals = torch.tensor([0.0, 1e-8, -1e-8, 1.0, -1.0, 19.9, 20.0, 20.1, -19.9, 100.0, -100.0, 1e4, -1e4, float("inf"),
                             float("-inf"), float("nan")])
    col_vals = torch.tensor([0.0, 1.0, -1.0, 0.25, 1e30, -1e30, float("inf"), float("nan")])
    gen = torch.Generator().manual_seed(31)
    for B, N in ((64, 16), (32, 64), (16, 130)):
        raw = torch.empty(B, N, 4)
        raw[..., 3] = sig_vals[torch.randint(0, len(sig_vals), (B, N), generator=gen)]
        raw[..., :3] = col_vals[torch.randint(0, len(col_vals), (B, N, 3), generator=gen)]
        # most entries ordinary, so that the special ones meet ordinary neighbours
        plain = torch.rand(B, N, generator=gen) < 0.7
        raw[plain] = torch.randn(int(plain.sum()), 4, generator=gen)
        ts = torch.sort(torch.rand(B, N, generator=gen) * 4 + 2, dim=1).values
        ts[:, 3] = ts[:, 2]                                  # a coincident pair: delta = 0
        d = torch.randn(B, 3, generator=gen)
        want = oracle.volume_render(raw, ts, d)
        got = volume_render(raw.to(dev), ts.to(dev), d.to(dev))
        for name, a, b in zip(("rgb", "disp", "alpha", "acc", "w"), got, want):
            a, b = a.cpu().numpy(), b.numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b)), (B, N, name, int(np.isnan(a).sum()), int(np.isnan(b).sum()))
            inf = np.isinf(b)
            assert np.array_equal(a[inf], b[inf]), (B, N, name)
            fin = np.isfinite(b)
            assert np.isfinite(a[fin]).all(), (B, N, name)
            if fin.any():
                # sums of terms up to 1e30 cancel: the bound is relative to the largest finite term a ray can hold
                scale = max(1.0, float(np.abs(b[fin]).max()))
                assert np.abs(a[fin] - b[fin]).max() <= 1e-4 * scale, (B, N, name, np.abs(a[fin] - b[fin]).max(), scale)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_non_finite_inputs_propagate_like_the_reference(dev, oracle, synthetic, precision):
    """Garbage in: a ray with a NaN origin and one with an infinite direction.  In the reference every sample of such a
    ray is NaN through the network (torch's ReLU keeps NaN), its outputs are NaN, and every other ray is untouched.  The
    fp32 kernel does the same; the 16-bit kernels' integer ReLU would not (it zeroes a NaN whose sign bit is set), so
    their range guard sends these weights to the fp32 kernel, with a warning -- the result has the reference's NaNs."""
    import warnings as _w
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    sd = synthetic.synthetic_state_dict(0, "structured")
    rays = scene_rays(oracle, synthetic, 6).clone()
    rays[5, 1] = float("nan")
    rays[17, 4] = float("inf")
    N = 32
    u = torch.rand(rays.shape[0], N, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = oracle.render_nerf(rays, sd, N, u=u)
    bad = torch.isnan(want[0]).any(dim=1)
    assert bad[5] and bad[17] and int(bad.sum()) == 2
    net = Nerf(precision=precision).to(dev)
    net.load_state_dict(sd)
    with torch.no_grad(), _w.catch_warnings(record=True) as caught:
        _w.simplefilter("always")
        got = render_nerf(rays.to(dev), net, N, u=u.to(dev))
    if precision != "fp32":
        assert any("non-finite" in str(w.message) for w in caught), [str(w.message) for w in caught]
    for name, a, b in zip(("rgb", "disp", "alpha", "acc", "w"), got, want):
        a, b = a.cpu().numpy(), b.numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b)), (precision, name)
        ok = ~np.isnan(b)
        assert np.abs(a[ok] - b[ok]).max() <= 1e-4 * max(1.0, np.abs(b[ok]).max()), (precision, name)   # it IS the fp32 kernel by now


@pytest.mark.parametrize("precision", ["fp16", "bf16", "fp32"])
def test_non_finite_weights_propagate_like_the_reference(dev, oracle, synthetic, precision):
    """A checkpoint with a NaN in it (a diverged run saved to disk): one weight of layers_0.2, and -- separately -- an
    infinite bias of layers_1.0.  In the reference the NaN spreads to every output of every ray (NaN * h in one row, then
    every row of the next layer).  A 16-bit kernel's integer ReLU could wash that row to zero and render a plausible
    picture from a broken network; the packer flags non-finite weights, so these weights render with the fp32 kernel
    (with a warning) and come out as the reference's do."""
    import warnings as _w
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    rays = scene_rays(oracle, synthetic, 6)
    N = 16
    u = torch.rand(rays.shape[0], N, generator=torch.Generator().manual_seed(3))
    for key, index, value in (("layers_0.2.weight", (209, 112), float("nan")), ("layers_1.0.bias", (7,), float("inf"))):
        sd = {k: v.clone() for k, v in synthetic.synthetic_state_dict(0, "structured").items()}
        sd[key][index] = value
        with torch.no_grad():
            want = oracle.render_nerf(rays, sd, N, u=u)
        net = Nerf(precision=precision).to(dev)
        net.load_state_dict(sd)
        with torch.no_grad(), _w.catch_warnings(record=True) as caught:
            _w.simplefilter("always")
            got = render_nerf(rays.to(dev), net, N, u=u.to(dev))
        if precision != "fp32":
            assert any("not finite" in str(w.message) or "non-finite" in str(w.message) for w in caught), [str(w.message) for w in caught]
        for name, a, b in zip(("rgb", "disp", "alpha", "acc", "w"), got, want):
            a, b = a.cpu().numpy(), b.numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b)), (precision, key, name, int(np.isnan(a).sum()), int(np.isnan(b).sum()))
            ok = np.isfinite(b)
            if ok.any():
                assert np.abs(a[ok] - b[ok]).max() <= 1e-4 * max(1.0, np.abs(b[ok]).max()), (precision, key, name)


def test_render_beyond_two_to_the_31_samples(dev, synthetic):
    """A batch whose sample count does not fit 32 bits: 17,000,000 rays x 128 samples = 2.176e9 ray-samples in ONE call
    (the C ABI's sizes are int64).  Every index on the path -- point ids, the counter RNG's sample counter, per-workgroup
    ray ranges, output offsets -- has to be 64-bit where it can grow: the last rays of the batch, and rays around the
    2^31-sample mark, must come out exactly as when they are rendered on their own with the matching ray-id offset."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
    packed = net.packed_weights(_lib.BF16)
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    base = camera_rays([pose], [1000, 1000, synthetic.focal_from_fov(1000)]).to(dev)          # 1 M distinct rays
    B, N = 17_000_000, 128
    assert B * N > 2 ** 31
    rays = base.repeat(17, 1).contiguous()
    tb = torch.linspace(2, 6, N + 1).to(dev)
    px = torch.empty((B, 4), dtype=torch.float32, device=dev)
    st = _lib.stream_ptr(dev)
    _lib.check(lib.nerf_amd_render_pixels_forward(_lib.ptr(rays), None, _lib.ptr(tb), _lib.ptr(packed), _lib.BF16,
                                                  _lib.FLAG_DEVICE_RNG, 99, 0, _lib.ptr(px), None, B, N, st), "render")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(px[:, :3]).all())
    mark = 2 ** 31 // N                                         # the ray whose first sample is sample 2^31
    for lo in (0, mark - 500, mark + 1, B - 1000):
        part = rays[lo:lo + 1000].contiguous()
        small = torch.empty((1000, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.nerf_amd_render_pixels_forward(_lib.ptr(part), None, _lib.ptr(tb), _lib.ptr(packed), _lib.BF16,
                                                      _lib.FLAG_DEVICE_RNG, 99, lo, _lib.ptr(small), None, 1000, N, st), "render")
        torch.cuda.synchronize()
        a, b = px[lo:lo + 1000], small
        assert bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all()), lo
    # the same scene ray under another ray id draws other jitter: the 17 copies of the table are not copies of pixels
    assert not torch.equal(px[:1000, :3], px[1_000_000:1_001_000, :3])


def test_integration_training_stub_from_the_document(dev, golden, synthetic):
    """INTEGRATION.md's second code block -- the training step of train.py:47-57 unrolled over the C ABI -- run as
    printed on golden G6: loss and every gradient norm the reference's autograd produced, within the bf16 training
    bounds of tests/test_gpu_training.py."""
    import os
    import re
    import sys
    import types
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    first = re.search(r"```python\n(# utils/_nerf_amd\.py.*?)```", text, flags=re.S).group(1)
    second = re.search(r"```python\n(# utils/_nerf_amd_train\.py.*?)```", text, flags=re.S).group(1)
    first = first.replace('ctypes.CDLL("libnerf_amd.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    base = types.ModuleType("utils._nerf_amd")
    exec(compile(first, "INTEGRATION.md:stub", "exec"), base.__dict__)
    pkg = types.ModuleType("utils")
    saved = {k: sys.modules.get(k) for k in ("utils", "utils._nerf_amd")}
    sys.modules["utils"], sys.modules["utils._nerf_amd"] = pkg, base
    try:
        ns = {}
        exec(compile(second, "INTEGRATION.md:train stub", "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    g = golden("train.npz")
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    images = ns["pack_for_training"](net)
    saved_rng = torch.get_rng_state()
    try:
        torch.manual_seed(66)                                 # G6's seed: the stub draws torch.rand(B, N) like the reference
        loss, grads = ns["loss_and_gradients"](t(g["rays"]).to(dev), t(g["gt"]).to(dev), images, int(g["N"]))
    finally:
        torch.set_rng_state(saved_rng)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss"])) <= 1e-3 * float(g["loss"])
    off = 0
    for k, shape in synthetic.PARAM_SPECS:
        n = int(np.prod(shape))
        got = grads[off:off + n].cpu().numpy()
        off += n
        assert abs(np.linalg.norm(got) / float(g[f"gnorm/{k}"]) - 1) <= 8.4e-2, k      # rel_l2_bound("default", 4096)


def test_range_warning_is_the_references(dev, synthetic, oracle):
    """The reference warns ``input not in range -1,1, check rescaling`` (UserWarning, utils/xyz.py:8-9) whenever a
    coordinate of a query point leaves [-1, 1] -- at the price of two device->host syncs per gamma call.  Here the verdict
    is formed on the device (nerf_amd_range_check: first and last sample of every ray) and raised lazily; it must be the
    reference's verdict, case by case: 24 random ray bundles whose extent straddles the unit cube (computed from the
    oracle's query points with the reference's own condition), jitter given / positions given in arbitrary order / the
    stand-alone encoder and Nerf.forward on the points themselves; and the pixels do not depend on it."""
    import warnings
    from nerf_simple_amd.utils import xyz
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    net = Nerf(precision="fp32").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    gen = torch.Generator().manual_seed(77)
    B, N = 8, 16

    def verdict(fn):
        xyz.flush_range_warning()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = fn()
            xyz.flush_range_warning()
        hits = [x for x in w if str(x.message) == xyz.RANGE_WARNING]
        assert all(issubclass(x.category, UserWarning) for x in hits)      # (the reference raises one per gamma call: up to six)
        return bool(hits), out

    seen = {True: 0, False: 0}
    for case in range(24):
        scale = float(torch.empty(1).uniform_(0.1, 0.8, generator=gen))            # the bundle reaches +-2 scale around the centre
        o = (torch.rand(B, 3, generator=gen) - 0.5) * 0.4
        d = torch.nn.functional.normalize(torch.randn(B, 3, generator=gen), dim=1) * scale
        rays = torch.cat([o - 4 * d, d], dim=1).contiguous()              # centred on the cube at t = 4
        u = torch.rand(B, N, generator=gen)
        ts = oracle.sample_ts(u)
        if case % 3 == 2:
            ts = ts[:, torch.randperm(N, generator=gen)].contiguous()      # explicit positions, not sorted
        q, _ = oracle.query_points(rays, ts)
        want = bool(torch.any(q < -1) or torch.any(q > 1))                 # the reference's condition on all six columns
        seen[want] += 1
        kw = {"ts": ts.to(dev)} if case % 3 == 2 else {"u": u.to(dev)}
        got, out = verdict(lambda: render_nerf(rays.to(dev), net, N, **kw))
        assert got == want, (case, scale, want)
        if case < 6:
            qd = q.to(dev)
            assert verdict(lambda: net.forward(qd))[0] == want, case
            assert verdict(lambda: xyz.positional_encoder(qd))[0] == want, case
            col = qd[:, 1:2].contiguous()
            assert verdict(lambda: xyz.gamma(col, 3))[0] == bool(torch.any(col < -1) or torch.any(col > 1)), case
    assert seen[True] >= 5 and seen[False] >= 5, seen
    # lego-scale cameras always warn (every sample sits outside the unit cube); the render itself is unchanged by the check
    pose = torch.from_numpy(oracle.spherical_to_pose(4, -30, 0)).float()
    rays = oracle.camera_rays(pose, [8, 8, synthetic.focal_from_fov(8)]).to(dev)
    u = torch.rand(64, 32, generator=gen).to(dev)
    got, a = verdict(lambda: render_nerf(rays, net, 32, u=u))
    assert got
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        b = render_nerf(rays, net, 32, u=u)
        xyz.flush_range_warning()
    assert all(torch.equal(x, y) for x, y in zip(a, b))
