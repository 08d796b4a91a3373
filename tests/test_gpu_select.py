"""GPU parity of the ray selection (reference train.py:47-49, utils/dataload.py:141-153).

``rg.select(mode, N)`` is ``ray_ids = torch.randperm(n)[:N]; rays = data[ray_ids]`` on torch's CPU generator, and the
training loop gathers ``train_imgs[ray_ids]`` next.  The device path (csrc/select.hip + csrc/host_rng.hip) must give the
SAME ids -- bit for bit, they are integers -- the same rows, and leave torch's generator in the same state, for every
table size from 1 to the reference's 16 M rays, from any stream position; the counter-RNG form must equal the oracle's
restatement of it.  Checker: torch.randperm itself (CPU), oracle.randperm_prefix / oracle.select_ids_counter.
"""
import ctypes

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


def synthetic_tables(n, dev):
    """rays[i] and colours[i] are functions of i, so a gathered row names its source."""
    i = torch.arange(n, device=dev, dtype=torch.float32)
    rays = torch.stack([i, i * 0.5, -i, i + 0.25, 1.0 / (i + 1.0), torch.ones_like(i)], dim=1).contiguous()
    cols = torch.stack([i * 2.0, i + 7.0, -i * 0.125], dim=1).contiguous()
    return rays, cols


def test_raw_draws_and_advance_are_the_generators(dev, oracle):
    """nerf_amd_mt19937_raw = the generator's 32-bit outputs; nerf_amd_mt19937_advance(q) = its 624 state words 1 + q
    blocks on -- every bit of every word, also from a freshly seeded state (whose word 0 holds 31 bits that are not
    part of the generator's state: the jump starts one block later for exactly that reason)."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils import host_rng as H
    lib = _lib.lib()
    for seed, pre in ((0, 0), (3, 100), (9, 624), (11, 1000)):
        g = torch.Generator()
        g.manual_seed(seed)
        if pre:
            torch.rand(pre, generator=g)
        left, _, nxt, words = H._parse(g.get_state())
        w32 = words.astype(np.uint32)
        words_dev = torch.from_numpy(w32.view(np.int32)).to(dev)
        for n in (1, 7, 624, 5000):
            out = torch.empty(n, dtype=torch.int32, device=dev)
            state_out = torch.empty(624, dtype=torch.int32, device=dev)
            _lib.check(lib.nerf_amd_mt19937_raw(_lib.ptr(words_dev), 625 - left, _lib.ptr(out), n, _lib.ptr(state_out),
                                                _lib.stream_ptr(dev)), "raw")
            want, mt = oracle.mt19937_uniform(w32, 625 - left, n, raw=True)
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want), (seed, pre, n)
            assert np.array_equal(state_out.cpu().numpy().view(np.uint32), mt), (seed, pre, n)
        for q in (0, 1, 5, 40, 1000):
            state_out = torch.empty(624, dtype=torch.int32, device=dev)
            _lib.check(lib.nerf_amd_mt19937_advance(_lib.ptr(words_dev), _lib.ptr(H.advance_poly(q, dev)), _lib.ptr(state_out),
                                                    _lib.stream_ptr(dev)), "advance")
            _, mt = oracle.mt19937_uniform(w32, 624, 624 * q + 1, raw=True)          # one draw into block 1 + q
            assert np.array_equal(state_out.cpu().numpy().view(np.uint32), mt), (seed, pre, q)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 100, 625, 8192, 30_000, 31_000, 200_003, 16_000_000])
def test_select_is_torch_randperm(dev, n):
    """``rg.select`` / ``select_batch`` against the reference's statements run on the CPU: ids, rays, colours, and the
    generator afterwards (its whole state image), from a fresh seed and from mid-stream positions, prefixes and full
    permutations.  n = 16 M is the reference's table (25 images of 800 x 800); 30,000 / 31,000 straddle the switch from
    drawing every number to jumping over the unused ones."""
    from nerf_simple_amd.utils.dataload import RayGenerator
    rays_tab, cols_tab = synthetic_tables(n, dev)
    rg = RayGenerator({"train": rays_tab}, {"train": cols_tab})
    rays_host, cols_host = (rays_tab.cpu(), cols_tab.cpu()) if n <= 200_003 else (None, None)
    saved = torch.get_rng_state()
    try:
        cases = [(5, 0, 4096), (6, 333, 64), (7, 624, min(n, 4096))] if n > 100 else [(5, 0, 4096), (6, 333, n), (7, 7, 1), (8, 1, 3)]
        for seed, pre, B in cases:
            torch.manual_seed(seed)
            if pre:
                torch.rand(pre)
            st = torch.get_rng_state()
            want = torch.randperm(n)[:B]                        # dataload.py:151
            after = torch.get_rng_state()
            follow = torch.rand(5)
            torch.set_rng_state(st)
            rays, gt, ids = rg.select_batch("train", B)
            assert ids.dtype == torch.int64 and ids.shape == (min(B, n),)
            assert torch.equal(ids.cpu(), want), (n, seed, pre, B)
            assert torch.equal(torch.get_rng_state(), after), (n, seed, pre, B)
            assert torch.equal(torch.rand(5), follow)
            if rays_host is not None:
                assert torch.equal(rays.cpu(), rays_host[want]) and torch.equal(gt.cpu(), cols_host[want])
            else:
                assert torch.equal(rays, rays_tab[ids]) and torch.equal(gt, cols_tab[ids])
                assert torch.equal(rays[:, 0].cpu(), want.float())                    # row i of the table starts with i
            # the reference's two-value form
            torch.set_rng_state(st)
            rays2, ids2 = rg.select(mode="train", N=B)
            assert torch.equal(ids2, ids) and torch.equal(rays2, rays)
    finally:
        torch.set_rng_state(saved)


def test_select_then_jitter_is_one_stream(dev):
    """The reference's iteration draws randperm(n) and then torch.rand(B, N) from the same generator (train.py:47-51):
    one GeneratorSession gives both, and the generator ends where the reference's two calls leave it."""
    from nerf_simple_amd.utils import host_rng as H
    from nerf_simple_amd.utils.dataload import RayGenerator
    n, B, N = 1_000_000, 4096, 128
    rays_tab, cols_tab = synthetic_tables(n, dev)
    rg = RayGenerator({"train": rays_tab}, {"train": cols_tab})
    saved = torch.get_rng_state()
    try:
        torch.manual_seed(21)
        for it in range(4):
            st = torch.get_rng_state()
            want_ids, want_u = torch.randperm(n)[:B], torch.rand(B, N)
            after = torch.get_rng_state()
            torch.set_rng_state(st)
            for fused in (False, True):          # two dependent jumps (shuffle, jitter segments) / one with summed distances
                torch.set_rng_state(st)
                session = H.GeneratorSession(dev)
                out = (torch.empty((B, 6), device=dev), torch.empty((B, 3), device=dev), torch.empty(B, dtype=torch.int64, device=dev))
                if fused:
                    buf = torch.full((B, N), -1.0, device=dev)
                    u = rg.select_from_session(session, "train", B, *out, jitter=(B, N, buf))
                    assert u is buf
                else:
                    rg.select_from_session(session, "train", B, *out)
                    u = session.rand(B, N)
                session.finish()
                assert torch.equal(out[2].cpu(), want_ids) and torch.equal(u.cpu(), want_u), (it, fused)
                assert torch.equal(torch.get_rng_state(), after), (it, fused)
    finally:
        torch.set_rng_state(saved)


def test_counter_select_equals_the_oracle(dev, oracle):
    """device_rng=True: the permutation prefix from the counter RNG, against its numpy restatement; the seed offset in
    device memory (what a captured graph replays with) gives the batch of seed + offset."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.dataload import RayGenerator
    for n, B in ((1, 1), (7, 7), (100, 100), (5000, 4096), (1_000_003, 4096), (16_000_000, 4096)):
        rays_tab, cols_tab = synthetic_tables(n, dev)
        rg = RayGenerator({"train": rays_tab}, {"train": cols_tab})
        rays, gt, ids = rg.select_batch("train", B, device_rng=True, seed=77)
        want = oracle.select_ids_counter(n, B, 77)
        assert np.array_equal(ids.cpu().numpy(), want), (n, B)
        assert torch.equal(rays, rays_tab[ids]) and torch.equal(gt, cols_tab[ids])
        assert len(set(want.tolist())) == B
        off = torch.tensor([5], dtype=torch.int64, device=dev)
        out = (torch.empty_like(rays), torch.empty_like(gt), torch.empty_like(ids))
        rg.launch("train", B, None, 77, ctypes.c_void_p(off.data_ptr()), *out)
        assert np.array_equal(out[2].cpu().numpy(), oracle.select_ids_counter(n, B, 77, 5)), (n, B)


def test_select_abi_partial_outputs(dev):
    """Any of the three outputs may be NULL (C ABI); the library writes nothing else."""
    from nerf_simple_amd import _lib
    lib = _lib.lib()
    n, B = 1000, 64
    rays_tab, cols_tab = synthetic_tables(n, dev)
    ws = torch.empty(int(lib.nerf_amd_select_workspace_bytes(B)), dtype=torch.uint8, device=dev)
    ids = torch.full((B + 2,), -7, dtype=torch.int64, device=dev)
    rays = torch.full((B + 1, 6), -7.0, device=dev)
    st = _lib.stream_ptr(dev)
    _lib.check(lib.nerf_amd_select_rays(None, 9, None, n, B, None, None, None, None, _lib.ptr(ids), _lib.ptr(ws), st), "ids only")
    _lib.check(lib.nerf_amd_select_rays(None, 9, None, n, B, _lib.ptr(rays_tab), None, _lib.ptr(rays), None, None, _lib.ptr(ws), st),
               "rays only")
    assert int(ids[B]) == -7 and int(ids[B + 1]) == -7 and float(rays[B].abs().max()) == 7.0
    assert torch.equal(rays[:B], rays_tab[ids[:B]])


def test_graphed_step_selects_its_own_rays(dev, oracle, synthetic, golden):
    """GraphedTrainStep(rays_from=rg): (a) device_rng=True -- the selection is the first node of graph A, keyed by the
    step counter in device memory: every replay trains on the oracle's batch for (seed, step), and equals the same step
    fed by hand; (b) reference stream -- the ids are torch.randperm's, the jitter follows in the same stream, and the
    generator ends where the reference's iteration leaves it."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    from nerf_simple_amd.utils.dataload import RayGenerator
    d = golden("dataset.npz")
    hw = int(d["hw"])
    rays_tab = torch.cat([oracle.camera_rays(torch.from_numpy(oracle.spherical_to_pose(4, -30, float(phi))).float(),
                                             [hw, hw, synthetic.focal_from_fov(hw)]) for phi in d["views"]]).contiguous()
    gt_tab = torch.from_numpy(np.ascontiguousarray(d["gt"]))
    n, B, N = rays_tab.shape[0], 256, 64
    rg = RayGenerator.from_tables(rays_tab, gt_tab, device=dev)

    def fresh():
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        return net, FusedAdam(net, lr=5e-4)

    # (a) counter RNG, selection inside the graph
    net_a, opt_a = fresh()
    auto = GraphedTrainStep(net_a, opt_a, B, N, device_rng=True, seed=11, rays_from=rg)
    net_b, opt_b = fresh()
    hand = GraphedTrainStep(net_b, opt_b, B, N, device_rng=True, seed=11)
    for step in (1, 2, 3):
        la = float(auto.step())
        want = torch.from_numpy(oracle.select_ids_counter(n, B, 11, step))
        assert torch.equal(auto.ray_ids.cpu(), want), step
        lb = float(hand.step(rays_tab[want].to(dev), gt_tab[want].to(dev)))
        # the same kernels on the same batch; from the second step on the weights carry the order of the gradient atomics
        assert la == lb if step == 1 else abs(la - lb) <= 1e-4 * abs(lb), (step, la, lb)
    assert float((opt_a.flat - opt_b.flat).abs().max()) <= 3 * 5e-4
    with pytest.raises(RuntimeError):
        auto.step(rays_tab[:B].to(dev), gt_tab[:B].to(dev))
    # (b) the reference's stream
    saved = torch.get_rng_state()
    try:
        net_c, opt_c = fresh()
        ref = GraphedTrainStep(net_c, opt_c, B, N, rays_from=rg)
        net_d, opt_d = fresh()
        byhand = GraphedTrainStep(net_d, opt_d, B, N)
        torch.manual_seed(5)
        for step in range(3):
            st = torch.get_rng_state()
            ids, u = torch.randperm(n)[:B], torch.rand(B, N)
            after = torch.get_rng_state()
            torch.set_rng_state(st)
            lc = float(ref.step())
            assert torch.equal(ref.ray_ids.cpu(), ids), step
            assert torch.equal(torch.get_rng_state(), after), step
            ld = float(byhand.step(rays_tab[ids].to(dev), gt_tab[ids].to(dev), u=u.to(dev)))
            assert lc == ld if step == 0 else abs(lc - ld) <= 1e-4 * abs(ld), (step, lc, ld)
        assert float((opt_c.flat - opt_d.flat).abs().max()) <= 3 * 5e-4
    finally:
        torch.set_rng_state(saved)


def test_graphed_step_survives_a_checkpoint_restore(dev, synthetic, golden):
    """load_state_dict between graphed steps (restoring a checkpoint after a diverged run) moves the parameters'
    versions: the module's cache must not replace the two images whose addresses are baked into the graphs -- the step
    re-packs into them (advisor finding, round 3).  After the restore the run continues exactly like a fresh one."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    g = golden("train.npz")
    rays, gt, u = (torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in ("rays", "gt", "u"))
    N = int(g["N"])
    sd0 = synthetic.synthetic_state_dict(0, "default")

    def run(restore):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(sd0)
        opt = FusedAdam(net, lr=5e-3)
        stepper = GraphedTrainStep(net, opt, rays.shape[0], N, check_every=1)
        fwd, bwd = stepper._packed_fwd.data_ptr(), stepper._packed_bwd.data_ptr()
        losses = [float(stepper.step(rays, gt, u=u)) for _ in range(2)]
        if restore:
            net.load_state_dict(sd0)                         # in place into the flat vector: versions move
            net.packed_weights(_lib.BF16)                    # a render in between re-packs into a NEW buffer of the cache
            opt.exp_avg.zero_()
            opt.exp_avg_sq.zero_()
            opt.step_count = 0
            losses = []
        losses += [float(stepper.step(rays, gt, u=u)) for _ in range(3)]
        assert (stepper._packed_fwd.data_ptr(), stepper._packed_bwd.data_ptr()) == (fwd, bwd)
        assert net._packed[(dev, _lib.BF16)].buf.data_ptr() == fwd
        return losses[:3]

    l_fresh, l_rest = run(False), run(True)
    assert l_rest[0] == l_fresh[0], (l_rest, l_fresh)                     # the restored weights, bit for bit
    assert all(abs(a - b) <= 1e-4 * abs(b) for a, b in zip(l_rest, l_fresh)), (l_rest, l_fresh)


def test_integration_select_stub_from_the_document(dev, oracle):
    """INTEGRATION.md's third code block -- train.py:47-49 over the C ABI (counter-RNG form) -- run as printed, on top of
    the first block (the library handle): B distinct rows, the oracle's ids for (seed, offset), rows gathered from both tables."""
    import os
    import re
    import sys
    import types
    from nerf_simple_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    first = re.search(r"```python\n(# utils/_nerf_amd\.py.*?)```", text, flags=re.S).group(1)
    third = re.search(r"```python\n(# utils/_nerf_amd_select\.py.*?)```", text, flags=re.S).group(1)
    first = first.replace('ctypes.CDLL("libnerf_amd.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    base = types.ModuleType("utils._nerf_amd")
    exec(compile(first, "INTEGRATION.md:stub", "exec"), base.__dict__)
    saved = {k: sys.modules.get(k) for k in ("utils", "utils._nerf_amd")}
    sys.modules["utils"], sys.modules["utils._nerf_amd"] = types.ModuleType("utils"), base
    try:
        ns = {}
        exec(compile(third, "INTEGRATION.md:select stub", "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    n, B = 100_000, 512
    table, colours = synthetic_tables(n, dev)
    rays, gt, ids = ns["select"](table, colours, B, 5)
    assert np.array_equal(ids.cpu().numpy(), oracle.select_ids_counter(n, B, 5))
    assert torch.equal(rays, table[ids]) and torch.equal(gt, colours[ids]) and len(set(ids.tolist())) == B
    word = torch.tensor([9], dtype=torch.int64, device=dev)
    _, _, ids9 = ns["select"](table, colours, B, 5, step_word=word)
    assert np.array_equal(ids9.cpu().numpy(), oracle.select_ids_counter(n, B, 5, 9))


def test_ray_generator_from_samples(dev, oracle, synthetic):
    """RayGenerator.from_samples takes what the reference's load_data returns -- samples[mode] = [{'img': HxWx3 float64
    array in [0,1], 'transform': 4x4 pose}], cam_params = [H, W, f] -- and builds rays_dataset (utils/dataload.py:114-129:
    images in order, pixels row-major, [origin, R dir]) and the colour table (train.py:33 + .float()) on the device: origins
    exact, directions within the fma rounding of the 3-term rotation (as test_generate_rays_golden), colours exact; a
    select on it returns rows of those tables."""
    from nerf_simple_amd.utils.dataload import RayGenerator
    H, W = 12, 10
    f = float(synthetic.focal_from_fov(W))
    rng = np.random.default_rng(3)
    samples = {"train": [], "val": []}
    for mode, phis in (("train", (0.0, 40.0, 200.0)), ("val", (90.0,))):
        for phi in phis:
            samples[mode].append({"img": rng.random((H, W, 3)),                       # float64, as cv2 image / 255.0
                                  "transform": torch.from_numpy(oracle.spherical_to_pose(4, -30, phi)).float()})
    rg = RayGenerator.from_samples(samples, [H, W, f], device=dev)
    assert (rg.H, rg.W) == (H, W) and set(rg.rays_dataset) == {"train", "val"}
    for mode, items in samples.items():
        want = torch.cat([oracle.camera_rays(s["transform"], [H, W, f]) for s in items])        # the reference's construction
        got = rg.rays_dataset[mode].cpu()
        assert got.shape == (len(items) * H * W, 6)
        assert torch.equal(got[:, :3], want[:, :3])
        assert float((got[:, 3:] - want[:, 3:]).abs().max()) <= 2.4e-7
        train_imgs = torch.stack([torch.from_numpy(s["img"]) for s in items]).reshape(-1, 3)    # train.py:33
        assert torch.equal(rg.colours[mode].cpu(), train_imgs.float())
    saved = torch.get_rng_state()
    try:
        torch.manual_seed(2)
        rays, gt, ids = rg.select_batch("train", 64)
        torch.manual_seed(2)
        assert torch.equal(ids.cpu(), torch.randperm(3 * H * W)[:64])
    finally:
        torch.set_rng_state(saved)
    assert torch.equal(rays, rg.rays_dataset["train"][ids]) and torch.equal(gt, rg.colours["train"][ids])
    with pytest.raises(RuntimeError):
        RayGenerator({"train": rg.rays_dataset["train"].cpu()})                    # tables live on the GPU
