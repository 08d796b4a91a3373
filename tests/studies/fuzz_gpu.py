#!/usr/bin/env python3
"""Randomised shape sweeps on the GPU box, beyond what the test suite runs every round.

    python tests/studies/fuzz_gpu.py render     # 340 random (B, N <= 768, jitter mode) cases, three precisions:
                                        # the fused render kernel == MLP launch + compositor launch, bit for bit
    python tests/studies/fuzz_gpu.py select     # 600 random (n, B <= n) selections, from collision-ridden tiny tables to 3e6
                                        # rows: ids, rows and torch's generator afterwards == torch.randperm(n)[:B]
    python tests/studies/fuzz_gpu.py train      # 25 ragged (B, N) batches: fused training gradients vs the CPU oracle's
                                        # fp32 autograd under the test suite's stated bounds (tests/test_gpu_training.py)

    python tests/studies/fuzz_gpu.py storage    # 60 random (B, N) batches through the 8-bit storage form of the training step
                                        # (tests/test_gpu_storage.py's checks: every stored element = its bf16 value rounded
                                        # to e4m3 under the block exponent; the products = float64 products of the decoded buffers)

Uses oracle/ as the checker, like the tests.  Batches of a few points can exceed the per-tensor bound on the two sigma
tensors: with similar colours along a ray d loss / d sigma is a difference of nearly equal terms, and bf16 colour noise
of 1e-3 is then tens of per cent of it (2 x 3 points: 15 % of sum |d sigma_i|); it averages out with the batch
(DESIGN.md section 8).
"""
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def render():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib(); dev = torch.device("cuda:0")
    pose = torch.from_numpy(spherical_to_pose(4, -30, 70)).float()
    allrays = camera_rays([pose], [100, 100, synthetic.focal_from_fov(100)]).float().contiguous().to(dev)
    st = _lib.stream_ptr(dev)
    rng = np.random.Generator(np.random.PCG64(7))
    bad = 0
    for prec in ("fp16", "bf16", "fp32"):
        net = Nerf(precision=prec).to(dev); net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        code = _lib.precision_code(prec); packed = net.packed_weights(code)
        n_cases = 150 if prec != "fp32" else 40
        for k in range(n_cases):
            N = int(rng.integers(1, 769)); B = int(rng.integers(1, max(2, min(10000, 300000 // N))))
            rays = allrays[:B].contiguous(); tb = torch.linspace(2, 6, N + 1).to(dev)
            mode = k % 3
            if mode == 0: jit, flags = None, 2
            elif mode == 1: jit, flags = torch.rand(B, N, device=dev), 0
            else: jit, flags = torch.sort(torch.rand(B, N, device=dev) * 4 + 2, dim=1).values.contiguous(), 1
            raw, ts = torch.empty(B, N, 4, device=dev), torch.empty(B, N, device=dev)
            _lib.check(lib.nerf_amd_mlp_forward_rays(_lib.ptr(rays), _lib.ptr(jit), _lib.ptr(tb), _lib.ptr(packed), code, flags, k, 3 * k, _lib.ptr(raw), _lib.ptr(ts), B, N, st), "a")
            two = [torch.empty(s_, device=dev) for s_ in ((B, 3), (B,), (B, N), (B,), (B, N))]
            _lib.check(lib.nerf_amd_volume_render_rays(_lib.ptr(raw), _lib.ptr(ts), _lib.ptr(rays), *[_lib.ptr(x) for x in two], B, N, st), "b")
            one = [torch.full(s_, -7.0, device=dev) for s_ in ((B, 3), (B,), (B, N), (B,), (B, N))]
            _lib.check(lib.nerf_amd_render_forward(_lib.ptr(rays), _lib.ptr(jit), _lib.ptr(tb), _lib.ptr(packed), code, flags, k, 3 * k, *[_lib.ptr(x) for x in one], None, B, N, st), "c")
            torch.cuda.synchronize()
            for a, b in zip(one, two):
                if not bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all()):
                    bad += 1; print("MISMATCH", prec, B, N, flags); break
    print("sweep done, mismatches:", bad)



def train():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_training as T
    from nerf_simple_amd.utils import synthetic
    import importlib.util
    spec = importlib.util.spec_from_file_location("nerf_oracle", os.path.join(ROOT, "oracle", "nerf_oracle.py"))
    oracle = importlib.util.module_from_spec(spec); spec.loader.exec_module(oracle)
    dev = torch.device("cuda:0")
    rng = np.random.Generator(np.random.PCG64(11))
    bad = 0
    shapes = [(1, 2), (1, 64), (2, 3), (3, 257), (1, 300), (5, 512), (100, 2)]
    shapes += [(int(rng.integers(1, 60)), int(rng.integers(2, 131))) for _ in range(18)]
    for B, N in shapes:
        for kind in ("default",):
            gen = torch.Generator().manual_seed(B * 1000 + N)
            pose = torch.from_numpy(oracle.spherical_to_pose(4, -30, 0)).float()
            side = int(np.ceil(np.sqrt(B)))
            rays = oracle.camera_rays(pose, [side, side, synthetic.focal_from_fov(side)])[:B].contiguous()
            gt = torch.rand(B, 3, generator=gen); u = torch.rand(B, N, generator=gen)
            try:
                loss, grads = T._fused_grads(dev, synthetic, kind, rays, gt, u, N)
                fin = all(bool(torch.isfinite(torch.as_tensor(g)).all()) for g in (grads.values() if isinstance(grads, dict) else grads))
                T._compare_with_oracle(oracle, synthetic, kind, rays, gt, u, N, loss, grads, f"{B}x{N}")
                print(f"ok   {B:3d} x {N:3d}  P={B*N:5d} loss={loss:.5f} finite={fin}", flush=True)
                if not fin: bad += 1
            except AssertionError as e:
                bad += 1; print(f"FAIL {B} x {N}: {str(e)[:200]}", flush=True)
    print("fuzz done, failures:", bad)



def select():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    from nerf_simple_amd.utils.dataload import RayGenerator
    dev = torch.device("cuda:0")
    rng = np.random.Generator(np.random.PCG64(11))
    bad = 0
    saved = torch.get_rng_state()
    for k in range(600):
        kind = k % 4
        n = int(rng.integers(1, 40)) if kind == 0 else int(rng.integers(40, 6000)) if kind == 1 else \
            int(rng.integers(6000, 60000)) if kind == 2 else int(rng.integers(60000, 3_000_000))
        B = n if rng.random() < 0.3 else int(rng.integers(1, min(n, 8192) + 1))
        i = torch.arange(n, device=dev, dtype=torch.float32)
        rays = torch.stack([i, -i, i * 0.5, i + 1, i + 2, i + 3], 1).contiguous()
        cols = torch.stack([i, i + 0.5, -i], 1).contiguous()
        rg = RayGenerator({"train": rays}, {"train": cols})
        torch.manual_seed(int(rng.integers(0, 2 ** 31)))
        pre = int(rng.integers(0, 1500))
        if pre:
            torch.rand(pre)
        st = torch.get_rng_state()
        want = torch.randperm(n)[:B]
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        r, g, ids = rg.select_batch("train", B)
        ok = torch.equal(ids.cpu(), want) and torch.equal(torch.get_rng_state(), after) and torch.equal(r, rays[ids]) and torch.equal(g, cols[ids])
        seed = int(rng.integers(0, 2 ** 40))
        _, _, idc = rg.select_batch("train", B, device_rng=True, seed=seed)
        okc = np.array_equal(idc.cpu().numpy(), O.select_ids_counter(n, B, seed)) if n <= 60000 else len(set(idc.tolist())) == B
        if not (ok and okc):
            bad += 1
            print("MISMATCH", n, B, pre, ok, okc)
    torch.set_rng_state(saved)
    print("select sweep done, mismatches:", bad)


def storage():
    import importlib.util
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    spec = importlib.util.spec_from_file_location("storage_tests", os.path.join(ROOT, "tests", "test_gpu_storage.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    bad = 0
    shapes = [(int(rng.integers(1, 400)), int(rng.integers(1, 40))) for _ in range(int(os.environ.get('FUZZ_BIG', '45')))] + \
             [(int(rng.integers(1, 9)), int(rng.integers(1, 9))) for _ in range(15)]
    for n, (B, N) in enumerate(shapes):
        kind = ("default", "structured")[n % 2]
        try:
            a16 = T.run_chain(dev, synthetic, B, N, False, kind, seed=100 + n)
            a8 = T.run_chain(dev, synthetic, B, N, True, kind, seed=100 + n)
            P = a16["P"]
            for k in ("raw", "ts", "d_raw"):
                assert np.array_equal(a16[k], a8[k]), k
            for name in ("acts", "dys"):
                ref = T.decode_bf16_layers(a16[name], P)
                vals, raws, exps = T.decode_e4m3_layers(a8[name], P)
                for L in range(10):
                    T.check_rounding(vals[L], raws[L], exps[L], ref[L], 128 if L == 9 else 256, (name, L))
            want = T.expected_grads(a8, None)
            off = 0
            for k, p in Nerf().named_parameters():
                m = p.numel()
                got = a8["grads"][off:off + m].reshape(p.shape).astype(np.float64)
                off += m
                w = want[k].reshape(p.shape)
                err, scale = np.abs(got - w).max(), max(np.abs(w).max(), 1e-30)
                assert err <= 2.0 ** -11 * scale + 1e-12, f"{k}: error {err / scale:.3e} of the largest entry (bound 2^-11 = 4.9e-4)"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", B, N, kind, str(e)[:200])
        if n % 10 == 9:
            print(f"  {n + 1} / {len(shapes)} shapes", flush=True)
    print("storage sweep done, mismatches:", bad)


if __name__ == "__main__":
    if len(sys.argv) != 2 or sys.argv[1] not in ("render", "train", "select", "storage"):
        sys.exit(__doc__)
    {"render": render, "train": train, "select": select, "storage": storage}[sys.argv[1]]()
