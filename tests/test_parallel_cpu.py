"""World-size-2 (and 3, ragged) gloo tests of the multi-process layer on CPU:
ray sharding + pixel all-gather reproduce the single-process image exactly, and
the flattened gradient all-reduce reproduces the global-batch gradient.  The
per-shard renderer injected here is the CPU oracle; on GPUs it is the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, job, tmp):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nerf_oracle as O
        from nerf_simple_amd import parallel
        from nerf_simple_amd.utils import synthetic
        sd = synthetic.synthetic_state_dict(0, "structured")
        if job == "render":
            pose = torch.from_numpy(O.spherical_to_pose(4, -30, 20)).float()
            n_side = 9                                    # 81 rays: ragged for world 2
            rays = O.camera_rays(pose, [n_side, n_side, synthetic.focal_from_fov(n_side)])
            u = torch.rand(rays.shape[0], 16, generator=torch.Generator().manual_seed(3))

            def render_fn(r, us, ray_id0):
                with torch.no_grad():
                    rgb, disp, _, _, _ = O.render_nerf(r, sd, 16, u=us)
                return rgb, disp

            rgb, disp = parallel.render_image_sharded(rays, render_fn, u=u)
            np.savez(os.path.join(tmp, f"render_{rank}.npz"), rgb=rgb.numpy(), disp=disp.numpy())
        elif job == "grads":
            g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"))
            rays, u, gt = (torch.from_numpy(g[k]) for k in ("rays", "u", "gt"))
            lo, hi = parallel.shard_range(rays.shape[0], rank, world)
            sd0 = synthetic.synthetic_state_dict(0, "default")
            params = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
            rgb, _, _, _, _ = O.render_nerf(rays[lo:hi], params, int(g["N"]), u=u[lo:hi])
            torch.nn.functional.mse_loss(rgb, gt[lo:hi]).backward()
            parallel.allreduce_gradients(list(params.values()))
            np.savez(os.path.join(tmp, f"grads_{rank}.npz"), **{k: p.grad.numpy() for k, p in params.items()})
        elif job == "flatgrads":
            # gradients laid out as the fused backward hands them out: views of ONE flat vector
            shapes = [(4, 3), (4,), (2, 4), (2,)]
            params = [torch.zeros(s_, requires_grad=True) for s_ in shapes]
            flat = torch.arange(sum(int(np.prod(s_)) for s_ in shapes), dtype=torch.float32) * (rank + 1)
            off = 0
            for p_, s_ in zip(params, shapes):
                k = int(np.prod(s_))
                p_.grad = flat[off:off + k].view(s_)
                off += k
            assert parallel.flat_grad_view(params) is not None
            assert parallel.flat_grad_view(params).data_ptr() == flat.data_ptr()
            parallel.allreduce_gradients(params)
            want = torch.arange(flat.numel(), dtype=torch.float32) * (sum(range(1, world + 1)) / world)
            assert torch.equal(flat, want)                       # reduced in place, no copies
            assert all(p_.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p_ in params)
            # the flat-vector form GraphedTrainStep calls between its two graphs
            x = torch.full((7,), float(rank + 1))
            assert parallel.allreduce_flat_(x) is x and bool((x == sum(range(1, world + 1)) / world).all())
            # the bucketed, overlapped form GraphedTrainStep uses with more than one replica: two buckets of ONE flat
            # vector, each started asynchronously (work runs beside whatever the caller does next), both awaited
            # before the optimizer; the buckets are views, so the flat vector ends up reduced in place
            flat2 = torch.arange(12, dtype=torch.float32) * (rank + 1)
            late, head = flat2[5:], flat2[:5]
            h1 = parallel.allreduce_start_(late)
            busy = torch.ones(1000).sum()                         # "the head-gradient launch"
            h2 = parallel.allreduce_start_(head)
            parallel.allreduce_wait_(h1)
            parallel.allreduce_wait_(h2)
            assert float(busy) == 1000.0
            assert torch.equal(flat2, torch.arange(12, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
            # ragged pixel shards gathered into a caller-provided table
            n_tot = 5
            lo, hi = parallel.shard_range(n_tot, rank, world)
            shard = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).repeat(1, 4)
            table = torch.full((n_tot, 4), -1.0)
            assert parallel.gather_pixels(shard, n_tot, out=table) is table
            assert torch.equal(table[:, 0], torch.arange(n_tot, dtype=torch.float32))
            # and a list that is NOT one buffer takes the bucket path
            loose = [torch.zeros(3, requires_grad=True), torch.zeros(2, requires_grad=True)]
            for p_ in loose:
                p_.grad = torch.full_like(p_, float(rank + 1))
            assert parallel.flat_grad_view(loose) is None
            parallel.allreduce_gradients(loose)
            assert all(bool((p_.grad == sum(range(1, world + 1)) / world).all()) for p_ in loose)
        elif job == "bcast":
            lin = torch.nn.Linear(5, 3)
            with torch.no_grad():
                for p in lin.parameters():
                    p.fill_(float(rank + 1))
            parallel.broadcast_parameters(lin, src=0)
            assert all(bool((p == 1.0).all()) for p in lin.parameters())
    finally:
        dist.destroy_process_group()


def _run(world, job, tmp):
    mp.spawn(_worker, args=(world, _free_port(), job, str(tmp)), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_matches_single_process(world, tmp_path, oracle, synthetic):
    _run(world, "render", tmp_path)
    sd = synthetic.synthetic_state_dict(0, "structured")
    pose = torch.from_numpy(oracle.spherical_to_pose(4, -30, 20)).float()
    rays = oracle.camera_rays(pose, [9, 9, synthetic.focal_from_fov(9)])
    u = torch.rand(81, 16, generator=torch.Generator().manual_seed(3))
    want_rgb, want_disp = oracle.render_image(sd, rays, 81, N=16, u=u)
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"render_{r}.npz"))
        # per-ray independence: sharding changes nothing except MKL's blocking,
        # which may differ in the last ulp between batch shapes
        np.testing.assert_allclose(got["rgb"], want_rgb.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(got["disp"], want_disp.numpy(), rtol=2e-6, atol=0)
        assert got["rgb"].min() >= 0 and got["rgb"].max() <= 1


def test_gradient_allreduce_equals_global_batch(tmp_path, golden):
    _run(2, "grads", tmp_path)
    g = golden("train.npz")
    a = np.load(os.path.join(tmp_path, "grads_0.npz"))
    b = np.load(os.path.join(tmp_path, "grads_1.npz"))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), "replicas must hold identical averaged gradients"
        want = g[f"grad/{k}"] if f"grad/{k}" in g.files else None
        if want is not None:
            np.testing.assert_allclose(a[k], want, rtol=2e-4, atol=1e-7, err_msg=k)
        else:
            np.testing.assert_allclose(a[k][:16, :16], g[f"gradc/{k}"], rtol=2e-4, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(np.linalg.norm(a[k]), g[f"gnorm/{k}"], rtol=1e-4)


def test_gradient_allreduce_in_place_on_flat_vector(tmp_path):
    """The fused backward's flat gradient vector is all-reduced in place (world 2, gloo)."""
    _run(2, "flatgrads", tmp_path)


def test_broadcast_parameters(tmp_path):
    _run(2, "bcast", tmp_path)


def test_shard_range_partition():
    from nerf_simple_amd.parallel import shard_range
    for n in (0, 1, 7, 640000):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
