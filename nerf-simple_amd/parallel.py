"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU in the tests).

The reference has no distributed code at all; this is
new.  The render path shards by RAYS -- every ray is independent through
sampling, MLP and compositing (reference utils/rendering.py:13-85 has no
cross-ray op) -- so the data path needs no collective.  Two exchanges exist:

  * full-image renders: ONE all-gather of the packed [rgb, disparity] pixels
    (16 B/ray; 1.28 MB per rank for 800x800 on 8 GPUs, latency-bound);
  * training: ONE all-reduce of the flattened gradient (595,844 fp32 = 2.38 MB).

Jitter is indexed by GLOBAL ray id (explicit ``u`` rows, or the counter RNG's
``ray_id0``), so an image does not depend on the number of ranks.
"""
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


_single_rank_collectives = False


def force_collectives(on=True):
    """Rehearsal switch: with a process group of ONE rank the collectives below are normally skipped; switched on,
    they are issued all the same, so the exact call pattern of a multi-GPU job -- RCCL all-gather / all-reduce on the
    launch stream, the asynchronous bucketed exchange -- runs on a single MI355X (tests/test_gpu_multirank.py,
    `NERF_BENCH_FORCE_DIST=1 python bench.py`)."""
    global _single_rank_collectives
    _single_rank_collectives = bool(on)


def collectives_active(group=None):
    """True when an exchange has to be issued: more than one replica, or the single-rank rehearsal."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _single_rank_collectives


def shard_range(n, rank, world):
    """Contiguous, balanced range of rank ``rank``: [lo, hi)."""
    return rank * n // world, (rank + 1) * n // world


def gather_pixels(shard, n_total, group=None, out=None):
    """All-gather per-rank pixel shards [n_r, C] (contiguous ranges in rank
    order, from shard_range) into the full [n_total, C] table on every rank.
    Equal shards use one all_gather_into_tensor; ragged ones are padded to the
    largest shard so it is still a single collective."""
    rank, world = world_info(group)
    if not collectives_active(group):
        return shard if out is None else out.copy_(shard)
    C = shard.shape[1]
    if out is None:
        out = torch.empty((n_total, C), dtype=shard.dtype, device=shard.device)
    if n_total % world == 0:
        dist.all_gather_into_tensor(out, shard.contiguous(), group=group)
        return out
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    m = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros((m, C), dtype=shard.dtype, device=shard.device)
    padded[:shard.shape[0]] = shard
    buf = torch.empty((world * m, C), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    for r, (lo, hi) in enumerate(sizes):
        out[lo:hi] = buf[r * m:r * m + (hi - lo)]
    return out


def render_image_sharded(rays, render_fn, group=None, u=None):
    """rays [n,6] (the same full table on every rank) -> (rgb [n,3] clipped to
    [0,1], disparity [n]) on every rank.

    render_fn(rays_shard, u_shard, ray_id0) -> (rgb [m,3], disp [m]) renders one
    shard; in production it is the HIP path (rendering._render_batched), the
    gloo tests inject the CPU oracle.  Clipping happens AFTER compositing, the
    disparity is left un-clipped (reference utils/rendering.py:103-105)."""
    rank, world = world_info(group)
    n = rays.shape[0]
    lo, hi = shard_range(n, rank, world)
    rgb, disp = render_fn(rays[lo:hi], None if u is None else u[lo:hi], lo)
    shard = torch.cat([torch.clip(rgb, 0., 1.), disp.reshape(-1, 1)], dim=1)
    full = gather_pixels(shard, n, group)
    return full[:, :3], full[:, 3]


def flat_grad_view(params):
    """If the gradients of ``params`` are consecutive views of ONE contiguous fp32 buffer (what
    the fused backward hands out: nerf_amd_param_gradients writes a single flat vector in
    state_dict order), return that buffer as a 1-D tensor sharing their memory; else None."""
    params = list(params)
    if not params or any(p.grad is None for p in params):
        return None
    g0 = params[0].grad
    if g0.dtype != torch.float32 or not g0.is_contiguous():
        return None
    base, off = g0.untyped_storage().data_ptr(), g0.storage_offset()
    start = off
    for p in params:
        g = p.grad
        if (g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base or
                g.storage_offset() != off):
            return None
        off += g.numel()
    if g0.untyped_storage().nbytes() < off * 4:
        return None
    return torch.as_strided(g0, (off - start,), (1,), start)


def allreduce_flat_(flat, group=None):
    """In-place mean over the data-parallel replicas of ONE flat gradient vector (the fused
    backward's 2.38 MB bucket): a single all-reduce (RCCL reduces to the mean itself; gloo has no AVG: sum, then a scale)."""
    rank, world = world_info(group)
    if collectives_active(group):
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat /= world
    return flat


def allreduce_start_(t, group=None):
    """Start the in-place mean of ``t`` over the data-parallel replicas without blocking the current stream: the
    collective waits for what is queued on the current stream NOW and runs on the backend's own stream, so kernels
    launched afterwards overlap it.  Returns a handle for allreduce_wait_ (None with a single replica)."""
    rank, world = world_info(group)
    if not collectives_active(group):
        return None
    avg = dist.get_backend(group) == "nccl"                 # RCCL reduces to the mean itself; gloo has no AVG
    work = dist.all_reduce(t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True)
    return work, t, (None if avg else world)


def allreduce_wait_(handle):
    """Make the current stream wait for a collective started by allreduce_start_ (and finish the mean)."""
    if handle is None:
        return
    work, t, div = handle
    work.wait()
    if div:
        t /= div


def allreduce_gradients(params, group=None):
    """Average gradients over data-parallel replicas with ONE collective: the
    grads are flattened into a single contiguous bucket (2.38 MB for the NeRF
    MLP), all-reduced (sum), divided by the world size and scattered back.
    With equal per-rank batch sizes and a per-rank MSE mean, the result is the
    gradient of the global-batch MSE (reference train.py:52-54)."""
    rank, world = world_info(group)
    params = [p for p in params if p.grad is not None]
    if not collectives_active(group) or not params:
        return
    flat = flat_grad_view(params)
    if flat is not None:                       # the fused backward's flat vector: reduce it in place
        allreduce_flat_(flat, group=group)
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    off = 0
    for p in params:
        k = p.grad.numel()
        p.grad.copy_(flat[off:off + k].view_as(p.grad))
        off += k


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank ``src``'s weights (one flat broadcast)."""
    rank, world = world_info(group)
    if not collectives_active(group):
        return
    ps = list(module.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in ps:
            k = p.numel()
            p.copy_(flat[off:off + k].view_as(p))
            off += k
