"""Training-side entry points: the step of reference train.py:47-57 on the GPU.

What runs where in a training step (BASELINE config 5); every stage is a hand-written HIP kernel
behind the C ABI, torch supplies tensors, streams, autograd bookkeeping and the collective:

  sampling + encoding + 12 dense layers, forward    nerf_amd_mlp_forward_train (the fused inference
                                                    kernel, also saving point-blocked bf16 activations
                                                    and ReLU mask bit planes)
  sigma -> alpha compositing, forward               nerf_amd_volume_render
  MSE loss and its gradient                         nerf_amd_mse_loss
  compositing, backward (suffix-sum scan)           nerf_amd_volume_render_backward
  dense layers, backward dX chain (on-chip)         nerf_amd_mlp_backward
  dense layers, dW = dY^T X and db = sum dY         nerf_amd_param_gradients (one split-K launch for
                                                    all 14 products into ONE flat gradient vector)
  gradient exchange                                 RCCL all-reduce of that flat vector, in place (parallel.py)
  optimizer                                         optim.FusedAdam (one launch + re-pack) or
                                                    torch.optim.Adam (reference train.py:43)
  the whole step as captured hipGraphs              GraphedTrainStep (below)

Precision: the fused training kernels exist in bf16 only (gradients need bf16's exponent range; fp16
would need loss scaling).  A module built with precision='bf16' or 'fp16' trains through them --
``precision`` selects the INFERENCE kernel only.  precision='fp32' trains EXACTLY, as the reference does
(fp32 weights, activations and gradients: utils/generic_mlp.py, every nn.Linear and its backward on the
strided fp32 MFMA GEMM nerf_amd_linear_f32, layer by layer): the slow path, exact to fp32 round-off -- its gradients match
the reference's autograd to 1e-5; long dW / db reductions are summed with float atomics, so the last bits vary from run to run -- for checks and small problems; GraphedTrainStep is the fused bf16 step only.
"""

import torch

from . import _lib


def img_mse(gt, pred):
    """mean((pred - gt)^2)  (reference train.py:16-19)."""
    if not torch.is_tensor(gt):
        gt = torch.from_numpy(gt).float()
    return torch.mean((pred - gt) ** 2)


def img_psnr(gt, pred):
    """20 log10(max(gt)) - 10 log10(mse): the peak is max(gt), not 1.0
    (reference train.py:21-26)."""
    if not torch.is_tensor(gt):
        gt = torch.from_numpy(gt).float()
    ten = torch.tensor(10.0)
    return 20 * torch.log(torch.max(gt)) / torch.log(ten) - 10 * torch.log(img_mse(gt, pred)) / torch.log(ten)


# --------------------------------------------------------------------------
# MSELoss with the HIP kernel (value and gradient in one launch)
# --------------------------------------------------------------------------
class _MseLoss(torch.autograd.Function):
    """nn.MSELoss()(pred, target) (reference train.py:42,52: mean over all elements) through
    nerf_amd_mse_loss: one launch writes the loss and 2 (pred - target) / n."""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.contiguous()
        if pred.shape != target.shape:
            raise RuntimeError(f"mse_loss: shapes {tuple(pred.shape)} and {tuple(target.shape)} differ")
        dev = pred.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        g_pred = torch.empty_like(pred)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nerf_amd_mse_loss(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(loss), _lib.ptr(g_pred),
                                                    pred.numel(), _lib.stream_ptr(dev)), "nerf_amd_mse_loss")
        ctx.save_for_backward(g_pred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (g_pred,) = ctx.saved_tensors
        return g_pred * g, None


def mse_loss(pred, target):
    """criterion(rgb, gt_colors) of reference train.py:42,52 on device tensors (fp32)."""
    _lib.require_cuda_f32(pred, "pred")
    _lib.require_cuda_f32(target, "target")
    return _MseLoss.apply(pred, target.detach())


# --------------------------------------------------------------------------
# compositor with a HIP backward
# --------------------------------------------------------------------------
class _VolumeRender(torch.autograd.Function):
    """volume_render with the HIP backward.  ``from_rays``: ``dirs`` is the [B,6] ray table and the
    kernels normalise rays[:,3:] themselves (render_nerf, utils/rendering.py:37,43)."""

    @staticmethod
    def forward(ctx, raw, ts, dirs, from_rays):
        B, N = raw.shape[0], raw.shape[1]
        dev = raw.device
        raw, ts, dirs = raw.contiguous(), ts.contiguous(), dirs.contiguous()
        rgb = torch.empty((B, 3), dtype=torch.float32, device=dev)
        disp = torch.empty((B,), dtype=torch.float32, device=dev)
        acc = torch.empty((B,), dtype=torch.float32, device=dev)
        alpha = torch.empty((B, N), dtype=torch.float32, device=dev)
        w = torch.empty((B, N), dtype=torch.float32, device=dev)
        lib = _lib.lib()
        with torch.cuda.device(dev):
            if from_rays:
                _lib.check(lib.nerf_amd_volume_render_rays(
                    _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), _lib.ptr(rgb), _lib.ptr(disp),
                    _lib.ptr(alpha), _lib.ptr(acc), _lib.ptr(w), B, N, _lib.stream_ptr(dev)),
                    "nerf_amd_volume_render_rays")
            else:
                _lib.check(lib.nerf_amd_volume_render(
                    _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), 3, _lib.ptr(rgb), _lib.ptr(disp),
                    _lib.ptr(alpha), _lib.ptr(acc), _lib.ptr(w), B, N, _lib.stream_ptr(dev)),
                    "nerf_amd_volume_render")
        ctx.save_for_backward(raw, ts, dirs)
        ctx.from_rays = from_rays
        if N == 1:        # the reference's empty sample axis (utils/rendering.py:60-61; csrc/composite_device.h)
            alpha, w = alpha[:, :0], w[:, :0]
        return rgb, disp, alpha, acc, w

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_alpha, g_acc, g_w):
        raw, ts, dirs = ctx.saved_tensors
        B, N = raw.shape[0], raw.shape[1]
        dev = raw.device
        d_raw = torch.empty_like(raw)

        def c(g):
            return None if (g is None or g.numel() == 0) else g.contiguous().float()
        g_rgb, g_disp, g_alpha, g_acc, g_w = map(c, (g_rgb, g_disp, g_alpha, g_acc, g_w))
        lib = _lib.lib()
        with torch.cuda.device(dev):
            if ctx.from_rays:
                _lib.check(lib.nerf_amd_volume_render_rays_backward(
                    _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), _lib.ptr(g_rgb), _lib.ptr(g_disp),
                    _lib.ptr(g_alpha), _lib.ptr(g_acc), _lib.ptr(g_w), _lib.ptr(d_raw), B, N,
                    _lib.stream_ptr(dev)), "nerf_amd_volume_render_rays_backward")
            else:
                _lib.check(lib.nerf_amd_volume_render_backward(
                    _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), 3, _lib.ptr(g_rgb), _lib.ptr(g_disp),
                    _lib.ptr(g_alpha), _lib.ptr(g_acc), _lib.ptr(g_w), _lib.ptr(d_raw), B, N,
                    _lib.stream_ptr(dev)), "nerf_amd_volume_render_backward")
        return d_raw, None, None, None


def volume_render_autograd(nerf_outs, ts, dirs):
    """volume_render (reference utils/rendering.py:47-85) with gradients to
    nerf_outs; ts and dirs get none (they carry none in the reference either)."""
    return _VolumeRender.apply(nerf_outs.float(), ts.detach(), dirs.detach(), False)


# --------------------------------------------------------------------------
# fused dense layers: HIP forward (saving activations) + HIP dX chain + HIP dW
# --------------------------------------------------------------------------
def ctypes_stream(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)


def _check_fused_trainable(precision):
    if _lib.precision_code(precision) == _lib.F32:
        raise RuntimeError("the fused training step is bf16 (NERF_AMD_EUNSUP for precision='fp32'): build the module with "
                           "precision='bf16' (or 'fp16': bf16 training, fp16 inference), or train the fp32 module with "
                           "training.train_step (exact, layer by layer)")


class _FusedDense(torch.autograd.Function):
    """(rays, jitter) -> raw [B,N,4], ts [B,N]  -- or, with ``rays`` None, points v [P,6] -> raw [P,1,4] --
    through nerf_amd_mlp_forward_train[_points]; backward: nerf_amd_mlp_backward for every layer's
    pre-activation gradient, then nerf_amd_param_gradients (split-K GEMMs + column sums) into ONE flat
    fp32 vector in state_dict order, handed back to autograd as 24 views."""

    @staticmethod
    def forward(ctx, net, rays, jit, tbins, flags, seed, ray_id0, N, pts, *params):
        lib = _lib.lib()
        src = rays if rays is not None else pts
        B, dev = src.size(0), src.device
        P = B * N
        # a diverged run is loud: the status words of the previous forwards (non-finite activations / weights) are
        # looked at here, without waiting (_StatusWatch below; the graphed step does the same)
        watch = net.__dict__.get("_watch")
        if watch is None:
            watch = net.__dict__["_watch"] = _StatusWatch()
            net.__dict__["_watch_calls"] = 0
        bad = watch.poll()
        if bad is not None:
            what = " and ".join(w for w, on in (("activations", bad[1]), ("weights", bad[2])) if on)
            raise FloatingPointError(f"non-finite values inside the network in training forward {bad[0]} ({what}): "
                                     "NaN / inf weights or inputs, the run has diverged")
        packed = net.packed_weights(_lib.BF16)
        raw = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        ts = torch.empty((B, N), dtype=torch.float32, device=dev)
        acts = torch.empty(int(lib.nerf_amd_train_activation_bytes(P)), dtype=torch.uint8, device=dev)
        posx = torch.empty((P, 64), dtype=torch.bfloat16, device=dev)
        posd = torch.empty((P, 32), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            if rays is not None:
                _lib.check(lib.nerf_amd_mlp_forward_train(
                    _lib.ptr(rays), _lib.ptr(jit), _lib.ptr(tbins), _lib.ptr(packed), flags, int(seed), int(ray_id0),
                    _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(acts), B, N, st), "nerf_amd_mlp_forward_train")
                # encoder outputs in the reference's column order: the inputs of the dW products of
                # layers_0.0 / skip_conn_layer / color_fc.0 (same sample positions: ts given)
                _lib.check(lib.nerf_amd_sample_encode_bf16(
                    _lib.ptr(rays), _lib.ptr(ts), None, _lib.FLAG_TS_GIVEN, 0, 0,
                    _lib.ptr(posx), _lib.ptr(posd), None, B, N, st), "nerf_amd_sample_encode_bf16")
            else:
                _lib.check(lib.nerf_amd_mlp_forward_train_points(
                    _lib.ptr(pts), _lib.ptr(packed), _lib.ptr(raw), _lib.ptr(acts), P, st),
                    "nerf_amd_mlp_forward_train_points")
                _lib.check(lib.nerf_amd_encode_points_bf16(_lib.ptr(pts), _lib.ptr(posx), _lib.ptr(posd), P, st),
                           "nerf_amd_encode_points_bf16")
        net.__dict__["_watch_calls"] += 1
        watch.push(packed, net.__dict__["_watch_calls"])
        ctx.net, ctx.P = net, P
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(acts, posx, posd)
        ctx.mark_non_differentiable(ts)
        return raw, ts

    @staticmethod
    def backward(ctx, g_raw, _g_ts):
        lib = _lib.lib()
        acts, posx, posd = ctx.saved_tensors
        net, P = ctx.net, ctx.P
        dev = acts.device
        g = g_raw.reshape(P, 4).contiguous().float()
        image = net.packed_weights(_lib.BF16_BWD)
        dys = torch.empty_like(acts)
        flat = torch.empty(int(lib.nerf_amd_param_count()), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(lib.nerf_amd_mlp_backward(_lib.ptr(g), _lib.ptr(image), _lib.ptr(acts), _lib.ptr(dys), P, st),
                       "nerf_amd_mlp_backward")
            _lib.check(lib.nerf_amd_param_gradients(_lib.ptr(g), _lib.ptr(acts), _lib.ptr(dys), _lib.ptr(posx),
                                                    _lib.ptr(posd), _lib.ptr(scratch), _lib.ptr(flat), P, st),
                       "nerf_amd_param_gradients")
        grads, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for s_ in shp:
                n *= s_
            grads.append(flat[off:off + n].view(shp))
            off += n
        return (None,) * 9 + tuple(grads)


def nerf_forward_autograd(net, v, precision):
    """Nerf.forward (reference utils/nets.py:34-43) with gradients to the parameters: the fused
    training forward on points, v [P,6] -> [P,4] (precision 'fp32': the exact layer-by-layer path)."""
    if _lib.precision_code(precision) == _lib.F32:
        from .utils import generic_mlp
        return generic_mlp.forward(net, v)
    params = [p for _, p in net.named_parameters()]
    raw, _ = _FusedDense.apply(net, None, None, None, 0, 0, 0, 1, v.detach().contiguous(), *params)
    return raw.reshape(-1, 4)


def render_nerf_autograd(rays, net, N, tn, tf, jit, flags, precision, seed, ray_id0):
    """render_nerf (reference utils/rendering.py:13-45) with gradients to the
    parameters of ``net``; returns the same 5-tuple."""
    from .utils.rendering import _tbins
    if _lib.precision_code(precision) == _lib.F32:
        # exact fp32: the reference's own composition (sampling -> net.forward -> volume_render), each stage a HIP kernel
        from .utils import generic_mlp
        from .utils.rendering import ALL_OUTPUTS, _render_generic

        class _Exact:
            @staticmethod
            def forward(q):
                return generic_mlp.forward(net, q)

        return _render_generic(rays, _Exact, N, tn, tf, jit, flags, ALL_OUTPUTS, seed, ray_id0)
    dev = rays.device
    params = [p for _, p in net.named_parameters()]
    raw, ts = _FusedDense.apply(net, rays, jit, _tbins(tn, tf, N, dev), flags, seed, ray_id0, N, None, *params)
    return _VolumeRender.apply(raw, ts, rays, True)


# --------------------------------------------------------------------------
# one optimisation step (reference train.py:47-57)
# --------------------------------------------------------------------------
def lr_decay_factor(lr_init, lr_final, num_iters):
    """Per-iteration multiplicative decay (reference train.py:36-39)."""
    import math
    return math.exp(math.log(lr_final / lr_init) / num_iters)


def train_step(net, optimizer, rays, gt, N, *, tn=2, tf=6, u=None, decay=1.0, group=None,
               precision=None, device_rng=False, seed=0, ray_id0=0):
    """zero_grad -> render_nerf -> MSELoss(rgb, gt) -> backward -> [grad all-reduce]
    -> optimizer.step -> lr *= decay.  Returns the (detached) loss.
    Only ``rgb`` feeds the loss, as in the reference (train.py:52).
    The dense layers train in bf16 whatever the module's inference precision ('bf16' or 'fp16');
    precision 'fp32' trains exactly, layer by layer in fp32 (utils/generic_mlp.py): slow, and the reference's numbers."""
    from . import parallel
    from .utils.rendering import render_nerf
    optimizer.zero_grad(set_to_none=True)
    rgb, _, _, _, _ = render_nerf(rays, net, N, tn, tf, u=u, precision=precision,
                                  device_rng=device_rng, seed=seed, ray_id0=ray_id0)
    loss = mse_loss(rgb, gt)
    loss.backward()
    parallel.allreduce_gradients(net.parameters(), group=group)
    optimizer.step()
    if decay != 1.0:
        for pg in optimizer.param_groups:
            pg["lr"] = pg["lr"] * decay
    return loss.detach()


# --------------------------------------------------------------------------
# the same step as ONE captured hipGraph (launch-bound at 4096-ray batches)
# --------------------------------------------------------------------------
class _HyperRing:
    """Pinned host ring for the per-step scalars (Adam's six floats + the jitter seed offset as an int64): one slot per
    step, written by the host before it launches the step and read on the device by the first node of graph A
    (nerf_amd_hyper_fetch: slot = device counter % slots) -- no copy between two graph launches.  A slot is rewritten only
    after the event recorded behind the launch that read it has completed; host index and device counter advance in
    lockstep (``reset`` puts both back to 0)."""

    def __init__(self, dev, slots=16):
        self.ring = torch.zeros((slots, 8), dtype=torch.float32).pin_memory()
        with torch.cuda.device(dev):
            self.ring_dev = int(_lib.lib().nerf_amd_pinned_device_address(_lib.ptr(self.ring)))
        if self.ring_dev <= 0:
            raise RuntimeError("the pinned host ring of the step's scalars is not mapped into the device's address space")
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.events = [None] * slots
        self.k = 0

    def reset(self):
        self.k = 0
        self.counter.zero_()

    def push(self, values, seed_offset=0):
        i = self.k % self.ring.shape[0]
        if self.events[i] is not None:
            self.events[i].synchronize()
            self.events[i] = None
        h = self.ring[i]
        for j, v in enumerate(values):
            h[j] = v
        h[6:8].view(torch.int64)[0] = int(seed_offset)

    def fetch(self, dst, dev):
        """Enqueue (or capture) the device-side read of the next slot into ``dst``."""
        import ctypes
        _lib.check(_lib.lib().nerf_amd_hyper_fetch(ctypes.c_void_p(self.ring_dev), int(self.ring.shape[0]), _lib.ptr(dst), _lib.ptr(self.counter),
                                                   _lib.stream_ptr(dev)), "nerf_amd_hyper_fetch")

    def launched(self, dev):
        """The launch that reads the slot just pushed is in the stream."""
        i = self.k % self.ring.shape[0]
        self.k += 1
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.events[i] = ev


class _StatusWatch:
    """Asynchronous read-back of the status words behind a packed 16-bit weight image (include/nerf_amd.h
    nerf_amd_packed_status_offset): the training forward sets word 0 when a point shows a non-finite value inside the
    network, the re-pack of every step sets word 1 when a weight is not finite.  In the reference such a step ends in a NaN loss for everyone to see; here the kernels' integer ReLU can
    turn the NaNs into finite garbage, so the flag is what makes a diverged run loud.  ``push`` enqueues an 8-byte copy
    into pinned memory behind the forward, ``poll`` looks at the copies that have completed -- no host wait."""

    def __init__(self, slots=4):
        self.bufs = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(slots)]
        self.pending = []                       # (event, slot, step)
        self.k = 0

    def push(self, packed, step):
        if len(self.pending) >= len(self.bufs):
            return                              # every slot still in flight: skip this sample
        slot = self.k % len(self.bufs)
        self.k += 1
        off = int(_lib.lib().nerf_amd_packed_status_offset(_lib.BF16))
        self.bufs[slot].copy_(packed[off:off + 8].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(packed.device))
        self.pending.append((ev, slot, step))

    def poll(self):
        """The first completed sample with a flag set, as (step, activations flagged, weights flagged), or None."""
        hit = None
        while self.pending and self.pending[0][0].query():
            _, slot, step = self.pending.pop(0)
            words = (int(self.bufs[slot][0]), int(self.bufs[slot][1]))
            if hit is None and (words[0] != 0 or words[1] != 0):
                hit = (step, words[0] != 0, words[1] != 0)
        return hit


class GraphedTrainStep:
    """``train_step`` (reference train.py:47-57) for the fused bf16 path with every buffer
    allocated once and the launches captured into hipGraphs that are replayed per iteration:

        graph A: forward (saving activations) -> compositor + MSE gradient + compositor backward (one
                 kernel) -> dX chain -> all 24 parameter gradients, with the encoder rows, the loss
                 value, the gradient zero fill and the d_raw pack on a parallel branch
                 (9 kernel nodes, all through the C ABI: no torch kernels, no memset node)
        graph B: Adam over the flat parameter vector -> re-pack the two MFMA weight images (one kernel)

    With a process group of more than one replica the flat gradient vector is averaged between graph A and graph B:

        buckets=1 (default)  ONE in-place all-reduce of the 2.38 MB vector; fully exposed, and the cheapest form
                             measured: +16 ... 26 us per step with RCCL on one rank (1.185 -> 1.201 ms).
        buckets=2            the gradients are produced in two launches and reduced in two buckets
                             (include/nerf_amd.h nerf_amd_grad_bucket_range):
                               graph A1: ... -> dX chain -> gradients of the LATE layers (the tail of the vector)
                               all-reduce of that bucket (1.27 MB) starts on the collective's own stream
                               graph A2: gradients of layers_0.* (the head)      <- runs while bucket 1 is on the wire
                               all-reduce of the head bucket (1.12 MB); both awaited; graph B
                             Only the second exchange is exposed -- but the split itself costs 75 ... 90 us per step (a
                             second 64 MB tail of split-K atomics and a second ramp of the HBM-bound gradient kernel),
                             measured with RCCL on one rank: 1.288 vs 1.201 ms.  It pays only where one 1.27 MB
                             all-reduce takes longer than that.

    ``timing=True`` records events around the exchange (``collective_times()``).

    Step-dependent scalars do not live in kernel arguments: the jitter comes from the ``u``
    buffer (filled per call; default the reference's one ``torch.rand(B, N)`` draw from the CPU
    generator, continued on the device by utils/host_rng.py), Adam's learning rate and bias
    corrections from a 6-float device vector (nerf_amd_adam_step_hyper) fed through a ring of
    pinned buffers.  ``optimizer`` must be ``optim.FusedAdam``.

    ``device_rng=True``: the jitter is drawn inside the kernels by the counter RNG instead (no ``u`` traffic, no host
    work, no wait for the CPU generator's state): seed = ``seed`` + the step count, which travels with the Adam scalars
    (the launches carry NERF_AMD_SEED_IN_MEMORY, so the replayed graph reads the current value); ``ray_id0`` offsets
    the ray ids of this replica, so data-parallel ranks given rank * n_rays draw different jitter.  Same values as the
    eager ``train_step(..., device_rng=True, seed=seed + k, ray_id0=ray_id0)`` at step k (1-based).

    ``step(rays, gt, u=None, decay=1.0)`` returns the loss as a 0-d device tensor (no sync).

    ``rays_from`` = a ``utils.dataload.RayGenerator`` (tables resident in HBM): ``step()`` without rays then runs the
    first lines of the reference's iteration itself (train.py:47-49: ``rg.select(mode, N=batch_size)`` and the
    ``train_imgs[ray_ids]`` gather) on the device.  With ``device_rng=True`` the selection is a node of graph A (counter RNG
    keyed by seed + step, the step counter read from device memory): every replay selects the batch of the NEXT step
    beside its own dX chain -- the selection depends on the counter only, never on the weights -- so it costs the step
    nothing and nothing runs on the host; ``ray_ids`` (the rows the last step trained on) is recomputed on demand.
    Otherwise the ids come from torch's CPU generator continued on the device: the reference's own ``ray_ids`` followed
    by its jitter draw, the generator left exactly where the reference's iteration leaves it.

    ``storage='e4m3'``: what the forward saves for the weight gradients and what the dX chain writes for them travels
    through HBM as 8-bit floats with one power-of-two exponent per 32 features x 32 points instead of bf16 (half the
    bytes of the step's three HBM-bound kernels; the dW products run on the block-scaled 8-bit MFMA).  The forward's
    outputs, the loss and d_raw are bit for bit those of the default; the gradients carry the operands' 8-bit rounding,
    inside the same criterion (a fraction of the reference's own minibatch deviation; tests/test_gpu_storage.py).

    ``check_every`` (default 16, 0 = never): every so many steps the forward's range flag is copied back without
    waiting; a later ``step`` raises FloatingPointError once such a copy shows non-finite values inside the network
    (NaN / inf weights or inputs: a diverged run) -- the reference would show a NaN loss there.
    """

    def __init__(self, net, optimizer, n_rays, N, *, tn=2, tf=6, group=None, timing=False, buckets=1,
                 device_rng=False, seed=0, ray_id0=0, check_every=16, rays_from=None, select_mode="train", storage="bf16"):
        from . import parallel
        from .optim import FusedAdam
        from .utils.rendering import _tbins
        if not isinstance(optimizer, FusedAdam):
            raise RuntimeError("GraphedTrainStep needs optim.FusedAdam (one flat parameter vector)")
        if optimizer.net is not net:
            raise RuntimeError("the optimizer belongs to another module")
        _check_fused_trainable(net.precision)
        self.net, self.opt, self.group = net, optimizer, group
        if buckets not in (1, 2):
            raise ValueError("buckets must be 1 (one all-reduce between the two graphs, the default) or 2 (overlapped)")
        # group=None is the default process group, as everywhere in parallel.py (train_step(group=None) reduces over it too)
        self.exchange = parallel.collectives_active(group)
        self.bucketed = self.exchange and buckets == 2
        self.timing, self._events = bool(timing), []
        self.device_rng, self.seed, self.ray_id0 = bool(device_rng), int(seed), int(ray_id0)
        self.check_every, self._watch = int(check_every), _StatusWatch()
        self.B, self.N = int(n_rays), int(N)
        if storage not in ("bf16", "e4m3"):
            raise ValueError("storage must be 'bf16' (the default) or 'e4m3'")
        self.storage, self._e4m3 = storage, storage == "e4m3"
        dev = optimizer.flat.device
        self.dev = dev
        self.rays_from, self.select_mode = rays_from, select_mode
        if rays_from is not None:
            table = rays_from.rays_dataset[select_mode]
            if select_mode not in rays_from.colours:
                raise RuntimeError(f"rays_from has no colour table for mode {select_mode!r}")
            if table.device != dev:
                raise RuntimeError(f"rays_from lives on {table.device}, the module on {dev}")
            if int(table.shape[0]) < self.B:
                raise RuntimeError(f"a batch of {self.B} rays from a table of {int(table.shape[0])}")
        lib = _lib.lib()
        B, N_, P = self.B, self.N, self.B * self.N
        f32 = dict(dtype=torch.float32, device=dev)
        self.rays = torch.zeros((B, 6), **f32)
        self.rays[:, 5] = -1.0                      # a valid direction: the capture warm-up runs on these buffers
        self.gt = torch.zeros((B, 3), **f32)
        self.u = torch.zeros((B, N_), **f32)
        self.tbins = _tbins(tn, tf, N_, dev)
        self.raw = torch.empty((B, N_, 4), **f32)
        self.ts = torch.empty((B, N_), **f32)
        if self._e4m3:
            self.acts = torch.empty(int(lib.nerf_amd_train_activation_bytes_e4m3(P)), dtype=torch.uint8, device=dev)
            self.dys = torch.empty(int(lib.nerf_amd_train_gradient_bytes_e4m3(P)), dtype=torch.uint8, device=dev)
            self.scratch8 = torch.empty(max(int(lib.nerf_amd_param_gradients_scratch_e4m3_bytes(P)), 16), dtype=torch.uint8, device=dev)
        else:
            nb = int(lib.nerf_amd_train_activation_bytes(P))
            self.acts = torch.empty(nb, dtype=torch.uint8, device=dev)
            self.dys = torch.empty(nb, dtype=torch.uint8, device=dev)
        self.posx = torch.empty((P, 64), dtype=torch.bfloat16, device=dev)
        self.posd = torch.empty((P, 32), dtype=torch.bfloat16, device=dev)
        self.rgb = torch.empty((B, 3), **f32)
        self.d_raw = torch.empty((B, N_, 4), **f32)
        self.grads = torch.zeros(int(lib.nerf_amd_param_count()), **f32)
        self.scratch = torch.empty(max(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), 16), dtype=torch.uint8,
                                   device=dev)
        self.loss = torch.zeros((), **f32)
        self._ids_next = torch.zeros((B,), dtype=torch.int64, device=dev)      # rays_from: rows of the table (see ray_ids)
        self._ids_cur, self._ids_step, self._primed_for = torch.zeros_like(self._ids_next), -1, -1
        self._select_ws = torch.empty(max(int(lib.nerf_amd_select_workspace_bytes(B)), 256), dtype=torch.uint8, device=dev)
        self._select_ws2 = torch.empty_like(self._select_ws)                    # eager selections beside the captured one
        import ctypes
        first, count = ctypes.c_int64(), ctypes.c_int64()
        self.buckets = []                                   # views of the flat gradient vector, in exchange order
        for b in (1, 2):
            _lib.check(lib.nerf_amd_grad_bucket_range(b, ctypes.byref(first), ctypes.byref(count)), "nerf_amd_grad_bucket_range")
            self.buckets.append(self.grads[first.value:first.value + count.value])
        self.hyper = torch.zeros(8, **f32)                 # [lr, b1, b2, eps, 1-b1^t, sqrt(1-b2^t), seed offset (int64)]
        self._ring = _HyperRing(dev)
        self._side = torch.cuda.Stream(dev)
        # parameters' .grad are views of the flat gradient vector, as after the eager fused backward
        off = 0
        for p in optimizer.params:
            k = p.numel()
            p.grad = self.grads[off:off + k].view(p.shape)
            off += k
        self._capture()

    # ---- the two launch sequences --------------------------------------------------
    def _forward_backward(self, bucket=0):
        """Main branch: forward -> compositing + MSE gradient + compositing backward -> dX chain -> dW (all products,
        or with ``bucket`` = 1 only those of the late layers: _head_gradients adds the rest).
        Side branch (ONE fork / join inside the captured graph, behind the compositor and beside the dX chain): the encoder
        rows of the dW products, loss value, gradient-vector zero fill, d_raw pack, (8-bit form: the narrow operands'
        conversion,) and the next step's batch selection."""
        lib, B, N_, P = _lib.lib(), self.B, self.N, self.B * self.N
        packed = self.net.packed_weights(_lib.BF16)
        image = self.net.packed_weights(_lib.BF16_BWD)
        ck, ptr = _lib.check, _lib.ptr
        main = torch.cuda.current_stream(self.dev)
        side = self._side
        st, ss = ctypes_stream(main), ctypes_stream(side)
        # first node: this step's scalars (Adam's, the jitter seed offset) from the pinned host ring into `hyper`
        self._ring.fetch(self.hyper, self.dev)
        if self.device_rng:
            # counter RNG; `u` = the address of this step's seed offset inside the hyper vector (int64 at float slot 6)
            import ctypes
            jit = ctypes.c_void_p(self.hyper.data_ptr() + 24)
            flags, seed, rid = _lib.FLAG_DEVICE_RNG | _lib.FLAG_SEED_IN_MEMORY, self.seed, self.ray_id0
        else:
            jit, flags, seed, rid = ptr(self.u), 0, 0, 0
        ck(lib.nerf_amd_mlp_forward_train(ptr(self.rays), jit, ptr(self.tbins), ptr(packed),
                                          flags | (_lib.FLAG_STORE_E4M3 if self._e4m3 else 0), seed, rid,
                                          ptr(self.raw), ptr(self.ts), ptr(self.acts), B, N_, st),
           "nerf_amd_mlp_forward_train")
        # only rgb feeds the loss (train.py:52): disparity, alpha, acc, w are not materialised
        ck(lib.nerf_amd_volume_render_mse_backward(ptr(self.raw), ptr(self.ts), ptr(self.rays), ptr(self.gt), ptr(self.rgb),
                                                   ptr(self.d_raw), B, N_, st), "nerf_amd_volume_render_mse_backward")
        fork = torch.cuda.Event()
        fork.record(main)                           # behind the compositor: what the side branch needs (rgb, d_raw) is final
        backward = lib.nerf_amd_mlp_backward_e4m3 if self._e4m3 else lib.nerf_amd_mlp_backward
        ck(backward(ptr(self.d_raw), ptr(image), ptr(self.acts), ptr(self.dys), P, st), "nerf_amd_mlp_backward")
        side.wait_event(fork)
        # The side branch carries everything the dW products need besides dY -- and nothing else runs beside the forward:
        # a kernel enqueued next to a persistent kernel that fills every CU either delays its start (~10 us per branch at a
        # replayed graph's root) or crawls beside it and slows the compositor behind it.  Order: the encoder rows read
        # this step's rays and jitter, so they come before the selection overwrites the batch.
        # same sample positions as the forward drew them: ts = f(jitter) bit for bit (the same flags / u / seed / tbins)
        ck(lib.nerf_amd_sample_encode_bf16(ptr(self.rays), jit, ptr(self.tbins), flags, seed, rid,
                                           ptr(self.posx), ptr(self.posd), None, B, N_, ss),
           "nerf_amd_sample_encode_bf16")
        if self._e4m3:       # the encoder rows in the products' 8-bit form
            ck(lib.nerf_amd_param_gradients_convert_e4m3(ptr(self.posx), ptr(self.posd), None, ptr(self.scratch8), P, 1, ss),
               "nerf_amd_param_gradients_convert_e4m3")
        ck(lib.nerf_amd_mse_loss(ptr(self.rgb), ptr(self.gt), ptr(self.loss), None, B * 3, ss), "nerf_amd_mse_loss")
        ck(lib.nerf_amd_param_gradients_begin(ptr(self.d_raw), ptr(self.scratch), ptr(self.grads), P, ss),
           "nerf_amd_param_gradients_begin")
        if self._e4m3:       # the packed d_raw likewise
            ck(lib.nerf_amd_param_gradients_convert_e4m3(None, None, ptr(self.scratch), ptr(self.scratch8), P, 2, ss),
               "nerf_amd_param_gradients_convert_e4m3")
        if self.rays_from is not None and self.device_rng:
            # rg.select + the colour gather (train.py:47-49) for the NEXT step, beside the dX chain: this step's rays and
            # colours have been read for the last time (encoder rows, forward, compositor, loss), and the selection depends
            # on the step counter only (device memory: every replay selects the batch of step + 1), never on the weights --
            # so the first lines of the next iteration cost the step nothing.  step() primes the first batch.
            import ctypes
            self.rays_from.launch(self.select_mode, B, None, self._select_seed(1), ctypes.c_void_p(self.hyper.data_ptr() + 24),
                                  self.rays, self.gt, self._ids_next, stream=ss, workspace=self._select_ws)
        main.wait_stream(side)
        self._finish(bucket, st)

    def _finish(self, bucket, st):
        """The dW products (all, or one bucket's) from the saved tensors in this step's storage form."""
        lib, ptr, P = _lib.lib(), _lib.ptr, self.B * self.N
        if self._e4m3:
            _lib.check(lib.nerf_amd_param_gradients_finish_e4m3(ptr(self.acts), ptr(self.dys), ptr(self.scratch8), ptr(self.grads),
                                                                P, bucket, st), "nerf_amd_param_gradients_finish_e4m3")
        else:
            _lib.check(lib.nerf_amd_param_gradients_finish_bucket(ptr(self.acts), ptr(self.dys), ptr(self.posx), ptr(self.posd),
                                                                  ptr(self.scratch), ptr(self.grads), P, bucket, st),
                       "nerf_amd_param_gradients_finish_bucket")

    def _select_seed(self, offset=0):
        """The seed argument of nerf_amd_select_rays for the batch ``offset`` steps after the one the step counter in
        device memory names: the kernel forms (seed ^ KEY) + counter, so the offset goes inside the key.  Replicas
        (distinct ray_id0) draw distinct batches."""
        from .utils.dataload import SELECT_KEY
        m = 0xffffffffffffffff
        base = (self.seed ^ (self.ray_id0 * 0x9E3779B97F4A7C15)) & m
        return ((((base ^ SELECT_KEY) + int(offset)) & m) ^ SELECT_KEY) & m

    def _select_now(self, step, rays, gt, ids):
        """The batch of optimisation step ``step`` (1-based), eagerly: what the graph's prefetch produces one step ahead."""
        self.rays_from.launch(self.select_mode, self.B, None, self._select_seed(step), None, rays, gt, ids, workspace=self._select_ws2)

    @property
    def ray_ids(self):
        """Rows of the table the LAST step trained on (rays_from).  With the selection inside the graph the ids buffer
        already holds the next batch's, so these are recomputed on demand (ids only: three small launches)."""
        if self.rays_from is not None and self.device_rng:
            if self._ids_step != self.opt.step_count:
                self._select_now(self.opt.step_count, None, None, self._ids_cur)
                self._ids_step = self.opt.step_count
            return self._ids_cur
        return self._ids_next

    def _head_gradients(self):
        """The second launch of the bucketed form: the products of layers_0.* (bucket 2)."""
        self._finish(2, _lib.stream_ptr(self.dev))

    def _update(self):
        lib, opt = _lib.lib(), self.opt
        _lib.check(lib.nerf_amd_adam_step_hyper(_lib.ptr(opt.flat), _lib.ptr(self.grads), _lib.ptr(opt.exp_avg),
                                                _lib.ptr(opt.exp_avg_sq), opt.flat.numel(), _lib.ptr(self.hyper),
                                                _lib.stream_ptr(self.dev)), "nerf_amd_adam_step_hyper")
        self.net.repack_from_flat(opt.flat)

    def _set_hyper(self, step):
        pg = self.opt.param_groups[0]
        b1, b2 = float(pg["betas"][0]), float(pg["betas"][1])
        self._ring.push((float(pg["lr"]), b1, b2, float(pg["eps"]), 1.0 - b1 ** step, (1.0 - b2 ** step) ** 0.5),
                        seed_offset=step)

    def _capture(self):
        with torch.cuda.device(self.dev):
            # both images must exist (and be cached) before capture: packing allocates.  Their addresses are baked into
            # the graphs, so this object owns them from here on (_own_images): the module's cache must never replace them
            self._packed_fwd = self.net.packed_weights(_lib.BF16)
            self._packed_bwd = self.net.packed_weights(_lib.BF16_BWD)
            self._set_hyper(1)
            params0 = self.opt.flat.clone()
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):                       # warm-up outside capture (lazy inits)
                self._forward_backward()
                self._head_gradients()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a):
                self._forward_backward(1 if self.bucketed else 0)
            self.graph_a2 = None
            if self.bucketed:
                self.graph_a2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_a2):
                    self._head_gradients()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b):
                self._update()
            # Without an exchange nothing has to happen between the two on most steps: one graph for the whole iteration
            # saves a launch seam (~5-10 us of idle GPU).  Steps that put a status copy / range check between forward and
            # re-pack (check_every) replay the two separate graphs.
            self.graph_ab = None
            if not self.exchange:
                self.graph_ab = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_ab):
                    self._forward_backward(0)
                    self._update()
            # capture executed nothing, and the warm-up did not touch the parameters
            assert torch.equal(self.opt.flat, params0)
            self._ring.reset()                                   # the warm-up consumed a slot
            # a fresh pair of training images: whatever the warm-up left in their status words is gone
            self._own_images(force=True)

    def _own_images(self, force=False):
        """The two training images the graphs read and re-pack are THIS object's buffers.  If the parameters moved behind
        the graphs' back since the last step (net.load_state_dict to restore a checkpoint, any in-place torch op: their
        versions tell), the module's cache would pack NEW buffers on its next query and free these -- while every replay
        still reads and rewrites them.  So: re-pack from the flat vector (the parameters are views of it) into the
        captured buffers, as new weights (status words cleared), and put exactly these buffers back into the cache."""
        from .utils.nets import _Packed
        net, dev = self.net, self.dev
        params = net._param_list()
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        ents = [net._packed.get((dev, c)) for c in (_lib.BF16, _lib.BF16_BWD)]
        bufs = (self._packed_fwd, self._packed_bwd)
        if not force and all(e is not None and e.stamp == stamp and e.buf is b for e, b in zip(ents, bufs)):
            return
        off = 0
        for p in params:                                   # FusedAdam made them views of its flat vector; still true?
            if p.data_ptr() != self.opt.flat.data_ptr() + 4 * off:
                raise RuntimeError("a parameter no longer lives in the optimizer's flat vector (its .data was replaced): "
                                   "build a new FusedAdam and GraphedTrainStep")
            off += p.numel()
        lib = _lib.lib()
        with torch.cuda.device(dev):
            for code, buf in zip((_lib.BF16, _lib.BF16_BWD), bufs):
                _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(self.opt.flat), _lib.ptr(buf), code, _lib.stream_ptr(dev)),
                           "nerf_amd_pack_weights")
                net._packed[(dev, code)] = _Packed(stamp, buf)
        net.drop_packed(dev, keep=(_lib.BF16, _lib.BF16_BWD))

    # ---- one iteration -------------------------------------------------------------
    def step(self, rays=None, gt=None, u=None, decay=1.0):
        from . import parallel
        if (rays is None) != (gt is None):
            raise RuntimeError("step(): rays and gt come together (or neither, with rays_from)")
        if rays is None and self.rays_from is None:
            raise RuntimeError("step() without rays needs GraphedTrainStep(..., rays_from=<utils.dataload.RayGenerator>)")
        if rays is not None and (rays.shape != self.rays.shape or gt.shape != self.gt.shape):
            raise RuntimeError(f"GraphedTrainStep was captured for rays {tuple(self.rays.shape)}, gt {tuple(self.gt.shape)}")
        bad = self._watch.poll()
        if bad is not None:
            what = " and ".join(w for w, on in (("activations", bad[1]), ("weights", bad[2])) if on)
            raise FloatingPointError(f"non-finite values inside the network in training step {bad[0]} ({what}): "
                                     "NaN / inf weights or inputs, the run has diverged")
        self._own_images()
        if rays is None and self.device_rng and self._primed_for != self.opt.step_count + 1:
            # the first step (or one after the step counter was moved by hand): the graph prefetches batch k + 1 while
            # step k runs, so batch k has to be there before the first replay
            self._select_now(self.opt.step_count + 1, self.rays, self.gt, self._ids_next)
        if rays is not None:
            if self.rays_from is not None and self.device_rng:
                raise RuntimeError("this GraphedTrainStep selects its rays inside the captured graph (rays_from, device_rng=True): "
                                   "step() takes no rays")
            self.rays.copy_(rays, non_blocking=True)
            self.gt.copy_(gt, non_blocking=True)
        session = None
        try:
            if self.device_rng:
                if u is not None:
                    raise RuntimeError("this GraphedTrainStep draws its jitter on the device (device_rng=True): u must be None")
            else:
                from .utils import host_rng
                if rays is None and host_rng.host_fallback():
                    self.rays_from.select_batch(self.select_mode, self.B, out=(self.rays, self.gt, self._ids_next))
                elif rays is None:
                    # the reference's iteration on torch's CPU stream, continued on the device: randperm(n)[:B] (the n - 1 - B
                    # draws nobody looks at are jumped over), the two gathers, then -- same stream -- the jitter draw
                    session = host_rng.GeneratorSession(self.dev)
                    drawn = self.rays_from.select_from_session(session, self.select_mode, self.B, self.rays, self.gt, self._ids_next,
                                                               workspace=self._select_ws,
                                                               jitter=(self.B, self.N, self.u) if u is None else None)
                    if drawn is not None:
                        u = self.u                       # the jitter came with the selection (one jump launch for both)
                if u is self.u:
                    pass
                elif u is None:
                    # the reference's one draw per call from torch's CPU generator, continued on the device; the generator is
                    # made current again (session.finish: a wait for the generator kernels alone) once the whole step is
                    # enqueued behind it, so the host never waits for the previous step here
                    if host_rng.host_fallback():
                        self.u.copy_(torch.rand(self.B, self.N), non_blocking=False)
                    else:
                        session = session or host_rng.GeneratorSession(self.dev)
                        session.rand(self.B, self.N, out=self.u)
                else:
                    self.u.copy_(u, non_blocking=True)
            return self._enqueue_step(decay)
        finally:
            if session is not None:
                session.finish()

    def _enqueue_step(self, decay):
        from . import parallel
        self.opt.step_count += 1
        self._set_hyper(self.opt.step_count)
        watch = bool(self.check_every) and self.opt.step_count % self.check_every == 0
        whole = self.graph_ab is not None and not watch
        (self.graph_ab if whole else self.graph_a).replay()
        self._ring.launched(self.dev)
        if self.rays_from is not None and self.device_rng:
            self._primed_for = self.opt.step_count + 1          # graph A left the next step's batch in the buffers
        if watch:
            # behind the forward, in front of graph B's re-pack (which clears the flag for the next step)
            self._watch.push(self._packed_fwd, self.opt.step_count)
            # the reference's |x| > 1 warning (utils/xyz.py:8-9) on the batch in the buffers (with the selection inside the
            # graph that is already the next step's), verdict raised lazily
            from .utils.xyz import range_check_rays
            if self.device_rng:
                import ctypes
                range_check_rays(self.rays, ctypes.c_void_p(self.hyper.data_ptr() + 24), self.tbins,
                                 _lib.FLAG_DEVICE_RNG | _lib.FLAG_SEED_IN_MEMORY, self.seed, self.ray_id0, self.N)
            else:
                range_check_rays(self.rays, self.u, self.tbins, 0, 0, 0, self.N)
        if self.bucketed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.timing else None
            if ev:
                ev[0].record()                                           # late gradients done, exchange starts
            h1 = parallel.allreduce_start_(self.buckets[0], group=self.group)
            self.graph_a2.replay()                                       # head gradients, beside the first exchange
            if ev:
                ev[1].record()
            h2 = parallel.allreduce_start_(self.buckets[1], group=self.group)
            parallel.allreduce_wait_(h1)
            parallel.allreduce_wait_(h2)
            if ev:
                ev[2].record()
                self._events.append(ev)
        elif self.exchange:
            # buckets=1: the whole 2.38 MB vector in one all-reduce between the two graphs, fully exposed
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if self.timing else None
            if ev:
                ev[0].record()
            parallel.allreduce_flat_(self.grads, group=self.group)
            if ev:
                ev[1].record()
                self._events.append([ev[0], ev[0], ev[1]])
        if not whole:
            self.graph_b.replay()
        # graph B re-packed the two training images; any other image of the module (fp16 / fp32 inference) is now
        # stale -- and the kernels wrote through the flat buffer, so the parameters' versions did not move
        self.net.drop_packed(self.dev, keep=(_lib.BF16, _lib.BF16_BWD))
        if decay != 1.0:
            for pg in self.opt.param_groups:
                pg["lr"] = pg["lr"] * decay
        return self.loss

    __call__ = step

    def reset_timing(self):
        """Forget the exchange events recorded so far (warm-up steps: the first collective creates the communicator)."""
        self._events = []

    def collective_times(self):
        """(span_ms, exposed_ms) averaged over the steps recorded with timing=True: from the end of the late-layer
        gradients to both buckets reduced, and the part of it behind the end of the head-gradient launch (what the
        step actually waits for).  Synchronises; clears the record."""
        if not self._events:
            return None
        torch.cuda.synchronize(self.dev)
        span = sum(e[0].elapsed_time(e[2]) for e in self._events) / len(self._events)
        exposed = sum(e[1].elapsed_time(e[2]) for e in self._events) / len(self._events)
        self._events = []
        return span, exposed
