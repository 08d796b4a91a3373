"""Training-side entry points: the step of reference train.py:47-57 on the GPU.

What runs where in a training step (BASELINE config 5), 16-bit precision (default):

  sampling + encoding + 12 dense layers, forward    HIP  nerf_amd_mlp_forward_train (the fused
                                                    inference kernel, also saving bf16 activations)
  sigma -> alpha compositing, forward               HIP  nerf_amd_volume_render
  compositing, backward (suffix-sum scan)           HIP  nerf_amd_volume_render_backward
  dense layers, backward dX chain (on-chip)         HIP  nerf_amd_mlp_backward
  dense layers, dW = dY^T X and db = sum dY         plain GEMMs / reductions over the point
                                                    dimension: vendor library (torch.mm)
  gradient exchange                                 RCCL all-reduce of one flat bucket (parallel.py)
  optimizer                                         torch.optim.Adam (reference train.py:43)

precision='fp32' keeps everything in fp32: HIP sampling/encoding and compositor,
the dense layers through torch.nn.functional.linear under autograd (library
GEMMs); that path is the one pinned bit-tight against golden G6.  Inference never
comes here: without gradients the fused forward kernel runs.
"""
import os

import torch
import torch.nn.functional as F

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
# compositor with a HIP backward
# --------------------------------------------------------------------------
class _VolumeRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, ts, dirs):
        B, N = raw.shape[0], raw.shape[1]
        dev = raw.device
        raw, ts, dirs = raw.contiguous(), ts.contiguous(), dirs.contiguous()
        rgb = torch.empty((B, 3), dtype=torch.float32, device=dev)
        disp = torch.empty((B,), dtype=torch.float32, device=dev)
        acc = torch.empty((B,), dtype=torch.float32, device=dev)
        alpha = torch.empty((B, N), dtype=torch.float32, device=dev)
        w = torch.empty((B, N), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nerf_amd_volume_render(
                _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), 3, _lib.ptr(rgb), _lib.ptr(disp),
                _lib.ptr(alpha), _lib.ptr(acc), _lib.ptr(w), B, N, _lib.stream_ptr(dev)),
                "nerf_amd_volume_render")
        ctx.save_for_backward(raw, ts, dirs)
        return rgb, disp, alpha, acc, w

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_alpha, g_acc, g_w):
        raw, ts, dirs = ctx.saved_tensors
        B, N = raw.shape[0], raw.shape[1]
        dev = raw.device
        d_raw = torch.empty_like(raw)

        def c(g):
            return None if g is None else g.contiguous().float()
        g_rgb, g_disp, g_alpha, g_acc, g_w = map(c, (g_rgb, g_disp, g_alpha, g_acc, g_w))
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nerf_amd_volume_render_backward(
                _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), 3, _lib.ptr(g_rgb), _lib.ptr(g_disp),
                _lib.ptr(g_alpha), _lib.ptr(g_acc), _lib.ptr(g_w), _lib.ptr(d_raw), B, N,
                _lib.stream_ptr(dev)), "nerf_amd_volume_render_backward")
        return d_raw, None, None


def volume_render_autograd(nerf_outs, ts, dirs):
    """volume_render (reference utils/rendering.py:47-85) with gradients to
    nerf_outs; ts and dirs get none (they carry none in the reference either)."""
    return _VolumeRender.apply(nerf_outs.float(), ts.detach(), dirs.detach())


# --------------------------------------------------------------------------
# the dense layers under autograd (library GEMMs)
# --------------------------------------------------------------------------
def _dense_layers(net, x, d, precision):
    """Data-flow of reference utils/nets.py:37-43 on already-encoded inputs,
    through the module's own nn.Linear parameters so autograd reaches them."""
    # 16-bit modes run the library GEMMs under bf16 autocast (fp16 would need loss scaling)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(_lib.precision_code(precision) != _lib.F32)):
        h = net.layers_0(x)
        h = net.skip_conn_layer(torch.cat([h, x.to(h.dtype)], dim=1))
        h = net.layers_1(h)
        sigma = net.sigma_fc(h)
        h = net.layers_2(h)
        rgb = net.color_fc(torch.cat([h, d.to(h.dtype)], dim=1))
        return torch.cat([rgb, sigma], dim=1).float()


def nerf_forward_autograd(net, v, precision):
    """Nerf.forward with gradients to the parameters: HIP encoder + library GEMMs."""
    from .utils.xyz import positional_encoder
    x, d = positional_encoder(v.detach(), net.Lp, net.Ld)
    return _dense_layers(net, x, d, precision)


# --------------------------------------------------------------------------
# fused dense layers: HIP forward (saving activations) + HIP dX chain + library dW
# --------------------------------------------------------------------------
class _FusedDense(torch.autograd.Function):
    """(rays, jitter) -> raw [B,N,4], ts [B,N] through nerf_amd_mlp_forward_train;
    backward: nerf_amd_mlp_backward for every layer's pre-activation gradient, then
    nerf_amd_param_gradients (split-K GEMMs + column sums) into ONE flat fp32
    vector in state_dict order, handed back to autograd as 24 views."""

    @staticmethod
    def forward(ctx, net, rays, jit, tbins, flags, seed, ray_id0, N, *params):
        lib = _lib.lib()
        B, dev = rays.size(0), rays.device
        P = B * N
        packed = net.packed_weights(_lib.BF16)
        raw = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        ts = torch.empty((B, N), dtype=torch.float32, device=dev)
        acts = torch.empty(int(lib.nerf_amd_train_activation_bytes(P)), dtype=torch.uint8, device=dev)
        posx = torch.empty((P, 64), dtype=torch.bfloat16, device=dev)
        posd = torch.empty((P, 32), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(lib.nerf_amd_mlp_forward_train(
                _lib.ptr(rays), _lib.ptr(jit), _lib.ptr(tbins), _lib.ptr(packed), flags, int(seed), int(ray_id0),
                _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(acts), B, N, st), "nerf_amd_mlp_forward_train")
            # encoder outputs in the reference's column order: the inputs of the dW products of
            # layers_0.0 / skip_conn_layer / color_fc.0 (same sample positions: ts given)
            _lib.check(lib.nerf_amd_sample_encode_bf16(
                _lib.ptr(rays), _lib.ptr(ts), None, _lib.FLAG_TS_GIVEN, 0, 0,
                _lib.ptr(posx), _lib.ptr(posd), None, B, N, st), "nerf_amd_sample_encode_bf16")
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
        return (None,) * 8 + tuple(grads)


def _fused_training_enabled(precision):
    return _lib.precision_code(precision) != _lib.F32 and os.environ.get("NERF_AMD_TRAIN_FUSED", "1") != "0"


def render_nerf_autograd(rays, net, N, tn, tf, jit, flags, precision, seed, ray_id0):
    """render_nerf (reference utils/rendering.py:13-45) with gradients to the
    parameters of ``net``; returns the same 5-tuple."""
    from .utils.rendering import _tbins
    B, dev = rays.size(0), rays.device
    if _fused_training_enabled(precision):
        params = [p for _, p in net.named_parameters()]
        raw, ts = _FusedDense.apply(net, rays, jit, _tbins(tn, tf, N, dev), flags, seed, ray_id0, N, *params)
        dn = rays[:, 3:] / torch.norm(rays[:, 3:], dim=1, keepdim=True)
        return _VolumeRender.apply(raw, ts, dn)
    lib = _lib.lib()
    P = B * N
    posx = torch.empty((P, 63), dtype=torch.float32, device=dev)
    posd = torch.empty((P, 27), dtype=torch.float32, device=dev)
    ts = torch.empty((B, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nerf_amd_sample_encode(
            _lib.ptr(rays), _lib.ptr(jit), _lib.ptr(_tbins(tn, tf, N, dev)), flags, int(seed), int(ray_id0),
            _lib.ptr(posx), _lib.ptr(posd), _lib.ptr(ts), B, N, _lib.stream_ptr(dev)),
            "nerf_amd_sample_encode")
    out = _dense_layers(net, posx, posd, precision).reshape(B, N, 4)
    dn = rays[:, 3:] / torch.norm(rays[:, 3:], dim=1, keepdim=True)
    return _VolumeRender.apply(out, ts, dn)


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
    Only ``rgb`` feeds the loss, as in the reference (train.py:52)."""
    from . import parallel
    from .utils.rendering import render_nerf
    optimizer.zero_grad(set_to_none=True)
    rgb, _, _, _, _ = render_nerf(rays, net, N, tn, tf, u=u, precision=precision,
                                  device_rng=device_rng, seed=seed, ray_id0=ray_id0)
    loss = F.mse_loss(rgb, gt)
    loss.backward()
    parallel.allreduce_gradients(net.parameters(), group=group)
    optimizer.step()
    if decay != 1.0:
        for pg in optimizer.param_groups:
            pg["lr"] = pg["lr"] * decay
    return loss.detach()
