"""Drop-in for the hot path of the reference's utils/rendering.py.

``render_nerf`` (alias ``render_rays``) and ``volume_render`` keep the
reference's names, positional arguments, defaults and return order
(reference utils/rendering.py:13,47); the bodies are HIP kernels behind
libnerf_amd.so.  ``render_image`` / ``render_poses`` reproduce the batching and
clipping semantics of the reference's image drivers (utils/rendering.py:88-153)
without the video writer.

Jitter: in parity mode (default) ``render_nerf`` draws exactly one
``torch.rand(B, N)`` from the CPU default generator per call, as the reference
does (utils/rendering.py:28), so a seeded run consumes the RNG identically.
``u=`` / ``ts=`` inject explicit jitter / sample positions; ``device_rng=True``
draws jitter from a counter RNG on the GPU instead (no PCIe copy, results
independent of batching and sharding, not bit-comparable to the reference).
"""
import torch
from tqdm import tqdm

from .. import _lib
from .host_rng import ReferenceJitter, reference_rand
from .nets import Nerf, guarded_launch

ALL_OUTPUTS = ("rgb", "disp", "alpha", "acc", "w")
_FUSED_MAX_N = 768          # csrc/nerf_layout.h FUSED_RENDER_MAX_N: rays the one-launch render composites in its LDS ring
_tbins_cache = {}


def _tbins(tn, tf, N, device):
    key = (float(tn), float(tf), int(N), device)
    t = _tbins_cache.get(key)
    if t is None:
        # computed on the host by torch.linspace so bin edges are bit-identical
        # to the reference's (utils/rendering.py:25)
        t = torch.linspace(tn, tf, N + 1).to(device)
        if len(_tbins_cache) > 64:
            _tbins_cache.clear()
        _tbins_cache[key] = t
    return t


def _per_sample(t_, N):
    """alpha / w as the reference shapes them: [B,N], except [B,0] at N == 1, where its delta construction
    leaves the sample axis empty (utils/rendering.py:60-61; the kernels reproduce rgb = acc = 0, disparity = NaN)."""
    return t_ if (t_ is None or N != 1) else t_[:, :0]


def volume_render(nerf_outs, ts, dirs, *, outputs=ALL_OUTPUTS):
    """nerf_outs [B,N,4], ts [B,N], dirs [B,3] -> (rgb [B,3], disp [B], alpha [B,N],
    acc [B], w [B,N])  (reference utils/rendering.py:47-85).  The second output is
    disparity; acc == 0 yields NaN there, as in the reference."""
    _lib.require_cuda_f32(nerf_outs, "nerf_outs")
    _lib.require_cuda_f32(ts, "ts")
    _lib.require_cuda_f32(dirs, "dirs")
    if nerf_outs.dim() != 3 or nerf_outs.shape[-1] != 4:
        raise RuntimeError("nerf_outs must be [B, N, 4]")
    B, N = nerf_outs.shape[0], nerf_outs.shape[1]
    if tuple(ts.shape) != (B, N) or tuple(dirs.shape) != (B, 3):
        raise RuntimeError("ts must be [B, N] and dirs [B, 3]")
    if torch.is_grad_enabled() and nerf_outs.requires_grad:
        from ..training import volume_render_autograd
        return volume_render_autograd(nerf_outs, ts, dirs)
    dev = nerf_outs.device
    raw, ts, dirs = nerf_outs.detach().contiguous(), ts.detach().contiguous(), dirs.detach().contiguous()
    rgb = torch.empty((B, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((B,), dtype=torch.float32, device=dev)
    acc = torch.empty((B,), dtype=torch.float32, device=dev)
    alpha = torch.empty((B, N), dtype=torch.float32, device=dev) if "alpha" in outputs else None
    w = torch.empty((B, N), dtype=torch.float32, device=dev) if "w" in outputs else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nerf_amd_volume_render(
            _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(dirs), 3, _lib.ptr(rgb), _lib.ptr(disp),
            _lib.ptr(alpha), _lib.ptr(acc), _lib.ptr(w), B, N, _lib.stream_ptr(dev)),
            "nerf_amd_volume_render")
    return rgb, disp, _per_sample(alpha, N), acc, _per_sample(w, N)


def render_nerf(rays, net, N, tn=2, tf=6, *, u=None, ts=None, outputs=ALL_OUTPUTS,
                precision=None, device_rng=False, seed=0, ray_id0=0):
    """Stratified sampling along rays, NeRF query, compositing
    (reference utils/rendering.py:13-45).

    rays [B,6] = [origin, direction] on the GPU; net: a ``Nerf`` (fused HIP path)
    or any object with ``.forward(query_pts[P,6]) -> [P,4]`` (generic path: the
    net runs as given, sampling and compositing still run here).
    Returns (rgb [B,3], disp [B], alpha [B,N], acc [B], w [B,N]); alpha / w are
    None when left out of ``outputs``.
    """
    _lib.require_cuda_f32(rays, "rays")
    if rays.dim() != 2 or rays.shape[1] != 6:
        raise RuntimeError("rays must be [B, 6]")
    B, N = rays.size(0), int(N)
    dev = rays.device
    rays = rays.detach().contiguous()

    fused = isinstance(net, Nerf) and net._fused_ok()
    training = fused and torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters())
    # everything that can raise cheaply is checked BEFORE the jitter is drawn, so a failed call
    # leaves torch's CPU generator untouched
    for name, t_ in (("ts", ts), ("u", u)):
        if t_ is not None and tuple(_lib.require_cuda_f32(t_, name).shape) != (B, N):
            raise RuntimeError("u / ts must be [B, N]")
    code = None
    if fused and not training:
        code = _lib.precision_code(net.precision if precision is None else precision)
        net.packed_weights(code)              # packs now if it has to: errors surface before the jitter is drawn

    flags, jit, pending_rng = 0, None, None
    if ts is not None:
        jit, flags = ts.contiguous(), _lib.FLAG_TS_GIVEN
    elif u is not None:
        jit = u.contiguous()
    elif device_rng:
        flags = _lib.FLAG_DEVICE_RNG
    else:
        # the reference's single CPU draw per call (:28-30): same numbers, same advance of torch's
        # CPU generator, produced on the device (host_rng.py)
        jit, pending_rng = reference_rand(B, N, dev)
    # the reference's |x| > 1 warning (utils/xyz.py:8-9, raised inside net.forward's positional_encoder): verdict formed on
    # the device from the first / last sample of every ray, raised lazily (xyz.range_check_rays); a foreign net encodes
    # its own inputs and warns (or not) by itself
    if fused:
        from .xyz import range_check_rays
        range_check_rays(rays, jit, None if ts is not None else _tbins(tn, tf, N, dev), flags, seed, ray_id0, N)
    if pending_rng is not None and not fused:
        pending_rng.finish()                  # the net's own forward follows: the generator must be current
        pending_rng = None
    if training:
        from ..training import render_nerf_autograd
        try:
            return render_nerf_autograd(rays, net, N, tn, tf, jit, flags,
                                        net.precision if precision is None else precision,
                                        seed, ray_id0)
        finally:
            if pending_rng is not None:
                pending_rng.finish()          # the forward is enqueued behind the generator kernel: only that is awaited
    if not fused:
        return _render_generic(rays, net, N, tn, tf, jit, flags, outputs, seed, ray_id0)

    def launch(code, packed):
        lib = _lib.lib()
        rgb = torch.empty((B, 3), dtype=torch.float32, device=dev)
        disp = torch.empty((B,), dtype=torch.float32, device=dev)
        acc = torch.empty((B,), dtype=torch.float32, device=dev)
        alpha = torch.empty((B, N), dtype=torch.float32, device=dev) if "alpha" in outputs else None
        w = torch.empty((B, N), dtype=torch.float32, device=dev) if "w" in outputs else None
        nws = int(lib.nerf_amd_render_workspace_bytes(code, B, N))        # 0: the fused one-launch render
        ws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws else None
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_render_forward(
                _lib.ptr(rays), _lib.ptr(jit), _lib.ptr(_tbins(tn, tf, N, dev)), _lib.ptr(packed[0]), code,
                flags, int(seed), int(ray_id0), _lib.ptr(rgb), _lib.ptr(disp), _lib.ptr(alpha),
                _lib.ptr(acc), _lib.ptr(w), _lib.ptr(ws), B, N, _lib.stream_ptr(dev)), "nerf_amd_render_forward")
        return rgb, disp, _per_sample(alpha, N), acc, _per_sample(w, N)

    try:
        return guarded_launch([net], code, launch)     # fp16: range guard (nets.guarded_launch)
    finally:
        if pending_rng is not None:
            pending_rng.finish()              # the render is enqueued behind it: only the generator kernel is awaited


render_rays = render_nerf      # the name BASELINE.json uses for the same function


def _render_generic(rays, net, N, tn, tf, jit, flags, outputs, seed, ray_id0):
    """Any other net object: sampling and query-point assembly in one HIP kernel (nerf_amd_query_points =
    utils/rendering.py:24-40; the same jitter sources as the fused path, counter RNG included), the net's own
    forward exactly as the reference calls it (:41), compositing by the HIP kernel with the directions taken
    from the rays (:37,43).  Gradients flow to whatever ``net.forward`` attaches to its output."""
    B, dev = rays.size(0), rays.device
    lib = _lib.lib()
    q = torch.empty((B * N, 6), dtype=torch.float32, device=dev)
    ts = torch.empty((B, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nerf_amd_query_points(_lib.ptr(rays), _lib.ptr(jit), _lib.ptr(_tbins(tn, tf, N, dev)), flags,
                                             int(seed), int(ray_id0), _lib.ptr(q), _lib.ptr(ts), B, N,
                                             _lib.stream_ptr(dev)), "nerf_amd_query_points")
    out = net.forward(q).reshape(B, N, 4).float()
    if torch.is_grad_enabled() and out.requires_grad:
        from ..training import _VolumeRender
        return _VolumeRender.apply(out, ts, rays, True)
    out = out.detach().contiguous()
    rgb = torch.empty((B, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((B,), dtype=torch.float32, device=dev)
    acc = torch.empty((B,), dtype=torch.float32, device=dev)
    alpha = torch.empty((B, N), dtype=torch.float32, device=dev) if "alpha" in outputs else None
    w = torch.empty((B, N), dtype=torch.float32, device=dev) if "w" in outputs else None
    with torch.cuda.device(dev):
        _lib.check(lib.nerf_amd_volume_render_rays(_lib.ptr(out), _lib.ptr(ts), _lib.ptr(rays), _lib.ptr(rgb), _lib.ptr(disp),
                                                   _lib.ptr(alpha), _lib.ptr(acc), _lib.ptr(w), B, N, _lib.stream_ptr(dev)),
                   "nerf_amd_volume_render_rays")
    return rgb, disp, _per_sample(alpha, N), acc, _per_sample(w, N)


def _render_batched(rays, net, batch_size, N, tn, tf, u, progress, id_base=0, **kw):
    """Shared body of the image drivers (reference utils/rendering.py:98-108,
    139-151): per batch render_nerf under no_grad, clip rgb to [0,1] AFTER
    compositing, disparity un-clipped.  Unlike the reference's
    ``range(n // batch_size)`` the tail batch is rendered too."""
    n = rays.size(0)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=rays.device)
    disp = torch.empty((n,), dtype=torch.float32, device=rays.device)
    if isinstance(net, Nerf) and net._fused_ok() and rays.is_cuda and 0 < n and N <= _FUSED_MAX_N:
        # The reference's batch_size bounds the [batch, N, ...] tensors of its per-sample path.  The fused render keeps
        # nothing per sample in HBM and its pixels do not depend on how the rays are batched (jitter rows follow the rays;
        # bit for bit: tests/test_gpu_parity.py test_headline_workload_properties), so the image is ONE launch: 40 launches
        # of 16,000 rays each end in a partly filled wave of tiles (+1.6 % at 800 x 800).  Longer rays (two-launch path,
        # 20 B per sample of workspace) and foreign nets keep the caller's batches.
        batch_size = n
    starts = range(0, n, batch_size)
    # The reference draws each batch's jitter inside render_nerf from the CPU generator: consecutive
    # pieces of one stream.  For the fused path they are all enqueued now on a side stream, so
    # batch k+1 is drawn while batch k renders (host_rng.ReferenceJitter); a generic ``net`` keeps
    # the per-call draw, since its forward may use the generator itself.
    ahead = None
    if (u is None and not kw.get("device_rng") and n > 0 and rays.is_cuda
            and isinstance(net, Nerf) and net._fused_ok()):
        ahead = ReferenceJitter([min(s + batch_size, n) - s for s in starts], N, rays.device)
    try:
        with torch.no_grad():
            for k, s in enumerate(tqdm(starts) if progress else starts):
                e = min(s + batch_size, n)
                uk = ahead.batch(k) if ahead is not None else (None if u is None else u[s:e])
                r, d, _, _, _ = render_nerf(rays[s:e], net, N, tn, tf, u=uk,
                                            outputs=("rgb", "disp", "acc"), ray_id0=id_base + s, **kw)
                rgb[s:e] = torch.clip(r, 0., 1.)
                disp[s:e] = d
    finally:
        if ahead is not None:
            ahead.finish()
    return rgb, disp


def render_image(net, rg, batch_size=64000, im_idx=0, im_set='val', *, N=128, tn=2, tf=6,
                 u=None, progress=False, **kw):
    """Render image ``im_idx`` of ``rg``'s ``im_set`` (reference utils/rendering.py:88-113).
    ``rg`` is any object with the reference RayGenerator's fields
    ``samples[im_set][i]['img']`` and ``rays_dataset[im_set]`` ([n_img*H*W, 6]).
    Returns (rgb [1,H,W,3], disparity [1,H,W,1], gt [1,H,W,3]) on the CPU."""
    gt = rg.samples[im_set][im_idx]['img']
    H, W = gt.shape[0], gt.shape[1]
    dev = next(net.parameters()).device
    rays = rg.rays_dataset[im_set][im_idx * H * W:(im_idx + 1) * H * W, :].to(dev).float()
    rgb, disp = _render_batched(rays, net, batch_size, N, tn, tf, u, progress, **kw)
    return rgb.cpu().reshape(1, H, W, 3), disp.cpu().reshape(1, H, W, 1), gt.reshape(1, H, W, 3)


def render_poses(net, poses, cam_params, batch_size, savepath='', *, N=128, tn=2, tf=6,
                 u=None, progress=False, **kw):
    """Render one image per pose (reference utils/rendering.py:116-153).
    poses: list of [4,4] float tensors; cam_params [H,W,f].  Returns
    (rgb_imgs, disp_imgs): lists of numpy [H,W,3] / [H,W].  The reference's mp4
    writer (cv2, :155-160) is out of scope; ``savepath`` is accepted and ignored.

    The reference builds the ray table of every pose on the CPU (:129-134) and moves it to the GPU batch
    by batch; at 800x800 that table takes 0.2 s per pose on 8 cores -- three times the render -- and
    24 B per ray of PCIe.  Here each pose's rays come from the device generator (generate_rays, N1:
    the same table to the fma rounding of the 3-term rotation, 2.4e-7 absolute)."""
    H, W = cam_params[0], cam_params[1]
    dev = next(net.parameters()).device
    # The reference reads every image back as soon as it is rendered (:150-151, a blocking copy).  Here the copy of
    # pose i runs on a side stream into pinned memory while pose i+1 renders; the arrays are handed out after the last
    # copy has landed (the returned numpy arrays view that pinned memory).
    copier = torch.cuda.Stream(device=dev)
    rgb_host, disp_host = [], []
    for i in range(len(poses)):
        pose = poses[i].detach().cpu() if torch.is_tensor(poses[i]) else poses[i]
        rays = generate_rays(pose, cam_params, dev)
        ui = None if u is None else u[i * H * W:(i + 1) * H * W]
        rgb, disp = _render_batched(rays, net, batch_size, N, tn, tf, ui, progress,
                                    id_base=i * H * W, **kw)
        rgb_h = torch.empty((H, W, 3), dtype=torch.float32).pin_memory()
        disp_h = torch.empty((H, W), dtype=torch.float32).pin_memory()
        copier.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(copier):
            rgb_h.copy_(rgb.view(H, W, 3), non_blocking=True)
            disp_h.copy_(disp.view(H, W), non_blocking=True)
        rgb.record_stream(copier)
        disp.record_stream(copier)
        rgb_host.append(rgb_h)
        disp_host.append(disp_h)
    copier.synchronize()
    return [x.numpy() for x in rgb_host], [x.numpy() for x in disp_host]


def render_rays_sharded(net, rays, batch_size, *, N=128, tn=2, tf=6, u=None, group=None, **kw):
    """Multi-GPU full-image render (BASELINE config 4's data path): every rank
    holds the same ray table [n,6] (any device), renders its contiguous share
    with the HIP path and all-gathers the packed [rgb, disparity] pixels
    (nerf_simple_amd.parallel).  Returns (rgb [n,3] clipped, disparity [n]) on
    every rank's GPU.  Jitter is indexed by global ray id, so the image does
    not depend on the number of ranks."""
    from .. import parallel
    dev = next(net.parameters()).device

    def render_fn(r, us, ray_id0):
        return _render_batched(r.to(dev).float().contiguous(), net, batch_size, N, tn, tf,
                               None if us is None else us.to(dev), False, id_base=ray_id0, **kw)

    return parallel.render_image_sharded(rays, render_fn, group=group, u=u)


def generate_rays(pose, cam_params, device, ray0=0, n_rays=None):
    """Pinhole rays of an HxW view on the GPU (reference utils/xyz.py:38-52 +
    utils/rendering.py:129-134 run on the CPU): pose [4,4] or [3,4] (host tensor /
    array), cam_params [H,W,f] -> rays [n_rays,6] for pixels ray0.. in row-major
    order."""
    import numpy as np
    H, W, f = int(cam_params[0]), int(cam_params[1]), float(cam_params[2])
    n = H * W - ray0 if n_rays is None else int(n_rays)
    h_pose = np.ascontiguousarray(np.asarray(pose, dtype=np.float32)[:3, :4])
    h_pose4 = np.zeros((3, 4), dtype=np.float32)
    h_pose4[:] = h_pose
    rays = torch.empty((n, 6), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().nerf_amd_generate_rays(
            h_pose4.ctypes.data, H, W, f, int(ray0), n, _lib.ptr(rays), _lib.stream_ptr(device)),
            "nerf_amd_generate_rays")
    return rays


def render_view(net, pose, cam_params, *, N=128, tn=2, tf=6, u=None, ray0=0, n_rays=None,
                precision=None, device_rng=False, seed=0):
    """One view (or the pixel range [ray0, ray0+n_rays) of it) in ONE library
    call: device ray generation -> render_nerf -> clip(rgb,0,1), i.e. the body of
    the reference's per-image loop (utils/rendering.py:139-151) without the
    Python batch loop.  Returns pixels [n,4] = [r,g,b,disparity] on the GPU.
    ``u`` [n,N] explicit jitter for these pixels; default: one CPU
    torch.rand(n,N) like the reference; device_rng=True: counter RNG keyed by
    global pixel id."""
    import numpy as np
    dev = next(net.parameters()).device
    H, W, f = int(cam_params[0]), int(cam_params[1]), float(cam_params[2])
    n = H * W - ray0 if n_rays is None else int(n_rays)
    if not (isinstance(net, Nerf) and net._fused_ok()):
        # a net the fused kernels are not built for (another Nerf(Lp, Ld, H), any module with .forward): device ray
        # generation, then the body of the reference's loop as it stands (render_nerf, clip)
        rays = generate_rays(pose, cam_params, dev, ray0, n)
        with torch.no_grad():
            rgb, disp, _, _, _ = render_nerf(rays, net, N, tn, tf, u=u, outputs=("rgb", "disp", "acc"),
                                             device_rng=device_rng, seed=seed, ray_id0=ray0)
        return torch.cat([rgb.clamp(0., 1.), disp[:, None]], dim=1)
    code = _lib.precision_code(net.precision if precision is None else precision)
    net.packed_weights(code)
    flags, jit = 0, None
    if u is not None:
        jit = _lib.require_cuda_f32(u, "u").contiguous()
    elif device_rng:
        flags = _lib.FLAG_DEVICE_RNG
    else:
        jit, pending_rng = reference_rand(n, N, dev)
        pending_rng.finish()
    lib = _lib.lib()
    h_pose = np.zeros((3, 4), dtype=np.float32)
    h_pose[:] = np.asarray(pose, dtype=np.float32)[:3, :4]

    def launch(code, packed):
        pixels = torch.empty((n, 4), dtype=torch.float32, device=dev)
        ws = torch.empty(max(int(lib.nerf_amd_render_image_workspace_bytes(code, n, N)), 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_render_image_forward(
                h_pose.ctypes.data, H, W, f, int(ray0), n, _lib.ptr(jit), _lib.ptr(_tbins(tn, tf, N, dev)),
                _lib.ptr(packed[0]), code, flags, int(seed), _lib.ptr(pixels), _lib.ptr(ws), int(N),
                _lib.stream_ptr(dev)), "nerf_amd_render_image_forward")
        return pixels

    return guarded_launch([net], code, launch)


def render_view_sharded(net, pose, cam_params, *, group=None, **kw):
    """Multi-GPU render_view: rank r renders its contiguous pixel range and ONE
    all-gather assembles pixels [H*W,4] on every rank (BASELINE config 4)."""
    from .. import parallel
    rank, world = parallel.world_info(group)
    n = int(cam_params[0]) * int(cam_params[1])
    lo, hi = parallel.shard_range(n, rank, world)
    u = kw.pop("u", None)
    shard = render_view(net, pose, cam_params, ray0=lo, n_rays=hi - lo,
                        u=None if u is None else u[lo:hi], **kw)
    return parallel.gather_pixels(shard, n, group)


def sample_pdf(ts, w, Nf, *, u=None, device_rng=False, seed=0, ray_id0=0):
    """Importance sampling for a fine pass (BASELINE config 4; NOT in the
    reference -- parity unpinned, follows the NeRF paper): ts [B,Nc] coarse
    positions, w [B,Nc] coarse weights -> sorted [B,Nc+Nf] positions (the coarse
    ones plus Nf inverse-CDF samples).  u [B,Nf] explicit uniforms; default one
    CPU torch.rand(B,Nf); device_rng=True: counter RNG."""
    _lib.require_cuda_f32(ts, "ts")
    _lib.require_cuda_f32(w, "w")
    B, Nc = ts.shape
    dev = ts.device
    flags, jit = 0, None
    if u is not None:
        jit = _lib.require_cuda_f32(u, "u").contiguous()
    elif device_rng:
        flags = _lib.FLAG_DEVICE_RNG
    else:
        jit, pending_rng = reference_rand(B, Nf, dev)
        pending_rng.finish()
    out = torch.empty((B, Nc + Nf), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nerf_amd_sample_pdf(
            _lib.ptr(ts.contiguous()), _lib.ptr(w.contiguous()), _lib.ptr(jit), flags, int(seed), int(ray_id0),
            _lib.ptr(out), B, int(Nc), int(Nf), _lib.stream_ptr(dev)), "nerf_amd_sample_pdf")
    return out


def render_hierarchical(rays, net_coarse, net_fine, Nc=64, Nf=128, tn=2, tf=6, *, u_c=None, u_f=None,
                        precision=None, device_rng=False, seed=0, ray_id0=0):
    """Coarse + fine render (BASELINE config 4: 64 + 128 samples).  The reference
    only has the single-pass render_nerf (its CoarseNet / FineNet are empty
    classes), so this composition is new: a coarse render_nerf pass with Nc
    stratified samples, sample_pdf on its weights, and a second render_nerf pass
    of ``net_fine`` on the merged Nc+Nf positions (explicit ts).  Each pass is the
    pinned render_nerf; only the sampler in between is unpinned.
    Returns (fine 5-tuple, coarse 5-tuple, ts_fine).  Every arithmetic step runs in the library
    (sample positions come out of the kernel); see render_hierarchical_view for the one-call form."""
    _lib.require_cuda_f32(rays, "rays")
    dev, B = rays.device, rays.size(0)
    rays = rays.detach().contiguous()
    code = _lib.precision_code(net_coarse.precision if precision is None else precision)
    net_coarse.packed_weights(code)
    flags, jit = _lib.FLAG_DEVICE_RNG, None
    if u_c is None and not device_rng:
        u_c, pending_rng = reference_rand(B, Nc, dev)
        pending_rng.finish()
    if u_c is not None:
        flags, jit = 0, _lib.require_cuda_f32(u_c, "u_c").contiguous()
    # the coarse pass through the stage-1 entry point: the sampler needs the positions the kernel drew
    lib = _lib.lib()

    def launch(code, packed):
        raw = torch.empty((B, Nc, 4), dtype=torch.float32, device=dev)
        ts_c = torch.empty((B, Nc), dtype=torch.float32, device=dev)
        outs = [torch.empty(s_, dtype=torch.float32, device=dev) for s_ in ((B, 3), (B,), (B, Nc), (B,), (B, Nc))]
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(lib.nerf_amd_mlp_forward_rays(
                _lib.ptr(rays), _lib.ptr(jit), _lib.ptr(_tbins(tn, tf, Nc, dev)), _lib.ptr(packed[0]), code, flags, int(seed),
                int(ray_id0), _lib.ptr(raw), _lib.ptr(ts_c), B, Nc, st), "nerf_amd_mlp_forward_rays")
            _lib.check(lib.nerf_amd_volume_render_rays(_lib.ptr(raw), _lib.ptr(ts_c), _lib.ptr(rays),
                                                       *[_lib.ptr(x) for x in outs], B, Nc, st), "nerf_amd_volume_render_rays")
        return ts_c, tuple(outs)

    ts_c, coarse = guarded_launch([net_coarse], code, launch)
    ts_f = sample_pdf(ts_c, coarse[4], Nf, u=u_f, device_rng=device_rng, seed=seed, ray_id0=ray_id0)
    fine = render_nerf(rays, net_fine, Nc + Nf, tn, tf, ts=ts_f, precision=precision)
    return fine, coarse, ts_f


def render_hierarchical_view(net_coarse, net_fine, pose, cam_params, Nc=64, Nf=128, *, tn=2, tf=6, u_c=None, u_f=None,
                             ray0=0, n_rays=None, precision=None, device_rng=False, seed=0):
    """BASELINE config 4 for one view (or its pixel range [ray0, ray0+n_rays)) in ONE library call:
    device ray generation -> coarse pass (Nc stratified samples) -> sample_pdf -> fine pass on the
    Nc+Nf merged positions -> clip(rgb,0,1)  (nerf_amd_render_hierarchical_forward: four launches,
    no torch arithmetic on the path).  Returns pixels [n,4] = [r,g,b,disparity] on the GPU.
    u_c [n,Nc] / u_f [n,Nf]: explicit uniforms for these pixels; default: the reference-style CPU
    draws torch.rand(n,Nc) then torch.rand(n,Nf); device_rng=True: counter RNG keyed by global pixel id.
    Parity unpinned (no reference counterpart)."""
    import numpy as np
    dev = next(net_coarse.parameters()).device
    H, W, f = int(cam_params[0]), int(cam_params[1]), float(cam_params[2])
    n = H * W - ray0 if n_rays is None else int(n_rays)
    code = _lib.precision_code(net_coarse.precision if precision is None else precision)
    flags = 0
    if device_rng:
        flags, u_c, u_f = _lib.FLAG_DEVICE_RNG, None, None
    else:
        if u_c is None:
            u_c, pend = reference_rand(n, Nc, dev)
            pend.finish()
        if u_f is None:
            u_f, pend = reference_rand(n, Nf, dev)
            pend.finish()
        u_c = _lib.require_cuda_f32(u_c, "u_c").contiguous()
        u_f = _lib.require_cuda_f32(u_f, "u_f").contiguous()
    lib = _lib.lib()
    h_pose = np.zeros((3, 4), dtype=np.float32)
    h_pose[:] = np.asarray(pose, dtype=np.float32)[:3, :4]

    def launch(code, packed):
        pixels = torch.empty((n, 4), dtype=torch.float32, device=dev)
        ws = torch.empty(max(int(lib.nerf_amd_render_hierarchical_workspace_bytes(n, Nc, Nf)), 256), dtype=torch.uint8,
                         device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_render_hierarchical_forward(
                h_pose.ctypes.data, H, W, f, int(ray0), n, _lib.ptr(u_c), _lib.ptr(u_f), _lib.ptr(_tbins(tn, tf, Nc, dev)),
                _lib.ptr(packed[0]), _lib.ptr(packed[-1]), code, flags, int(seed), _lib.ptr(pixels), _lib.ptr(ws), int(Nc),
                int(Nf), _lib.stream_ptr(dev)), "nerf_amd_render_hierarchical_forward")
        return pixels

    # one precision for both passes: if either network left the fp16 range, both render with bf16 operands
    return guarded_launch([net_coarse] if net_fine is net_coarse else [net_coarse, net_fine], code, launch)


def render_hierarchical_sharded(net_coarse, net_fine, pose, cam_params, Nc=64, Nf=128, *, group=None, **kw):
    """Multi-GPU config 4: rank r renders its contiguous pixel range coarse+fine in one library call and
    ONE all-gather assembles pixels [H*W,4] on every rank.  Jitter is keyed by global pixel id
    (device RNG) or sliced from the caller's u_c / u_f, so the image does not depend on the world size."""
    from .. import parallel
    rank, world = parallel.world_info(group)
    n = int(cam_params[0]) * int(cam_params[1])
    lo, hi = parallel.shard_range(n, rank, world)
    u_c, u_f = kw.pop("u_c", None), kw.pop("u_f", None)
    shard = render_hierarchical_view(net_coarse, net_fine, pose, cam_params, Nc, Nf, ray0=lo, n_rays=hi - lo,
                                     u_c=None if u_c is None else u_c[lo:hi], u_f=None if u_f is None else u_f[lo:hi],
                                     **kw)
    return parallel.gather_pixels(shard, n, group)
