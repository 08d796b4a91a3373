"""Ray selection of the training loop on the device (reference utils/dataload.py:114-153, train.py:47-49).

The reference keeps the ray table ``rays_dataset[mode]`` [n,6] and the pixel colours ``train_imgs`` [n,3] in HOST memory;
every iteration shuffles all n indices on one core (``torch.randperm(n)[:N]``: 0.2 s at n = 5 M, 0.66 s at the 16 M rays
of 25 images of 800 x 800), gathers the batch and copies it over PCIe -- in front of a training step that takes 1.2 ms
here.  ``RayGenerator`` below keeps both tables in HBM and selects on the GPU (csrc/select.hip): the reference's own
``ray_ids`` for the state of torch's CPU generator, which is left where ``torch.randperm(n)`` would leave it, or -- with
``device_rng=True`` -- a batch from the counter RNG with nothing going over PCIe at all.

What is NOT here: reading the dataset from disk (``load_data``: cv2 / natsort / json, out of scope -- SURVEY.md section 2).
``RayGenerator.from_samples`` takes what the reference's ``load_data`` returns, ``from_tables`` takes the two tables.
``select_imgs`` (dataload.py:155-183; only ever used in a commented-out line, train.py:48) draws from numpy's global
generator with another shuffle (``np.random.choice(replace=False)``): not reproduced.
"""
import numpy as np
import torch

from .. import _lib
from . import host_rng

SELECT_KEY = 0x73656c6563743a31             # csrc/select.hip SEL_KEY: the selection's Philox stream, apart from the jitter's
MAX_TABLE = 0xffffffff // 20                # torch.randperm switches algorithm at this length (randperm_cpu)


class RayGenerator:
    """``RayGenerator`` of the reference (utils/dataload.py:131-153) with its tables resident on the GPU.

    ``rays_dataset[mode]`` [n,6] fp32 = [origin, direction] per pixel, images concatenated in order, pixels row-major
    (dataload.py:114-129); ``colours[mode]`` [n,3] fp32 = ``train_imgs`` of train.py:33 after its ``.float()``."""

    def __init__(self, rays_dataset, colours=None, cam_params=None, samples=None):
        self.rays_dataset = {k: _lib.require_cuda_f32(v, f"rays_dataset[{k!r}]").contiguous() for k, v in rays_dataset.items()}
        self.colours = {k: _lib.require_cuda_f32(v, f"colours[{k!r}]").contiguous() for k, v in (colours or {}).items()}
        for k, v in self.rays_dataset.items():
            if v.dim() != 2 or v.shape[1] != 6:
                raise RuntimeError(f"rays_dataset[{k!r}] must be [n,6], got {tuple(v.shape)}")
            if k in self.colours and tuple(self.colours[k].shape) != (v.shape[0], 3):
                raise RuntimeError(f"colours[{k!r}] must be [{v.shape[0]},3], got {tuple(self.colours[k].shape)}")
        self.samples, self.cam_params = samples, cam_params
        if cam_params is not None:
            self.H, self.W, self.f = cam_params[0], cam_params[1], cam_params[2]
        self._ws = {}

    # ---- construction ---------------------------------------------------------------
    @classmethod
    def from_tables(cls, rays, colours=None, device=None, mode="train"):
        """One table (or a dict of tables by mode) of rays [n,6] and colours [n,3]; host tensors are moved once."""
        def up(t):
            t = torch.as_tensor(t)
            return t.float().to(device if device is not None else (t.device if t.is_cuda else "cuda")).contiguous()
        rays = rays if isinstance(rays, dict) else {mode: rays}
        colours = {} if colours is None else (colours if isinstance(colours, dict) else {mode: colours})
        return cls({k: up(v) for k, v in rays.items()}, {k: up(v) for k, v in colours.items()})

    @classmethod
    def from_samples(cls, samples, cam_params, device="cuda"):
        """From the reference's ``load_data`` output: samples[mode] = list of {'img': HxWx3 array in [0,1], 'transform':
        4x4 pose}, cam_params = [H, W, f].  Builds ``rays_dataset`` as the reference does (dataload.py:114-129), each
        image's rays on the device (nerf_amd_generate_rays), and the colour tables as train.py:33 + ``.float()``."""
        from .rendering import generate_rays
        rays, colours = {}, {}
        for mode, items in samples.items():
            if not items:
                continue
            rays[mode] = torch.cat([generate_rays(torch.as_tensor(s["transform"]).float().cpu(), cam_params, device)
                                    for s in items])
            colours[mode] = torch.cat([torch.as_tensor(np.asarray(s["img"])).reshape(-1, 3).float().to(device)
                                       for s in items]).contiguous()
        return cls(rays, colours, cam_params, samples)

    # ---- the reference's call -------------------------------------------------------
    def select(self, mode="train", N=4096):
        """``rays, ray_ids = rg.select(mode, N)`` (dataload.py:141-153): ``ray_ids = torch.randperm(n)[:N]`` from torch's
        CPU default generator -- the same ids, the generator left in the same state -- and ``rays = data[ray_ids]``, both
        on the GPU (``train_imgs[ray_ids]`` of train.py:49 then is an index with a device tensor, or use ``select_batch``)."""
        rays, _, ids = self.select_batch(mode, N, colours=False)
        return rays, ids

    def select_batch(self, mode="train", N=4096, *, colours=True, device_rng=False, seed=0, out=None):
        """(rays [B,6], gt [B,3] or None, ray_ids [B] int64), B = min(N, n): the two gathers of train.py:47-49 in one call.
        ``device_rng=True``: the permutation prefix comes from the counter RNG keyed by ``seed`` instead of torch's CPU
        generator (which is then not touched).  ``out`` = (rays, gt, ray_ids) preallocated buffers."""
        table = self.rays_dataset[mode]
        cols = self.colours.get(mode) if colours else None
        if colours and cols is None:
            raise RuntimeError(f"no colour table for mode {mode!r}")
        n, dev = int(table.shape[0]), table.device
        B = min(int(N), n)
        if n >= MAX_TABLE:
            raise RuntimeError(f"a table of {n} rays: torch.randperm uses another algorithm from {MAX_TABLE} on (NERF_AMD_EUNSUP)")
        if out is None:
            out = (torch.empty((B, 6), dtype=torch.float32, device=dev),
                   torch.empty((B, 3), dtype=torch.float32, device=dev) if cols is not None else None,
                   torch.empty((B,), dtype=torch.int64, device=dev))
        rays, gt, ids = out
        for name, t_, shape, dt in (("rays", rays, (B, 6), torch.float32), ("gt", gt, (B, 3), torch.float32), ("ray_ids", ids, (B,), torch.int64)):
            if t_ is not None and (tuple(t_.shape) != shape or t_.dtype != dt or t_.device != dev or not t_.is_contiguous()):
                raise RuntimeError(f"select_batch: out {name} must be a contiguous {dt} tensor of shape {shape} on {dev}")
        if device_rng:
            self.launch(mode, B, None, int(seed), None, rays, gt, ids)
            return rays, gt, ids
        if host_rng.host_fallback():
            # the reference's own statements (NERF_AMD_HOST_RNG=1, or an unknown generator layout): same numbers by definition
            host_ids = torch.randperm(n)[:B].to(dev)
            ids.copy_(host_ids)
            rays.copy_(table[host_ids])
            if gt is not None:
                gt.copy_(cols[host_ids])
            return rays, gt, ids
        session = host_rng.GeneratorSession(dev)
        try:
            self.select_from_session(session, mode, B, rays, gt, ids)
        finally:
            session.finish()
        return rays, gt, ids

    # ---- pieces for callers that own the stream of draws (training.GraphedTrainStep) ---
    def select_from_session(self, session, mode, B, rays, gt, ids, workspace=None, jitter=None):
        """The reference-stream selection inside an open ``host_rng.GeneratorSession`` (the jitter draw follows in the
        same session, as render_nerf's torch.rand follows rg.select in train.py:47-51).  ``jitter`` = (B, N, out): that
        draw is made here too -- the generator's jump over the shuffle and the jump to the jitter's segments are then one
        launch (``GeneratorSession.randperm_then_rand``) -- and returned."""
        n = int(self.rays_dataset[mode].shape[0])
        u = None
        if jitter is not None:
            draws, u = session.randperm_then_rand(n, B, jitter[0], jitter[1], out=jitter[2])
        else:
            draws = session.randperm_draws(n, B)
        self.launch(mode, B, draws, 0, None, rays, gt, ids, workspace=workspace)
        return u

    def workspace(self, B, device):
        key = (int(B), str(device))
        if key not in self._ws:
            nb = max(int(_lib.lib().nerf_amd_select_workspace_bytes(int(B))), 256)
            self._ws[key] = torch.empty(nb, dtype=torch.uint8, device=device)
        return self._ws[key]

    def launch(self, mode, B, draws, seed, seed_mem, rays, gt, ids, stream=None, workspace=None):
        """nerf_amd_select_rays on the current stream (or ``stream``, a ctypes stream pointer: graph capture);
        ``workspace``: the caller's own (a captured launch keeps its address), default one cached per batch size."""
        table = self.rays_dataset[mode]
        cols = self.colours.get(mode) if gt is not None else None
        dev = table.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nerf_amd_select_rays(
                _lib.ptr(draws), int(seed) & 0xffffffffffffffff, seed_mem, int(table.shape[0]), int(B), _lib.ptr(table),
                _lib.ptr(cols), _lib.ptr(rays), _lib.ptr(gt), _lib.ptr(ids), _lib.ptr(workspace if workspace is not None else self.workspace(B, dev)),
                stream if stream is not None else _lib.stream_ptr(dev)), "nerf_amd_select_rays")
