"""Drop-in for the reference's utils/xyz.py.

``gamma`` / ``positional_encoder`` (reference utils/xyz.py:6-36) run as HIP
kernels on the tensor's device.  The camera helpers (utils/xyz.py:38-91) are
host-side input generators, as in the reference.

Differences from the reference, on purpose:
  * tensors must already be on the GPU (no CPU path);
  * the "input not in range -1,1, check rescaling" UserWarning (utils/xyz.py:8-9) is raised LAZILY: the reference
    pays two device->host syncs per gamma call for it; here a kernel ORs the verdict into a device word
    (nerf_amd_range_check), the word travels to pinned memory asynchronously, and the warning is issued by the next call
    that finds the copy complete -- same text, same category, no host wait.  ``flush_range_warning()`` waits for it.
"""
import warnings

import numpy as np
import torch

from .. import _lib

RANGE_WARNING = "input not in range -1,1, check rescaling"          # the reference's text (utils/xyz.py:9)


class _RangeWatch:
    """Per device: the word nerf_amd_range_check ORs into, its pinned host copy and the event behind that copy."""

    def __init__(self, device):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None

    def poll(self, wait=False, stacklevel=4):
        if self.event is None:
            return
        if wait:
            self.event.synchronize()
        if self.event.query():
            self.event = None
            if int(self.host[0]):
                warnings.warn(RANGE_WARNING, UserWarning, stacklevel=stacklevel)

    def after_launch(self):
        """Behind a check kernel: if no copy is in flight, send the word home and clear it for the calls to come."""
        if self.event is None and not torch.cuda.is_current_stream_capturing():
            self.host.copy_(self.word, non_blocking=True)
            self.word.zero_()
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(self.word.device))


_range_watch = {}


def _watch(device):
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    if key not in _range_watch:
        _range_watch[key] = _RangeWatch(torch.device("cuda", key))
    return _range_watch[key]


def range_check_values(x, stacklevel=4):
    """The reference's check of one gamma() argument / of the query points [P,6] (all columns): enqueue, do not wait."""
    if x.numel() == 0 or torch.cuda.is_current_stream_capturing():
        return
    w = _watch(x.device)
    w.poll(stacklevel=stacklevel)
    x = x.detach().contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().nerf_amd_range_check(None, _lib.ptr(x), None, None, 0, 0, 0, _lib.ptr(w.word), x.numel(), 0,
                                                   _lib.stream_ptr(x.device)), "nerf_amd_range_check")
        w.after_launch()


def range_check_rays(rays, jit, tbins, flags, seed, ray_id0, N, stacklevel=4):
    """The same for the points render_nerf forms from its rays (first and last sample of every ray: a coordinate is
    monotone along its ray): ``jit`` is a tensor, a ctypes pointer (seed-in-memory form) or None."""
    if rays.shape[0] == 0:
        return
    w = _watch(rays.device)
    w.poll(stacklevel=stacklevel)
    jp = _lib.ptr(jit) if (jit is None or torch.is_tensor(jit)) else jit
    with torch.cuda.device(rays.device):
        _lib.check(_lib.lib().nerf_amd_range_check(_lib.ptr(rays), None, jp, _lib.ptr(tbins), int(flags), int(seed), int(ray_id0),
                                                   _lib.ptr(w.word), rays.shape[0], int(N), _lib.stream_ptr(rays.device)),
                   "nerf_amd_range_check")
        w.after_launch()


def flush_range_warning(device=None):
    """Wait for the pending range verdicts (all devices, or one) and issue the warning now if one is due."""
    for key, w in list(_range_watch.items()):
        if device is None or torch.device(device).index in (None, key):
            w.poll(wait=True, stacklevel=3)


def gamma(x, L=4):
    """x [P,C] -> [P, 2*L*C]: cat over levels i of [sin(2^i x), cos(2^i x)] along
    dim 1 (reference utils/xyz.py:6-14; C = 1 at every reference call site)."""
    assert torch.is_tensor(x), "input needs to be a torch tensor"
    _lib.require_cuda_f32(x, "x")
    if x.dim() != 2:
        raise RuntimeError("gamma expects a [P, C] tensor")
    P, C = x.shape
    range_check_values(x, stacklevel=3)                 # utils/xyz.py:8-9, lazily
    lib = _lib.lib()
    cols = []
    with torch.cuda.device(x.device):
        st = _lib.stream_ptr(x.device)
        for c in range(C):
            xc = x[:, c]
            out = torch.empty((P, 2 * L), dtype=torch.float32, device=x.device)
            _lib.check(lib.nerf_amd_gamma(_lib.ptr(xc), xc.stride(0) if P > 1 else 1,
                                          _lib.ptr(out), P, L, st), "nerf_amd_gamma")
            cols.append(out)
    if C == 1:
        return cols[0]
    return torch.stack(cols, dim=2).reshape(P, 2 * L * C)


def positional_encoder(vec, Lp=10, Ld=4):
    """vec [P,6] = x,y,z,d1,d2,d3 -> (posx [P,3+6Lp], posd [P,3+6Ld]), grouped
    per coordinate (reference utils/xyz.py:16-36)."""
    _lib.require_cuda_f32(vec, "vec")
    if vec.dim() != 2 or vec.shape[1] != 6:
        raise RuntimeError("positional_encoder expects a [P, 6] tensor")
    vec = vec.contiguous()
    P = vec.shape[0]
    range_check_values(vec, stacklevel=3)               # gamma's check on all six columns (utils/xyz.py:8-9, :26-31), lazily
    posx = torch.empty((P, 3 + 6 * Lp), dtype=torch.float32, device=vec.device)
    posd = torch.empty((P, 3 + 6 * Ld), dtype=torch.float32, device=vec.device)
    with torch.cuda.device(vec.device):
        _lib.check(_lib.lib().nerf_amd_positional_encoder(
            _lib.ptr(vec), _lib.ptr(posx), _lib.ptr(posd), P, Lp, Ld,
            _lib.stream_ptr(vec.device)), "nerf_amd_positional_encoder")
    return posx, posd


# --------------------------------------------------------------------------
# cameras (host side)
# --------------------------------------------------------------------------
def rays_single_cam(cam_params):
    """[H, W, f] -> camera-frame ray directions [3, H*W], pixel order h*W + w,
    dir(h,w) = ((w - W//2)/f, -(h - H//2)/f, -1)  (reference utils/xyz.py:38-52)."""
    H, W, f = cam_params
    rows = (torch.arange(H) - H // 2).reshape(H, 1).expand(H, W)
    cols = (torch.arange(W) - W // 2).reshape(1, W).expand(H, W)
    d = torch.stack((cols / f, -rows / f, -torch.ones_like(cols))).float()
    return d.reshape(3, -1)


def polar_to_mat(theta):
    """Rotation about x by theta radians, 4x4 (reference utils/xyz.py:55-61)."""
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[1., 0., 0., 0.], [0., c, s, 0.], [0., -s, c, 0.], [0., 0., 0., 1.]])


def phi_to_mat(phi):
    """Rotation about z by phi radians, 4x4 (reference utils/xyz.py:63-68)."""
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[c, s, 0., 0.], [-s, c, 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]])


def spherical_to_pose(r, theta, phi):
    """Camera-to-world pose for spherical (r, theta deg, phi deg):
    Rz(phi) @ Rx(theta) @ T(0,0,r)  (reference utils/xyz.py:70-81)."""
    trans = np.eye(4)
    trans[2, 3] = r
    return phi_to_mat(np.radians(phi)) @ polar_to_mat(np.radians(theta)) @ trans


def poses_to_render(r, theta, n_phi=40):
    """n_phi float32 poses at azimuths linspace(0, 360, n_phi) (reference utils/xyz.py:83-91)."""
    return [torch.from_numpy(spherical_to_pose(r, theta, phi)).float()
            for phi in np.linspace(0, 360.0, n_phi)]


def camera_rays(poses, cam_params):
    """poses: list/stack of [4,4] float tensors -> world rays [n*H*W, 6] =
    [origin, R @ dir], the table the reference builds at utils/rendering.py:129-134
    and utils/dataload.py:114-129."""
    if not torch.is_tensor(poses):
        poses = torch.stack(list(poses))
    if poses.dim() == 2:
        poses = poses.unsqueeze(0)
    H, W = cam_params[0], cam_params[1]
    d = rays_single_cam(cam_params)
    rd = torch.matmul(poses[:, :3, :3], d)
    o = poses[:, :3, 3:].expand(len(poses), 3, H * W)
    return torch.cat((o, rd), dim=1).permute(0, 2, 1).reshape(-1, 6)
