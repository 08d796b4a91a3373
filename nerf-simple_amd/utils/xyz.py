"""Drop-in for the reference's utils/xyz.py.

``gamma`` / ``positional_encoder`` (reference utils/xyz.py:6-36) run as HIP
kernels on the tensor's device.  The camera helpers (utils/xyz.py:38-91) are
host-side input generators, as in the reference.

Differences from the reference, on purpose:
  * tensors must already be on the GPU (no CPU path);
  * the "input not in range -1,1" UserWarning (utils/xyz.py:8-9) is not
    raised: it is not part of the results, fires on every lego-scale point,
    and costs two device->host syncs per call.
"""
import numpy as np
import torch

from .. import _lib


def gamma(x, L=4):
    """x [P,C] -> [P, 2*L*C]: cat over levels i of [sin(2^i x), cos(2^i x)] along
    dim 1 (reference utils/xyz.py:6-14; C = 1 at every reference call site)."""
    assert torch.is_tensor(x), "input needs to be a torch tensor"
    _lib.require_cuda_f32(x, "x")
    if x.dim() != 2:
        raise RuntimeError("gamma expects a [P, C] tensor")
    P, C = x.shape
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
