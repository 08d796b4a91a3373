"""Training-side entry points (reference train.py:16-26,41-57): autograd
wrappers around the HIP forward/backward kernels, loss/PSNR helpers.

The backward kernels are not built yet; until they are, asking for gradients
through the fused path raises instead of silently returning tensors without a
grad_fn.
"""
import torch


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


def _not_built(what):
    raise NotImplementedError(
        f"{what}: the HIP backward kernels are not built yet; wrap inference calls in "
        "torch.no_grad() (as the reference's render_image does, utils/rendering.py:99)")


def nerf_forward_autograd(net, v, precision):
    _not_built("Nerf.forward with gradients")


def volume_render_autograd(nerf_outs, ts, dirs):
    _not_built("volume_render with gradients")


def render_nerf_autograd(rays, net, N, tn, tf, jit, flags, precision, seed, ray_id0):
    _not_built("render_nerf with gradients")
