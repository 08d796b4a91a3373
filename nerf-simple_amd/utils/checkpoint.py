"""Checkpoint compatibility.

The only persistent format of the reference is ``torch.save(net.state_dict())``
with the 24 keys of its ``Nerf`` (reference train.py:84-91, loaded with
``strict=True`` at test.py:28).  ``Nerf`` here keeps exactly those keys as fp32
master parameters, so reference checkpoints load unchanged and checkpoints
written here load into the reference; the MFMA-ordered weight images are a
derived cache and are never stored.
"""
import torch

from .synthetic import PARAM_SPECS


def save_checkpoint(net, path):
    """state_dict -> ``path`` (same call as reference train.py:87,91)."""
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.save(sd, path)
    return path


def load_checkpoint(net, path, map_location="cpu"):
    """Load a reference (or own) ``.pth`` into ``net`` with ``strict=True``.
    The file is read with ``weights_only=True``: nothing in it is executed."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    check_state_dict(sd, net)
    net.load_state_dict(sd, strict=True)
    return net


def check_state_dict(sd, net=None):
    """Raise unless ``sd`` has exactly the reference's 24 keys with the shapes of ``net`` (a Nerf of any
    (Lp, Ld, H)); without ``net``: the shapes of the default Nerf()."""
    want = dict(PARAM_SPECS) if net is None else {k: tuple(v.shape) for k, v in net.state_dict().items()}
    if set(sd.keys()) != set(want):
        missing, extra = set(want) - set(sd), set(sd) - set(want)
        raise KeyError(f"not a Nerf checkpoint: missing {sorted(missing)}, unexpected {sorted(extra)}")
    for k, shape in want.items():
        if tuple(sd[k].shape) != shape:
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shape}")
