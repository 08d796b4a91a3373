"""Deterministic synthetic inputs for the render path: weights and cameras.

There is no dataset or checkpoint offline, so every configuration of
BASELINE.json is driven by (a) generator-seeded weights with the exact 24-key
state-dict layout of the reference ``Nerf`` (reference utils/nets.py:16-32) and
(b) synthetic pinhole cameras on a sphere (reference utils/xyz.py:38-52,70-81,
focal formula utils/dataload.py:102-104).

This module is pure numpy/torch-CPU host code.  It is shared by bench.py, the
tests and the golden-vector generator so that both sides of a parity check are
fed bit-identical weights without shipping 2.4 MB fixtures.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

# (state-dict key, shape) in the order torch registers them
# (reference utils/nets.py:16-32).
PARAM_SPECS = (
    ("layers_0.0.weight", (256, 63)), ("layers_0.0.bias", (256,)),
    ("layers_0.2.weight", (256, 256)), ("layers_0.2.bias", (256,)),
    ("layers_0.4.weight", (256, 256)), ("layers_0.4.bias", (256,)),
    ("layers_0.6.weight", (256, 256)), ("layers_0.6.bias", (256,)),
    ("layers_0.8.weight", (256, 256)), ("layers_0.8.bias", (256,)),
    ("skip_conn_layer.0.weight", (256, 319)), ("skip_conn_layer.0.bias", (256,)),
    ("layers_1.0.weight", (256, 256)), ("layers_1.0.bias", (256,)),
    ("layers_1.2.weight", (256, 256)), ("layers_1.2.bias", (256,)),
    ("sigma_fc.0.weight", (1, 256)), ("sigma_fc.0.bias", (1,)),
    ("layers_2.weight", (256, 256)), ("layers_2.bias", (256,)),
    ("color_fc.0.weight", (128, 283)), ("color_fc.0.bias", (128,)),
    ("color_fc.2.weight", (3, 128)), ("color_fc.2.bias", (3,)),
)
PARAM_COUNT = sum(int(np.prod(s)) for _, s in PARAM_SPECS)  # 595,844

LEGO_CAMERA_ANGLE_X = 0.6911112070083618  # lego transforms_train.json


def synthetic_state_dict(seed=0, kind="default"):
    """24-key fp32 state dict filled from a seeded numpy Generator.

    kind="default":    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases
                       (the nn.Linear default scale) -> a low-contrast "fog".
    kind="structured": hidden weights at He-uniform scale so activations keep
                       their variance through the 8 ReLU layers, and the sigma /
                       colour heads scaled so sigma spans both signs and rgb
                       spans [0,1]: an image with real contrast, which makes
                       PSNR comparisons meaningful.
    """
    if kind not in ("default", "structured"):
        raise ValueError("kind must be 'default' or 'structured'")
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for key, shape in PARAM_SPECS:
        fan_in = shape[1] if len(shape) == 2 else None
        if fan_in is None:
            # bias: fan_in of the layer it belongs to
            wkey = key[:-4] + "weight"
            fan_in = dict(PARAM_SPECS)[wkey][1]
        bound = 1.0 / math.sqrt(fan_in)
        if kind == "structured" and key.endswith("weight"):
            bound = math.sqrt(6.0 / fan_in)
        arr = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        if kind == "structured":
            # heads: zero-mean rows (their inputs are post-ReLU, i.e. positive
            # on average) and constant biases picked by eye for seed 0, so that
            # sigma straddles 0 (opaque and empty regions) and rgb fills [0,1]
            if key == "sigma_fc.0.weight":
                arr = (arr - arr.mean()) * 8.0
            elif key == "sigma_fc.0.bias":
                arr = arr * 0 + 5.0
            elif key == "color_fc.2.weight":
                arr = (arr - arr.mean(axis=1, keepdims=True)) * 0.5
            elif key == "color_fc.2.bias":
                arr = np.array([0.3, 0.0, 0.9], dtype=np.float32)
        sd[key] = torch.from_numpy(np.ascontiguousarray(arr))
    return sd


def perturbed_state_dict(sd, seed=1, rel=0.05):
    """A 'teacher' differing from ``sd`` by a relative weight perturbation;
    its CPU render is the synthetic PSNR target T of SURVEY.md section 8d."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for k, v in sd.items():
        noise = rng.standard_normal(size=tuple(v.shape)).astype(np.float32)
        out[k] = v * (1.0 + rel * torch.from_numpy(noise))
    return out


def flatten_state_dict(sd):
    """Concatenate the 24 tensors in PARAM_SPECS order -> fp32 [595844]."""
    return torch.cat([sd[k].reshape(-1).to(torch.float32) for k, _ in PARAM_SPECS])


def focal_from_fov(W, camera_angle_x=LEGO_CAMERA_ANGLE_X):
    """f = W / (2 tan(fov/2))  (reference utils/dataload.py:102-104)."""
    return W / (2.0 * np.tan(camera_angle_x / 2.0))


def points_in_scene(n, seed=0):
    """n query points [n,6]: xyz uniform in the lego-scale cube [-4.5,4.5]^3,
    unit directions (the value ranges SURVEY.md section 8d derives)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = rng.uniform(-4.5, 4.5, size=(n, 3)).astype(np.float32)
    d = rng.standard_normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return torch.from_numpy(np.concatenate([xyz, d], axis=1))
