"""MI355X-native NeRF volume-rendering path (drop-in for the hot path of
UCSD-Comp-Imaging/Nerf-Simple: utils/rendering.render_nerf, utils/nets.Nerf,
utils/xyz.positional_encoder).

The compute lives in hand-written HIP kernels for gfx950 behind a C-ABI
shared library (include/nerf_amd.h, csrc/); this package is the thin
Python/PyTorch host side that mirrors the reference's call signatures.
Importing the package does not load the library; the first kernel call does,
and raises if it is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
