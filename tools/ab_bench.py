#!/usr/bin/env python3
"""A/B builds and precisions of the fused MLP kernel in ONE process on ONE device: interleaved rounds,
HIP-event timing of the kernel alone on the 800x800x128 workload (cdna_hip_programming.md rule 24).

    python tools/ab_bench.py [--rounds 8] VARIANT [VARIANT ...]

VARIANT = [path/to/lib.so:]precision[:fused][:u], e.g.  bf16  fp16:fused  /tmp/old/libnerf_amd.so:bf16
(":u" = jitter read from a [B,N] buffer in HBM, as in the reference-compatible default of render_nerf,
instead of the counter RNG)
Without a path the in-tree library is used.  A second build to compare against is made by
checking the other revision out into a scratch directory and running `make -C nerf-simple_amd/csrc`
there; nothing in the shipped sources is switched by environment variables or macros.
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_simple_amd import _lib                      # noqa: E402
from nerf_simple_amd.utils import synthetic            # noqa: E402
from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--hw", type=int, default=800)
ap.add_argument("--samples", type=int, default=128)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

dev = torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(0, "structured")
flat = synthetic.flatten_state_dict(sd).to(dev)
pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
rays = camera_rays([pose], [args.hw, args.hw, synthetic.focal_from_fov(args.hw)]).float().contiguous().to(dev)
B, N = rays.shape[0], args.samples
raw = torch.empty(B, N, 4, device=dev)
ts = torch.empty(B, N, device=dev)
pixels = torch.empty(B, 4, device=dev)
tb = torch.linspace(2, 6, N + 1).to(dev)
u_buf = torch.rand(B, N, device=dev)
vp, i64, i32, u32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64


class Variant:
    def __init__(self, spec):
        parts = spec.split(":")
        self.given_u = parts[-1] == "u"
        if self.given_u:
            parts = parts[:-1]
        self.fused = parts[-1] == "fused"
        if self.fused:
            parts = parts[:-1]
        self.path = parts[0] if len(parts) == 2 else _lib.LIB_PATH
        self.precision = _lib.precision_code(parts[-1])
        self.name = spec
        h = ctypes.CDLL(os.path.abspath(self.path))
        h.nerf_amd_packed_bytes.restype, h.nerf_amd_packed_bytes.argtypes = i64, [i32]
        h.nerf_amd_pack_weights.restype, h.nerf_amd_pack_weights.argtypes = i32, [vp, vp, i32, vp]
        h.nerf_amd_mlp_forward_rays.restype = i32
        h.nerf_amd_mlp_forward_rays.argtypes = [vp, vp, vp, vp, i32, u32, u64, i64, vp, vp, i64, i32, vp]
        if self.fused:
            h.nerf_amd_render_pixels_forward.restype = i32
            h.nerf_amd_render_pixels_forward.argtypes = [vp, vp, vp, vp, i32, u32, u64, i64, vp, vp, i64, i32, vp]
        self.h = h
        self.packed = torch.empty(h.nerf_amd_packed_bytes(self.precision), dtype=torch.uint8, device=dev)
        _lib.check(h.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(self.packed), self.precision,
                                           _lib.stream_ptr(dev)), "pack")
        self.times = []

    def run(self):
        jit, flags = (_lib.ptr(u_buf), 0) if self.given_u else (None, 2)
        if self.fused:
            _lib.check(self.h.nerf_amd_render_pixels_forward(
                _lib.ptr(rays), jit, _lib.ptr(tb), _lib.ptr(self.packed), self.precision, flags, 1234, 0,
                _lib.ptr(pixels), None, B, N, _lib.stream_ptr(dev)), "render_pixels")
        else:
            _lib.check(self.h.nerf_amd_mlp_forward_rays(
                _lib.ptr(rays), jit, _lib.ptr(tb), _lib.ptr(self.packed), self.precision, flags, 1234, 0,
                _lib.ptr(raw), _lib.ptr(ts), B, N, _lib.stream_ptr(dev)), "mlp")


variants = [Variant(s) for s in args.variants]
outs = []
for v in variants:
    v.run()
    torch.cuda.synchronize()
    outs.append(pixels.clone() if v.fused else raw.clone())
for r in range(args.rounds):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        v.run()
        e1.record()
        torch.cuda.synchronize()
        v.times.append(e0.elapsed_time(e1))
for v in variants:
    t = v.times
    tf = B * N * 1186816 / (statistics.median(t) * 1e-3) / 1e12
    print(f"{v.name}: median {statistics.median(t):.3f} ms  min {min(t):.3f}  max {max(t):.3f}  -> {tf:.0f} TFLOP/s "
          f"({tf / 25:.1f}% of 2.5 PF)")
for i in range(1, len(variants)):
    if outs[i].shape == outs[0].shape:
        print(f"max |out[{variants[i].name}] - out[{variants[0].name}]| =", float((outs[i] - outs[0]).abs().max()),
              " scale", float(outs[0].abs().max()))
