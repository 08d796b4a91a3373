"""Where the inference tolerances of tests/test_gpu_parity.py come from (test infrastructure; uses the oracle).

None of them is "k x what the GPU showed".  Each is a factor on an error that an INDEPENDENT model of the stated numerics
predicts on the same inputs, computed here on the CPU:

  16-bit kernels (fp16 / bf16 MFMA operands, fp32 accumulate, fp32 bias):
      the oracle's network with every layer's two operands -- the weights and the layer's input, encoder features
      included -- rounded to the operand type (round-to-nearest-even, 2^-12 / 2^-9 relative), products and sums in fp32
      (tests/studies/precision_study.py: the emulation that predicted the PSNR deltas of DESIGN.md section 2 to 0.002 dB).
      Its deviation from the fp32 golden IS the error the operand type costs; the kernel may differ from the emulation by
      summation order, by its encoder's ~1 ulp sines and by fp32 round-off, so the GPU's error per output is allowed
      FACTOR_16 = 1.5 x the emulated error of that output, plus the fp32 bound below as a floor.
  fp32 kernel (exact-f32 MFMA: an fp32 fma chain per dot product):
      the same arithmetic as the reference in another summation order.  The reference's own fp32 result sits
      e_ref = |fp32 oracle - float64 oracle| from the truth; the kernel, measured against the same float64 truth, is
      allowed FACTOR_32 = 2 x that per output (two fp32 evaluations of one expression, max norm over ~1e3 ... 1e5 elements)
      + ULP_FLOOR = 4 x 2^-24 of the output's scale (the final rounding of two different summation orders).

TOL[(precision, weight set)] is the largest such bound over the MLP fixture (G2) and the render fixtures (G4, every
output, N = 32 ... 192): the tests that have no emulation of their own inputs (ragged sizes, the benched workload's
scattered rays) use it.
"""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FACTOR_16 = 1.5
FACTOR_32 = 2.0
ULP_FLOOR = 4 * 2.0 ** -24          # fp32 results are themselves rounded: a few units in the last place of a value of the
                                    # output's scale separate two equally valid summation orders of ~200 terms (acc = sum w)
NAMES = ("rgb", "disp", "alpha", "acc", "w")

_study = None


def study():
    global _study
    if _study is None:
        spec = importlib.util.spec_from_file_location("precision_study", os.path.join(ROOT, "tests", "studies", "precision_study.py"))
        _study = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_study)
    return _study


def scaled_err(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    scale = max(1.0, float(np.nanmax(np.abs(want))))
    return float(np.nanmax(np.abs(got - want))) / scale


def _sd(kind):
    from nerf_simple_amd.utils import synthetic
    return synthetic.synthetic_state_dict(0, kind)


def emulated_forward(sd, v, precision):
    """Nerf.forward with the 16-bit kernels' numerics: [P,6] -> [P,4]."""
    S = study()
    with torch.no_grad():
        return S.forward(sd, v, {k: precision for k in S.LAYERS})


def emulated_render(sd, rays, u, precision):
    """render_nerf with the 16-bit kernels' numerics (fp32 sampling and compositing, as in the kernels): the 5-tuple."""
    import nerf_oracle as O
    with torch.no_grad():
        ts = O.sample_ts(u)
        q, dn = O.query_points(rays, ts)
        out = emulated_forward(sd, q, precision).reshape(rays.shape[0], u.shape[1], 4)
        return O.volume_render(out, ts, dn)


def f64_forward(sd, v):
    import nerf_oracle as O
    with torch.no_grad():
        return O.nerf_forward({k: p.double() for k, p in sd.items()}, v.double())


def f64_render(sd, rays, u):
    """The float64 value of what render_nerf computes from the fp32 inputs (sample positions as the fp32 path forms them:
    they are inputs of the network, not part of its round-off)."""
    import nerf_oracle as O
    with torch.no_grad():
        ts = O.sample_ts(u)
        q, dn = O.query_points(rays, ts)
        out = O.nerf_forward({k: p.double() for k, p in sd.items()}, q.double()).reshape(rays.shape[0], u.shape[1], 4)
        return O.volume_render(out, ts.double(), dn.double())


_cache = {}


def mlp_model(kind, precision):
    """{'rgb': bound, 'sigma': bound} for the G2 fixture, and the truth to compare with ('golden' or the float64 output)."""
    key = ("mlp", kind, precision)
    if key not in _cache:
        g = np.load(os.path.join(GOLDEN, f"mlp_{kind}.npz"))
        v, want = torch.from_numpy(g["v"]), g["out"]
        if precision == "fp32":
            truth = f64_forward(_sd(kind), v).numpy()
            e = {"rgb": scaled_err(want[:, :3], truth[:, :3]), "sigma": scaled_err(want[:, 3], truth[:, 3])}
            _cache[key] = ({k: FACTOR_32 * x + ULP_FLOOR for k, x in e.items()}, truth)
        else:
            emu = emulated_forward(_sd(kind), v, precision).numpy()
            e = {"rgb": scaled_err(emu[:, :3], want[:, :3]), "sigma": scaled_err(emu[:, 3], want[:, 3])}
            floor = max(mlp_model(kind, "fp32")[0].values())
            _cache[key] = ({k: FACTOR_16 * x + floor for k, x in e.items()}, want)
    return _cache[key]


def render_model(kind, precision, N):
    """{output name: bound} for the G4 fixture at N samples, and the truth per output (golden arrays, or float64 for fp32)."""
    key = ("render", kind, precision, N)
    if key not in _cache:
        g = np.load(os.path.join(GOLDEN, f"render_{kind}.npz"))
        rays, u = torch.from_numpy(g["rays"]), torch.from_numpy(g[f"N{N}_u"])
        want = {n: g[f"N{N}_{n}"] for n in NAMES}
        if precision == "fp32":
            truth = {n: o.numpy() for n, o in zip(NAMES, f64_render(_sd(kind), rays, u))}
            _cache[key] = ({n: FACTOR_32 * scaled_err(want[n], truth[n]) + ULP_FLOOR for n in NAMES}, truth)
        else:
            emu = {n: o.numpy() for n, o in zip(NAMES, emulated_render(_sd(kind), rays, u, precision))}
            floor = render_model(kind, "fp32", N)[0]
            _cache[key] = ({n: FACTOR_16 * scaled_err(emu[n], want[n]) + floor[n] for n in NAMES}, want)
    return _cache[key]


class DerivedTol(dict):
    """TOL[(precision, weight set)]: the largest modelled bound over the MLP fixture and the render fixtures."""

    def __missing__(self, key):
        precision, kind = key
        worst = max(mlp_model(kind, precision)[0].values())
        for N in (32, 64, 128, 192):
            worst = max(worst, max(render_model(kind, precision, N)[0].values()))
        self[key] = worst
        return worst


TOL = DerivedTol()


def vs_fp32_result(key):
    """Bound for a comparison with the reference's fp32 RESULT (a golden, the fp32 oracle) where no float64 value of the
    inputs at hand is formed: |gpu - ref32| <= |gpu - truth| + |ref32 - truth| <= (FACTOR_32 + 1) e_ref for the fp32
    kernel; the 16-bit bounds are already stated against the fp32 result."""
    return TOL[key] * (FACTOR_32 + 1.0) / FACTOR_32 if key[0] == "fp32" else TOL[key]


def image_model(kind, precision):
    """For the G5 image (100 x 100, N = 32): (PSNR the modelled numerics reach against the CPU image [dB], the image the
    GPU is to be compared with for that PSNR).  16-bit: PSNR(emulation, fp32 golden); fp32: PSNR(fp32 golden, float64 truth)
    with the float64 truth as the comparison image."""
    key = ("image", kind, precision)
    if key not in _cache:
        import nerf_oracle as O
        from nerf_simple_amd.utils import synthetic
        g = np.load(os.path.join(GOLDEN, f"image_{kind}.npz"))
        u = torch.from_numpy(np.load(os.path.join(GOLDEN, "image_u.npz"))["u"])
        rays = O.camera_rays(torch.from_numpy(g["pose"]), [100, 100, synthetic.focal_from_fov(100)])
        cpu = torch.from_numpy(g["rgb"])
        parts = []
        for s in range(0, rays.shape[0], 2500):
            if precision == "fp32":
                parts.append(f64_render(_sd(kind), rays[s:s + 2500], u[s:s + 2500])[0].clamp(0, 1))
            else:
                parts.append(emulated_render(_sd(kind), rays[s:s + 2500], u[s:s + 2500], precision)[0].clamp(0, 1))
        img = torch.cat(parts)
        if precision == "fp32":
            _cache[key] = (float(O.img_psnr(img, cpu.double())), img)
        else:
            _cache[key] = (float(O.img_psnr(cpu, img.float())), cpu)
    return _cache[key]
