#!/usr/bin/env python3
"""Diagnostic build: where a tile's cycles go inside the fused render kernel (in-kernel s_memtime stamps).

    python tools/stamp_tiles.py --build        # here (no GPU): copy csrc/ to .scratch/stamp, insert stamps, make
    python tools/stamp_tiles.py --run          # on the MI355X: 40 launches of 800x800x128 fp16, print the table

The shipped kernels carry no stamp.  --build copies nerf-simple_amd/csrc to .scratch/stamp/csrc, inserts
`s_memtime` reads at the tile's phase boundaries (prologue, each of the 11 layers, ring barrier, compositing
block and inside it) and a debug symbol the values are copied out of, and builds a second libnerf_amd.so there;
every insertion point is matched literally and must occur exactly once, so the tool fails loudly when the kernel
source has moved on.  Stamps are kept in scalar registers and written out after the tile (waves 0 and 5 of every
workgroup, tiles 600-607 of the workgroup's range), into a buffer nothing else reads.  The table in DESIGN.md
section 5 comes from this tool.  (cdna_hip_programming.md section 7, in-kernel stamps.)
"""
import argparse
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, ".scratch", "stamp")
ST_N, ST_TILES = 20, 8

PATCHES = [
    ("namespace {\n", """namespace {
constexpr int ST_N = %d, ST_TILES = %d, ST_T0 = 600;
__device__ unsigned long long g_stamp[256 * 2 * ST_TILES * ST_N];
#define STAMP(i) do { sv[i] = __builtin_amdgcn_s_memtime(); } while (0)
""" % (ST_N, ST_TILES)),
    ("""__device__ __forceinline__ void run_tile(const Ctx& c, const MlpArgs& a, long long tile_base, State& st) {
    stage_inputs<RAYS, SAVE, COMP>(c, a, tile_base, st);""",
     """__device__ __forceinline__ void run_tile(const Ctx& c, const MlpArgs& a, long long tile_base, State& st, unsigned long long (&sv)[ST_N]) {
    STAMP(0);
    stage_inputs<RAYS, SAVE, COMP>(c, a, tile_base, st);
    STAMP(1);"""),
    ("run_tile<RAYS, SAVE, false>(c, a, tile_base, st);",
     "unsigned long long sv[ST_N]; run_tile<RAYS, SAVE, false>(c, a, tile_base, st, sv);"),
    ("run_tile<true, false, true>(c, a, tile_base, st);",
     "unsigned long long sv[ST_N]; run_tile<true, false, true>(c, a, tile_base, st, sv);"),
    ("            const int done_q = q_tile + TILE_PTS < n_pts ? q_tile + TILE_PTS : n_pts;",
     "            STAMP(13);\n            const int done_q = q_tile + TILE_PTS < n_pts ? q_tile + TILE_PTS : n_pts;"),
    ("                    const float dnorm = nerf_composite::unit_dir_norm(d[0], d[1], d[2], true);",
     """                    STAMP(16);
                    const float dnorm = nerf_composite::unit_dir_norm(d[0], d[1], d[2], true);
                    asm volatile("" :: "v"(dnorm));
                    STAMP(17);"""),
    ("                    nerf_composite::composite_ray(src, a.N, c.lane, dnorm, gray, out);\n",
     "                    nerf_composite::composite_ray(src, a.N, c.lane, dnorm, gray, out);\n                    STAMP(18);\n"),
    ("""                __builtin_amdgcn_s_barrier();                         // the next tile's prologue rewrites ring slots read above
                asm volatile("" ::: "memory");
            }
""", """                __builtin_amdgcn_s_barrier();                         // the next tile's prologue rewrites ring slots read above
                asm volatile("" ::: "memory");
            }
            STAMP(14);
            {
                const int ti = q_tile / TILE_PTS - ST_T0;
                if (ti >= 0 && ti < ST_TILES && (c.wave == 0 || c.wave == 5) && c.lane == 0) {
                    sv[15] = __builtin_amdgcn_s_memrealtime();
                    for (int i = 0; i < ST_N; ++i)
                        g_stamp[((blockIdx.x * 2 + (c.wave ? 1 : 0)) * ST_TILES + ti) * ST_N + i] = sv[i];
                }
            }
"""),
    ('extern "C" int NERF_LAUNCH(', '''#ifdef NERF_HALF
extern "C" int nerf_amd_debug_read_stamps(void* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(g_stamp));
}
#endif
extern "C" int NERF_LAUNCH('''),
]


def stamped_source():
    """mlp_bf16_16.hip with the stamps inserted; exits when an insertion point is not found exactly once."""
    src = open(os.path.join(ROOT, "nerf-simple_amd", "csrc", "mlp_bf16_16.hip")).read()
    for old, new in PATCHES:
        if src.count(old) != 1:
            sys.exit(f"stamp_tiles: insertion point not found exactly once:\n{old}")
        src = src.replace(old, new)
    for L in range(11):
        old = [ln for ln in src.split("\n") if ln.startswith(f"    run_layer<{L}, SAVE>(")]
        if len(old) != 1:
            sys.exit(f"stamp_tiles: run_layer<{L}> call not found exactly once")
        src = src.replace(old[0] + "\n", old[0] + f"\n    STAMP({L + 2});\n")
    return src


def build():
    src = stamped_source()
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    shutil.copytree(os.path.join(ROOT, "nerf-simple_amd", "csrc"), os.path.join(SCRATCH, "csrc"),
                    ignore=shutil.ignore_patterns("build", "*.so", "*.o"))
    os.makedirs(os.path.join(ROOT, ".scratch", "include"), exist_ok=True)     # the Makefile's ../../include
    shutil.copy(os.path.join(ROOT, "include", "nerf_amd.h"), os.path.join(ROOT, ".scratch", "include"))
    open(os.path.join(SCRATCH, "csrc", "mlp_bf16_16.hip"), "w").write(src)
    subprocess.check_call(["make", "-C", os.path.join(SCRATCH, "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    print("built", os.path.join(SCRATCH, "libnerf_amd.so"))


def run():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    dev = torch.device("cuda:0")
    h = ctypes.CDLL(os.path.join(SCRATCH, "libnerf_amd.so"))
    vp, i64, i32, u32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
    h.nerf_amd_packed_bytes.restype, h.nerf_amd_packed_bytes.argtypes = i64, [i32]
    h.nerf_amd_pack_weights.restype, h.nerf_amd_pack_weights.argtypes = i32, [vp, vp, i32, vp]
    h.nerf_amd_render_pixels_forward.restype = i32
    h.nerf_amd_render_pixels_forward.argtypes = [vp, vp, vp, vp, i32, u32, u64, i64, vp, vp, i64, i32, vp]
    h.nerf_amd_debug_read_stamps.restype, h.nerf_amd_debug_read_stamps.argtypes = i32, [vp]
    prec = _lib.precision_code("fp16")
    flat = synthetic.flatten_state_dict(synthetic.synthetic_state_dict(0, "structured")).to(dev)
    packed = torch.empty(h.nerf_amd_packed_bytes(prec), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    _lib.check(h.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(packed), prec, st), "pack")
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [800, 800, synthetic.focal_from_fov(800)]).float().contiguous().to(dev)
    B, N = rays.shape[0], 128
    tb = torch.linspace(2, 6, N + 1).to(dev)
    pixels = torch.empty(B, 4, device=dev)
    for _ in range(40):                 # back-to-back launches on random data: the clock has settled by the last
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(h.nerf_amd_render_pixels_forward(_lib.ptr(rays), None, _lib.ptr(tb), _lib.ptr(packed), prec, 2,
                                                    1234, 0, _lib.ptr(pixels), None, B, N, st), "render")
        e1.record()
        torch.cuda.synchronize()
    print(f"stamped build, last launch: {e0.elapsed_time(e1):.2f} ms")
    buf = np.zeros(256 * 2 * ST_TILES * ST_N, dtype=np.uint64)
    _lib.check(h.nerf_amd_debug_read_stamps(buf.ctypes.data_as(vp)), "stamps")
    s = buf.reshape(256, 2, ST_TILES, ST_N).astype(np.int64)
    names = ["prologue"] + [f"layer {L}" for L in range(11)] + ["outputs to ring + barrier", "compositing block"]
    for w, wn in ((0, "wave 0"), (1, "wave 5")):
        d = np.diff(s[:, w, :, :15], axis=-1)
        period = s[:, w, 1:, 0] - s[:, w, :-1, 0]
        print(f"== {wn}: tile period median {np.median(period):.0f} cycles, mean {period.mean():.0f}")
        for i, n in enumerate(names):
            x = d[:, :, i].ravel()
            print(f"  {n:28s} median {np.median(x):7.0f}  mean {x.mean():7.0f}  p10 {np.percentile(x, 10):7.0f}  p90 {np.percentile(x, 90):7.0f}")
        ev = (s[:, w, :, 14] - s[:, w, :, 13]) > 2000          # the tiles whose end composites rays
        for nm, a_, b_ in (("trigger -> ray load issued", 13, 16), ("ray load + norm", 16, 17),
                           ("composite_ray (128 samples)", 17, 18), ("closing barrier", 18, 14)):
            x = (s[:, w, :, b_] - s[:, w, :, a_])[ev]
            print(f"  [compositing tiles] {nm:28s} median {np.median(x):7.0f}  mean {x.mean():7.0f}")
        dt = (s[:, w, -1, 14] - s[:, w, 0, 14]).astype(float)
        dr = (s[:, w, -1, 15] - s[:, w, 0, 15]).astype(float)
        print(f"  in-kernel clock (s_memtime / s_memrealtime x 100 MHz) median {np.median(dt / dr) * 100:.0f} MHz")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    args = ap.parse_args()
    if args.build:
        build()
    if args.run:
        run()
    if not (args.build or args.run):
        ap.print_help()
