#!/usr/bin/env python3
"""Benchmark of the render hot path: ray-samples/sec at 800x800x128 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One step = one full 800x800 single-view render with 128 samples per ray (the two
launches of render_nerf: sampling + encoding + fused bf16 MLP, then compositing + clip), 640,000 rays
x 128 = 81.92 M ray-samples, synthetic camera and generator-seeded weights
(SURVEY.md section 8d), jitter from the device counter RNG, every input
resident in HBM before the timed region.  With N > 1 the rays of the image are
sharded contiguously over the ranks (one process per GPU, launched by
torch.distributed.run) and each step ends with one RCCL all-gather of the packed
[rgb, disparity] pixels, so the total work is fixed: strong scaling.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the fused MLP kernel against the dense bf16 MFMA peak, from HIP
                events bracketing that kernel inside the timed steps;
  cpu_baseline  the CPU oracle (a PyTorch-CPU port of the reference) timed on
                this box's host cores on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 1_186_816            # 2 x 593,408 MACs at true layer shapes
PEAK_BF16 = 2.5e15                     # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32 = 157.3e12
H = W = 800
N_SAMPLES = 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=16000,
                    help="rays of the CPU-baseline sample (x128 samples; the reference's test batch)")
    return ap.parse_args()


def host_cores():
    """CPU threads this process may actually use: affinity mask, then the cgroup
    CPU quota (a GPU box hands each job a share of a much larger host)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("NERF_BENCH_CPU_THREADS", "32")))


def cpu_baseline(sd, rays_cpu, n_rays):
    """Time the CPU oracle on one n_rays x 128 batch of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    mid = rays_cpu.shape[0] // 2
    rays = rays_cpu[mid:mid + n_rays]
    u = torch.rand(n_rays, N_SAMPLES, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        O.render_nerf(rays[:1000], sd, N_SAMPLES, u=u[:1000])          # warm-up
        t0 = time.perf_counter()
        out = O.render_nerf(rays, sd, N_SAMPLES, u=u)
        dt = time.perf_counter() - t0
    return {"value": n_rays * N_SAMPLES / dt, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} rays x {N_SAMPLES} samples (centre rows of the 800x800 view), "
                      f"oracle/nerf_oracle.py render_nerf fp32, 1 warm-up + 1 timed call, {dt:.2f} s",
            "seconds": dt}, (rays, u, out, O)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs (1-GPU box): NERF_BENCH_BACKEND=gloo NERF_BENCH_SHARE_GPU=1 lets
        # several ranks share cuda:0; the driver's runs use the defaults (nccl = RCCL, one GPU per rank)
        backend = os.environ.get("NERF_BENCH_BACKEND", "nccl")
        if os.environ.get("NERF_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    _lib.lib()                                           # fail loudly if the HIP library is missing

    sd = synthetic.synthetic_state_dict(0, "structured")
    net = Nerf(precision=args.precision).to(dev)
    net.load_state_dict(sd)
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays_cpu = camera_rays([pose], [H, W, synthetic.focal_from_fov(W)])       # [640000, 6]
    n_rays = rays_cpu.shape[0]
    from nerf_simple_amd import parallel
    lo, hi = parallel.shard_range(n_rays, rank, world)
    rays = rays_cpu[lo:hi].to(dev).contiguous()
    # every buffer of a step is allocated once, outside the timed region
    lib = _lib.lib()
    nr = hi - lo
    code = _lib.precision_code(args.precision)
    packed = net.packed_weights(code)
    tbins = torch.linspace(2, 6, N_SAMPLES + 1).to(dev)
    raw = torch.empty((nr, N_SAMPLES, 4), dtype=torch.float32, device=dev)
    ts = torch.empty((nr, N_SAMPLES), dtype=torch.float32, device=dev)
    shard = torch.empty((nr, 4), dtype=torch.float32, device=dev)
    image = torch.empty((n_rays, 4), dtype=torch.float32, device=dev) if world > 1 else shard
    events = []

    def step(record):
        # the two launches of nerf_amd_render_forward / render_nerf, through the C ABI
        # (sampling + encoding + fused MLP, then compositing + clip), then the all-gather
        st = _lib.stream_ptr(dev)
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.nerf_amd_mlp_forward_rays(
            _lib.ptr(rays), None, _lib.ptr(tbins), _lib.ptr(packed), code, _lib.FLAG_DEVICE_RNG, 1234, lo,
            _lib.ptr(raw), _lib.ptr(ts), nr, N_SAMPLES, st), "nerf_amd_mlp_forward_rays")
        if record:
            e1.record()
            events.append((e0, e1))
        _lib.check(lib.nerf_amd_volume_render_pixels(
            _lib.ptr(raw), _lib.ptr(ts), _lib.ptr(rays), _lib.ptr(shard), nr, N_SAMPLES, st),
            "nerf_amd_volume_render_pixels")
        if world > 1:
            parallel.gather_pixels(shard, n_rays, out=image)     # ONE RCCL all-gather per image

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    import gc
    for _ in range(args.warmup):
        step(False)
    gc.collect()
    gc.disable()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    mlp_ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
    if rank == 0:
        total_samples = n_rays * N_SAMPLES * args.steps
        value = total_samples / elapsed
        peak = PEAK_F32 if args.precision == "fp32" else PEAK_BF16      # fp16 and bf16 MFMA rates are equal
        launch_samples = (hi - lo) * N_SAMPLES
        achieved = launch_samples * FLOP_PER_SAMPLE / (mlp_ms * 1e-3) / 1e12
        kern = {"bf16": "nerf_mlp_bf16_16_kernel<true>", "fp16": "nerf_mlp_f16_16_kernel<true>",
                "fp32": "nerf_mlp_f32_kernel<true>"}[args.precision]
        # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (rocprofv3
        # cannot run inside this process): WRITE_SIZE + 2 x FETCH_SIZE (gfx950 wide-read correction),
        # 1.6e6 KB + 2 x 31.9e3 KB for the full 81.92 M-sample launch; algorithmic: 20 B/sample written.
        traffic, traffic_src = None, None
        if world == 1 and args.precision == "bf16":
            traffic = (1.6e6 + 2 * 31.9e3) * 1024
            traffic_src = "profiles/r01e_bench_rocprofv3_summary.txt (WRITE_SIZE + 2*FETCH_SIZE, same command)"
        res = {
            "metric": "ray-samples/sec at 800x800x128", "value": value, "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.precision], "data": "synthetic",
            "config": {"workload": "lego-camera 800x800 single-view render, 128 samples/ray (BASELINE config 3)",
                       "rays": n_rays, "samples_per_ray": N_SAMPLES, "rays_per_launch": hi - lo,
                       "jitter": "device counter RNG", "weights": "synthetic_state_dict(0,'structured')",
                       "parallelism": f"rays sharded x{world}" + (" + RCCL all_gather of [rgb,disp]" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": kern, "achieved": achieved, "peak": peak / 1e12,
                         "unit": "TFLOP/s", "frac": achieved * 1e12 / peak, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel_ms": mlp_ms, "flop_per_sample": FLOP_PER_SAMPLE,
                         "samples_per_launch": launch_samples},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, (crays, cu, cout, O) = cpu_baseline(sd, rays_cpu, args.cpu_rays)
            with torch.no_grad():
                g = render_nerf(crays.to(dev), net, N_SAMPLES, u=cu.to(dev))
            gpu_rgb, cpu_rgb = torch.clip(g[0].cpu(), 0, 1), torch.clip(cout[0], 0, 1)
            base["psnr_gpu_vs_cpu_db"] = float(O.img_psnr(cpu_rgb, gpu_rgb))
            base["max_abs_rgb_err"] = float((g[0].cpu() - cout[0]).abs().max())
            base["gpu_over_cpu"] = value / base["value"]
            res["cpu_baseline"] = base
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
