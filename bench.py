#!/usr/bin/env python3
"""Benchmark of the render hot path: ray-samples/sec at 800x800x128 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--mode render|train] [--precision fp16|bf16|fp32]

--mode render (default; BASELINE config 3).  One step = one full 800x800 single-view render with
128 samples per ray through the C ABI: ONE launch (nerf_amd_render_pixels_forward: sampling +
encoding + fused 16-bit MLP + compositing + clip), 640,000 rays x 128 = 81.92 M ray-samples,
synthetic camera and generator-seeded weights, jitter from the device counter
RNG, every input resident in HBM before the timed region.  With N > 1 the rays of the image are
sharded contiguously over the ranks and each step ends with one RCCL all-gather of the packed
[rgb, disparity] pixels: the total work is fixed (strong scaling).

--mode train (BASELINE config 5).  One step = one optimisation step of reference train.py:47-57 on
4096 rays x 64 samples PER GPU (weak scaling), bf16 kernels, FusedAdam, the step replayed as
hipGraphs; with N > 1 one RCCL all-reduce of the flat 2.38 MB gradient per step.

Launch: `python bench.py --gpus N` starts its own N rank processes (one per GPU, before any GPU
call is made in the parent); under `python -m torch.distributed.run ... bench.py --gpus N` the
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment is used as given.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the dominant kernel against its roofline, duration from HIP events on the launch
                stream inside the timed steps, HBM traffic from the committed PMC summary of this
                same command (null when no committed pass holds this kernel instantiation);
  cpu_baseline  the CPU oracle (a PyTorch-CPU port of the reference) timed on this box's host
                cores on a bounded sample (N=1 only), and the PSNR criterion of BASELINE.json on
                the weights that were timed;
  by_dtype      (render mode) the headline step for BOTH 16-bit operand types with the same number of timed
                steps: value, kernel_ms, frac, psnr_delta_vs_teacher_db, meets_0.05_db -- fp16 (the default)
                and bf16 (the type BASELINE config 3 names) side by side;
  aux           (render mode, N=1) a few seconds each of the other BASELINE.json configurations,
                measured AFTER the timed region of the headline: config 2 (400x400x64, exact-fp32 MFMA), config 4 (800x800 hierarchical 64+128 in one
                library call), config 5 (the whole training iteration -- batch selection from a 16 M-ray table
                included -- as replayed hipGraphs, N = 64 and 128) and shard8 (this GPU as rank 0 of 8: the 80,000-ray
                shard render + its RCCL all-gather, and the training step + its all-reduce, on the wall clock);
  N > 1         per-rank kernel ms, the collective's ms (events around it on the launch stream), and
                self-checks that fail loudly: backend is nccl (= RCCL), one distinct device per rank,
                the rank-sum all-reduce.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 1_186_816            # 2 x 593,408 MACs at true layer shapes
PEAK_BF16 = 2.5e15                     # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32 = 157.3e12
PEAK_HBM = 8.0e12                      # HBM3E spec, same table (6.3e12 achievable)
H = W = 800
N_SAMPLES = 128
TRAIN_RAYS, TRAIN_SAMPLES = 4096, 64   # per GPU (reference configs/lego.yaml:12; BASELINE config 5)
DW_BYTES_PER_POINT = 11_776            # operands nerf_amd_param_gradients reads once per point
DW_BYTES_PER_POINT_E4M3 = 5_856         # the same 15 products from 8-bit operands (d_raw rows 16 wide instead of 32)

PMC_SUMMARIES = [os.path.join(ROOT, "profiles", f) for f in
                 ("r04_bench_fp16_pmc.json", "r04_train_pmc.json", "r04_train_e4m3_pmc.json", "r03_bench_fp16_pmc.json", "r03_bench_bf16_pmc.json", "r03_train_pmc.json",
                  "r02_bench_pmc.json", "r02e_train_pmc.json")]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 150 (render) / 6000 (train): 7-10 s of GPU time")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", default="render", choices=["render", "train"])
    ap.add_argument("--precision", default="fp16", choices=["bf16", "fp16", "fp32"],
                    help="MFMA operand type of the render (train mode is bf16)")
    ap.add_argument("--storage", default="bf16", choices=["bf16", "e4m3"],
                    help="train mode: how the saved activations and dY travel through HBM (GraphedTrainStep(storage=...))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary configurations (render mode, N=1)")
    ap.add_argument("--cpu-rays", type=int, default=16000,
                    help="rays of the CPU-baseline sample (x128 samples; the reference's test batch)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 150 if a.mode == "render" else 6000
    if a.warmup is None:
        a.warmup = 5 if a.mode == "render" else 50
    return a


# ----------------------------------------------------------------------------------------------
# self-launch: N rank processes, started before this process has touched a GPU
# ----------------------------------------------------------------------------------------------
def spawn_ranks(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool supports dmabuf IPC only; with the legacy mode
        # RCCL's (and torch's) cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.
        # The image exports it already; it is repeated here so a rank started from a scrubbed environment still
        # gets it, and an explicit setting of the caller wins.  
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # wait for all ranks; a rank that dies takes the others down with it (they would wait for it in a collective)
    rc, live = 0, list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = max(rc, abs(code))
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


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
    return min(n, 32)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, rays_cpu, n_rays):
    """Time the CPU oracle on one n_rays x 128 batch of the same workload: 1 warm-up + median of 3."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    mid = rays_cpu.shape[0] // 2
    rays = rays_cpu[mid:mid + n_rays]
    u = torch.rand(n_rays, N_SAMPLES, generator=torch.Generator().manual_seed(1234))
    times = []
    with torch.no_grad():
        O.render_nerf(rays[:1000], sd, N_SAMPLES, u=u[:1000])          # warm-up
        for _ in range(3):
            t0 = time.perf_counter()
            out = O.render_nerf(rays, sd, N_SAMPLES, u=u)
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    return {"value": n_rays * N_SAMPLES / dt, "unit": "ray-samples/s", "cores": cores, "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"{n_rays} rays x {N_SAMPLES} samples (centre rows of the 800x800 view), "
                      f"oracle/nerf_oracle.py render_nerf fp32, 1 warm-up + median of 3 timed calls "
                      f"({', '.join(f'{t:.2f}' for t in times)} s)",
            "seconds": dt}, (rays, u, out, O)


def pmc_traffic(kernel, mode, precision=None):
    """HBM bytes per launch of ``kernel`` -- the FULL template instantiation, e.g.
    'nerf_mlp_f16_16_kernel<true, false, true>' -- from a committed PMC summary of THIS command (separate rocprofv3
    --pmc passes, tools/profile_gpu.sh): WRITE_SIZE + 2 x FETCH_SIZE (gfx950 wide-read correction,
    MI355X_MICROARCH.md section HBM), both in KB.  A summary counts only if it was taken from the same mode (and,
    for renders, the same --precision; summaries record their bench arguments) and holds exactly that instantiation;
    otherwise (None, None): the number is a stored constant of a profiled run, never of another workload."""
    for path in PMC_SUMMARIES:
        try:
            summary = json.load(open(path))
        except (OSError, ValueError):
            continue
        cmd = summary.get("bench_args")
        if cmd is not None:
            words = cmd.split()
            got_mode = words[words.index("--mode") + 1] if "--mode" in words else "render"
            got_prec = words[words.index("--precision") + 1] if "--precision" in words else "fp16"
            if got_mode != mode or (mode == "render" and precision is not None and got_prec != precision):
                continue
        for name, c in summary.get("kernels", {}).items():
            if name.startswith(kernel) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (c["WRITE_SIZE"] + 2.0 * c["FETCH_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def live_traffic(kernel, child_args, timeout=100):
    """HBM bytes per launch of ``kernel``, MEASURED IN THIS RUN: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate
    passes, as MI355X_MICROARCH.md's HBM section prescribes) over a 3-launch child run of this same script, then
    WRITE_SIZE + 2 x FETCH_SIZE (KB; gfx950's wide-read correction).  Returns (bytes, note) or (None, reason): the caller
    falls back to the committed summary.  The child is a separate process started with the program itself after `--`."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="nerf_pmc_", dir="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-aux"] + list(child_args)
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "NERF_BENCH_FORCE_DIST"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            if r.returncode != 0:
                return None, f"rocprofv3 pass {counter} failed (rc {r.returncode})"
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel in row.get("Kernel_Name", "").replace("(anonymous namespace)::", "") and row.get("Counter_Name") == counter:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, f"no {counter} rows for {kernel}"
            vals[counter] = sum(got) / len(got)
        return (vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"]) * 1024.0, (
            f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, {len(got)} launches each) of a child "
            f"`bench.py --steps 2`: WRITE_SIZE {vals['WRITE_SIZE']:.0f} KB + 2 x FETCH_SIZE {vals['FETCH_SIZE']:.0f} KB")
    except Exception as e:                                  # the number is evidence, not part of the contract: never fatal
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


_JSON_OUT = None


def emit(res):
    """The one JSON line, on the process's original stdout."""
    out = _JSON_OUT or sys.stdout
    print(json.dumps(res), file=out, flush=True)


def init_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = None
    # NERF_BENCH_FORCE_DIST=1 with one rank: the process group is created anyway and every collective of the multi-GPU
    # step is issued (parallel.force_collectives) -- the RCCL call pattern rehearsed on a single MI355X
    forced = world == 1 and os.environ.get("NERF_BENCH_FORCE_DIST") == "1"
    if forced:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or forced:
        # RCCL prints a version banner on stdout when its first communicator comes up; the contract is ONE JSON line
        # there.  From here on file descriptor 1 is stderr for everything but emit().
        global _JSON_OUT
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs (1-GPU box): NERF_BENCH_BACKEND=gloo NERF_BENCH_SHARE_GPU=1 lets several
        # ranks share cuda:0; the driver's runs use the defaults (nccl = RCCL, one GPU per rank)
        backend = os.environ.get("NERF_BENCH_BACKEND", "nccl")
        if os.environ.get("NERF_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        local_rank %= max(torch.cuda.device_count(), 1)       # a launcher may have narrowed this rank's visible devices
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if forced:
        from nerf_simple_amd import parallel
        parallel.force_collectives(True)
    return world, rank, dev, (dist if (world > 1 or forced) else None), backend


def timed_loop(step, args, dist, dev, world):
    def fence():
        if dist is not None:
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
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def ranks_seen(dist, dev, world):
    """What the collective backend itself reports -- and the self-checks of a multi-GPU run, which raise instead
    of letting a mis-launched job print a number: the backend is nccl (= RCCL on ROCm) unless the rehearsal knob
    NERF_BENCH_BACKEND says otherwise, every rank sits on its own device unless NERF_BENCH_SHARE_GPU=1, and an
    all-reduce of (rank + 1) over the group gives world (world + 1) / 2."""
    if dist is None:
        return {"world_size": 1, "rank_sum_check": True}
    t = torch.tensor([float(dist.get_rank() + 1)], device=dev)
    dist.all_reduce(t)
    props = torch.cuda.get_device_properties(dev)
    ident = (socket.gethostname(), str(getattr(props, "uuid", "")), dev.index)
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    # one device per rank: distinct (host, visible index) pairs, or -- when a launcher narrows each rank's visibility to
    # one GPU, so every rank sees index 0 -- distinct device UUIDs; only a collision in BOTH says ranks share a card
    distinct = (len({(h, i) for h, _, i in idents}) == world or
                (all(u for _, u, _ in idents) and len({(h, u) for h, u, _ in idents}) == world))
    seen = {"world_size": dist.get_world_size(), "rank_sum_check": float(t.item()) == world * (world + 1) / 2,
            "backend": dist.get_backend(), "devices_distinct": distinct, "device_index_per_rank": [i for _, _, i in idents]}
    if not seen["rank_sum_check"] or seen["world_size"] != world:
        raise SystemExit(f"collective self-check failed: {seen}")
    if "NERF_BENCH_BACKEND" not in os.environ and seen["backend"] != "nccl":
        raise SystemExit(f"multi-GPU runs use RCCL (backend 'nccl'), got {seen['backend']!r}")
    if os.environ.get("NERF_BENCH_SHARE_GPU") != "1" and not distinct:
        raise SystemExit(f"ranks share a device: {idents}")
    return seen


def gather_floats(dist, dev, world, value):
    """One float per rank -> list on every rank."""
    if dist is None:
        return [float(value)]
    t = torch.zeros(world, dtype=torch.float64, device=dev)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t)
    return [float(x) for x in t.tolist()]


# ----------------------------------------------------------------------------------------------
# render mode (BASELINE config 3)
# ----------------------------------------------------------------------------------------------
RENDER_KERNEL = {"bf16": "nerf_mlp_bf16_16_kernel<true, false, true>", "fp16": "nerf_mlp_f16_16_kernel<true, false, true>",
                 "fp32": "nerf_mlp_f32_kernel<true, true>"}


def event_timed(fn, steps, warmup, dev):
    """Average duration of fn() in ms from HIP events on the launch stream (all steps in one bracket)."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps


def render_leg(dev, sd, rays, precision, steps, warmup):
    """The headline step -- one launch of the fused render over ``rays`` -- for one operand type, timed exactly like the
    headline (W warm-up launches, then K launches between two device synchronisations on the wall clock, and HIP events
    around every launch on its stream).  Returns the by_dtype entry (PSNR fields are added by the cpu_baseline leg)."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf, packed_status
    lib = _lib.lib()
    net = Nerf(precision=precision).to(dev)
    net.load_state_dict(sd)
    code = _lib.precision_code(precision)
    packed = net.packed_weights(code)
    n = rays.shape[0]
    tb = torch.linspace(2, 6, N_SAMPLES + 1).to(dev)
    px = torch.empty((n, 4), dtype=torch.float32, device=dev)

    def one():
        _lib.check(lib.nerf_amd_render_pixels_forward(_lib.ptr(rays), None, _lib.ptr(tb), _lib.ptr(packed), code, _lib.FLAG_DEVICE_RNG,
                                                      1234, 0, _lib.ptr(px), None, n, N_SAMPLES, _lib.stream_ptr(dev)), "render leg")

    for _ in range(warmup):
        one()
    torch.cuda.synchronize(dev)
    evs = []
    t0 = time.perf_counter()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if packed_status(packed, code):
        raise SystemExit(f"the {precision} render left the operand range")
    kern_ms = sum(a.elapsed_time(b) for a, b in evs) / steps
    samples = n * N_SAMPLES
    peak = PEAK_F32 if precision == "fp32" else PEAK_BF16
    traffic, src = pmc_traffic(RENDER_KERNEL[precision], "render", precision)
    return {"value": samples * steps / elapsed, "unit": "ray-samples/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "warmup": warmup, "kernel": RENDER_KERNEL[precision], "kernel_ms": kern_ms,
            "tflops": samples * FLOP_PER_SAMPLE / (kern_ms * 1e-3) / 1e12,
            "frac": samples * FLOP_PER_SAMPLE / (kern_ms * 1e-3) / peak, "traffic": traffic, "traffic_source": src}


def shard8_bound(dev, sd, rays_800, precision, full_step_ms):
    """What one GPU can say about the 8-GPU step, END TO END (review item: the per-rank host cost around an 8.1 ms kernel
    decides >= 6x, and no 8-GPU node was available to measure the curve).  This process renders exactly what rank 0 of 8
    renders -- 80,000 rays of the 800 x 800 view, ray ids from 0, then the step's all_gather_into_tensor over RCCL (a group
    of ONE rank, the collective issued all the same) -- and times >= 200 such steps on the WALL CLOCK, with HIP events on
    the stream splitting the step into kernel and collective.  Same for the training step with its all-reduce."""
    import torch.distributed as dist
    from nerf_simple_amd import _lib, parallel
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    global _JSON_OUT
    if _JSON_OUT is None:                        # RCCL's banner goes to stdout: keep fd 1 for the one JSON line
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    own_group = not dist.is_initialized()
    if own_group:
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)
    parallel.force_collectives(True)
    try:
        lib = _lib.lib()
        code = _lib.precision_code(precision)
        net = Nerf(precision=precision).to(dev)
        net.load_state_dict(sd)
        packed = net.packed_weights(code)
        n_full = rays_800.shape[0]
        lo, hi = parallel.shard_range(n_full, 0, 8)
        rays = rays_800[lo:hi].contiguous()
        nr = hi - lo
        tb = torch.linspace(2, 6, N_SAMPLES + 1).to(dev)
        shard = torch.empty((nr, 4), dtype=torch.float32, device=dev)
        image = torch.empty((n_full, 4), dtype=torch.float32, device=dev)
        steps, warm = 400, 20
        evs = []

        def step(record):
            if record:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
            _lib.check(lib.nerf_amd_render_pixels_forward(_lib.ptr(rays), None, _lib.ptr(tb), _lib.ptr(packed), code,
                                                          _lib.FLAG_DEVICE_RNG, 1234, lo, _lib.ptr(shard), None, nr, N_SAMPLES,
                                                          _lib.stream_ptr(dev)), "shard render")
            if record:
                ev[1].record()
            dist.all_gather_into_tensor(image[lo:hi], shard)      # one rank: its own slice of the image; 8 ranks: the image
            if record:
                ev[2].record()
                evs.append(ev)

        for _ in range(warm):
            step(False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        torch.cuda.synchronize(dev)
        step_ms = (time.perf_counter() - t0) / steps * 1e3
        kern = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
        coll = sum(e[1].elapsed_time(e[2]) for e in evs) / steps
        out = {"workload": "rank 0 of 8: 80,000 rays x 128 samples of the 800x800 view + all_gather_into_tensor over RCCL "
                           "(process group of one rank, collective issued all the same), wall clock over 400 steps",
               "step_ms": step_ms, "kernel_ms": kern, "collective_ms": coll, "host_gap_ms": step_ms - kern - coll,
               "full_image_step_ms": full_step_ms, "speedup_bound_8": full_step_ms / step_ms,
               "note": "speedup_bound_8 = one-GPU full-image step / this step: what 8 GPUs reach if the wire time of a 1.28 MB "
                       "per-rank all-gather stays inside the measured collective floor; the wire itself needs the 8-GPU node"}
        # the training step of config 5 with its gradient all-reduce issued (2.38 MB, one rank)
        rg = synthetic_ray_table(dev)
        res = {}
        for name, group in (("no_exchange", None), ("all_reduce", dist.group.WORLD)):
            parallel.force_collectives(group is not None)
            tnet = Nerf(precision="bf16").to(dev)
            tnet.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
            st_ = GraphedTrainStep(tnet, FusedAdam(tnet, lr=5e-4), TRAIN_RAYS, TRAIN_SAMPLES, device_rng=True, seed=7, rays_from=rg,
                                   group=group, timing=group is not None)
            for _ in range(50):
                st_.step()
            st_.reset_timing()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2000):
                st_.step()
            torch.cuda.synchronize(dev)
            res[name] = (time.perf_counter() - t0) / 2000 * 1e3
            if group is not None:
                ct = st_.collective_times()
                res["collective_ms"] = ct[0] if ct else None
            del st_, tnet
        out["train"] = {"workload": "config 5 step (4096 x 64, batch selected from the 16 M-ray table in the graph) with and "
                                    "without the flat-gradient all-reduce over RCCL (one rank), wall clock over 2000 steps",
                        "step_ms_no_exchange": res["no_exchange"], "step_ms_all_reduce": res["all_reduce"],
                        "collective_ms": res.get("collective_ms"),
                        "weak_scaling_bound_8": res["no_exchange"] / res["all_reduce"]}
        return out
    finally:
        parallel.force_collectives(False)
        if own_group:
            dist.destroy_process_group()


TABLE_IMAGES = 25                      # 25 images of 800 x 800 = 16 M rays: the reference's lego table (num_train_imgs, configs/lego.yaml)


def synthetic_ray_table(dev, rank=0, seed=100):
    """The training set as the reference holds it (utils/dataload.py:114-129, train.py:33) -- rays_dataset['train'] [n,6]
    and train_imgs [n,3], n = 25 x 800 x 800 = 16 M -- resident in HBM (0.58 GB): rays of 25 synthetic cameras on a ring
    (generated on the device), random target colours."""
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.dataload import RayGenerator
    from nerf_simple_amd.utils.rendering import generate_rays
    from nerf_simple_amd.utils.xyz import spherical_to_pose
    cam = [H, W, synthetic.focal_from_fov(W)]
    rays = torch.cat([generate_rays(torch.from_numpy(spherical_to_pose(4, -30, 360.0 * k / TABLE_IMAGES + 3.0 * rank)).float(), cam, dev)
                      for k in range(TABLE_IMAGES)])
    gen = torch.Generator(device=dev).manual_seed(seed + rank)
    colours = torch.rand((rays.shape[0], 3), generator=gen, device=dev)
    return RayGenerator({"train": rays}, {"train": colours}, cam)


def aux_configs(dev, sd, rays_800):
    """The other BASELINE.json configurations, a few seconds each on this one GPU, through the C ABI / the same host
    objects the tests use.  Returns the ``aux`` object of the JSON line (PSNR of the bf16 render is added by the
    cpu_baseline leg, which owns the CPU renders)."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    import numpy as np
    lib = _lib.lib()
    st = lambda: _lib.stream_ptr(dev)                                    # noqa: E731
    aux = {}

    def net_of(precision, seed=0, kind="structured"):
        n = Nerf(precision=precision).to(dev)
        n.load_state_dict(synthetic.synthetic_state_dict(seed, kind))
        return n

    n_rays = rays_800.shape[0]
    px = torch.empty((n_rays, 4), dtype=torch.float32, device=dev)

    # ---- config 2: 400x400, 64 samples per ray, exact-fp32 MFMA
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    r2 = camera_rays([pose], [400, 400, synthetic.focal_from_fov(400)]).to(dev).contiguous()
    packed = net_of("fp32").packed_weights(_lib.F32)
    tb = torch.linspace(2, 6, 65).to(dev)
    px2 = torch.empty((r2.shape[0], 4), dtype=torch.float32, device=dev)
    ms = event_timed(lambda: _lib.check(lib.nerf_amd_render_pixels_forward(
        _lib.ptr(r2), None, _lib.ptr(tb), _lib.ptr(packed), _lib.F32, _lib.FLAG_DEVICE_RNG, 1234, 0, _lib.ptr(px2), None,
        r2.shape[0], 64, st()), "render fp32"), 12, 2, dev)
    samples = r2.shape[0] * 64
    tf_ = samples * FLOP_PER_SAMPLE / (ms * 1e-3)
    aux["c2"] = {"workload": "config 2: 400x400 render, 64 samples/ray, fp32 (exact-f32 MFMA), one launch", "ms": ms,
                 "ray_samples_per_s": samples / (ms * 1e-3), "kernel": RENDER_KERNEL["fp32"], "tflops": tf_ / 1e12,
                 "frac_of_peak": tf_ / PEAK_F32, "peak_tflops": PEAK_F32 / 1e12, "steps": 12}

    # ---- config 4: 800x800 hierarchical, 64 coarse + 128 fine (192 positions in the fine pass), ONE library call
    nc, nf = net_of("fp16", 0), net_of("fp16", 7)
    pc, pf = nc.packed_weights(_lib.FP16), nf.packed_weights(_lib.FP16)
    h_pose = np.zeros((3, 4), dtype=np.float32)
    h_pose[:] = spherical_to_pose(4, -30, 0)[:3, :4]
    f = float(synthetic.focal_from_fov(W))
    ws = torch.empty(int(lib.nerf_amd_render_hierarchical_workspace_bytes(n_rays, 64, 128)), dtype=torch.uint8, device=dev)
    tbc = torch.linspace(2, 6, 65).to(dev)
    ms = event_timed(lambda: _lib.check(lib.nerf_amd_render_hierarchical_forward(
        h_pose.ctypes.data, H, W, f, 0, n_rays, None, None, _lib.ptr(tbc), _lib.ptr(pc), _lib.ptr(pf), _lib.FP16,
        _lib.FLAG_DEVICE_RNG, 3, _lib.ptr(px), _lib.ptr(ws), 64, 128, st()), "hierarchical"), 8, 2, dev)
    evals = n_rays * (64 + 192)
    tf_ = evals * FLOP_PER_SAMPLE / (ms * 1e-3)
    aux["c4"] = {"workload": "config 4: 800x800 hierarchical 64 coarse + 128 fine in ONE library call (ray generation, coarse "
                             "fused render, sample_pdf, fine fused render on 192 merged positions), fp16 operands, 1 GPU; "
                             "parity unpinned (the reference has no hierarchical sampling)", "ms": ms,
                 "mlp_evaluations_per_s": evals / (ms * 1e-3), "kernel": RENDER_KERNEL["fp16"] + " x2 + sample_pdf_kernel + generate_rays_kernel",
                 "tflops": tf_ / 1e12, "frac_of_peak": tf_ / PEAK_BF16, "peak_tflops": PEAK_BF16 / 1e12, "steps": 8}
    del ws, px, px2

    # ---- config 5: the training step (train.py:47-57), replayed hipGraphs, at the sample count BASELINE names (64)
    #      and at the reference's own (Nf = 128, configs/lego.yaml:6)
    aux["c5"] = {}
    rg = synthetic_ray_table(dev)
    n_table = int(rg.rays_dataset["train"].shape[0])
    for N, steps in ((64, 600), (128, 300)):
        net = net_of("bf16", 0, "default")
        # the whole iteration of train.py:47-57 inside the replayed graphs: rg.select + the colour gather from the 16 M-row
        # tables (a fresh batch every step, selected one step ahead beside the dX chain), fresh jitter, forward, backward,
        # Adam, re-pack
        stepper = GraphedTrainStep(net, FusedAdam(net, lr=5e-4), TRAIN_RAYS, N, device_rng=True, seed=7, rays_from=rg)
        ms = event_timed(stepper.step, steps, 30, dev)
        P = TRAIN_RAYS * N
        aux["c5"][f"N{N}"] = {"workload": f"config 5: train.py iteration, 4096 rays x {N} samples selected on the device from a "
                                          f"{n_table}-ray table every step, bf16, FusedAdam, hipGraph replay, 1 GPU",
                              "ms": ms, "ray_samples_per_s": P / (ms * 1e-3), "table_rays": n_table,
                              "kernel": "select_partner_kernel + select_scan_kernel + select_gather_kernel + nerf_mlp_bf16_16_kernel<true, true, false> + composite_backward_kernel + nerf_mlp_bwd_kernel + dw_gemm_kernel + adam_hyper_kernel + pack_train_kernel",
                              "step_mfma_frac": 3 * FLOP_PER_SAMPLE * P / (ms * 1e-3) / PEAK_BF16, "peak_tflops": PEAK_BF16 / 1e12,
                              "final_loss": float(stepper.loss), "steps": steps}
        del stepper, net
        # the same iteration with the saved activations and dY in the 8-bit storage form (same gradient criterion:
        # tests/test_gpu_trajectory.py)
        net = net_of("bf16", 0, "default")
        stepper = GraphedTrainStep(net, FusedAdam(net, lr=5e-4), TRAIN_RAYS, N, device_rng=True, seed=7, rays_from=rg, storage="e4m3")
        ms8 = event_timed(stepper.step, steps, 30, dev)
        aux["c5"][f"N{N}"]["storage_e4m3"] = {
            "ms": ms8, "ray_samples_per_s": P / (ms8 * 1e-3), "final_loss": float(stepper.loss), "steps": steps,
            "kernel": "... + nerf_mlp_train_e4m3_kernel + composite_backward_kernel + nerf_mlp_bwd_e4m3_kernel + rows_to_e4m3_kernel + dw_gemm_e4m3_kernel + ...",
            "what": "saved activations and dY as e4m3 + one exponent per 32 features x 32 points; dW products on the block-scaled 8-bit MFMA"}
        del stepper, net
    del rg
    return aux


def run_render(args):
    world, rank, dev, dist, backend = init_rank(args)
    multi = dist is not None                 # collectives are issued: more than one rank, or the single-rank rehearsal
    from nerf_simple_amd import _lib, parallel
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.rendering import render_nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()                                     # fails loudly if the HIP library is missing

    sd = synthetic.synthetic_state_dict(0, "structured")
    net = Nerf(precision=args.precision).to(dev)
    net.load_state_dict(sd)
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays_cpu = camera_rays([pose], [H, W, synthetic.focal_from_fov(W)])       # [640000, 6]
    n_rays = rays_cpu.shape[0]
    lo, hi = parallel.shard_range(n_rays, rank, world)
    rays = rays_cpu[lo:hi].to(dev).contiguous()
    # every buffer of a step is allocated once, outside the timed region
    nr = hi - lo
    code = _lib.precision_code(args.precision)
    packed = net.packed_weights(code)
    tbins = torch.linspace(2, 6, N_SAMPLES + 1).to(dev)
    nws = int(lib.nerf_amd_render_workspace_bytes(code, nr, N_SAMPLES))       # 0 for the fused render
    ws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws else None
    shard = torch.empty((nr, 4), dtype=torch.float32, device=dev)
    image = torch.empty((n_rays, 4), dtype=torch.float32, device=dev) if multi else shard
    events = []

    def step(record):
        st = _lib.stream_ptr(dev)
        if record:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 if multi else 2)]
            ev[0].record()
        _lib.check(lib.nerf_amd_render_pixels_forward(
            _lib.ptr(rays), None, _lib.ptr(tbins), _lib.ptr(packed), code, _lib.FLAG_DEVICE_RNG, 1234, lo,
            _lib.ptr(shard), _lib.ptr(ws), nr, N_SAMPLES, st), "nerf_amd_render_pixels_forward")
        if record:
            ev[1].record()
        if multi:
            parallel.gather_pixels(shard, n_rays, out=image)     # ONE RCCL all-gather per image
            if record:
                ev[2].record()
        if record:
            events.append(ev)

    elapsed = timed_loop(step, args, dist, dev, world)
    seen = ranks_seen(dist, dev, world)
    kern_ms = sum(e[0].elapsed_time(e[1]) for e in events) / max(len(events), 1)
    coll_ms = sum(e[1].elapsed_time(e[2]) for e in events) / max(len(events), 1) if multi else 0.0
    kern_ms_ranks = gather_floats(dist, dev, world, kern_ms)
    coll_ms_ranks = gather_floats(dist, dev, world, coll_ms)
    from nerf_simple_amd.utils.nets import packed_status
    range_flags = packed_status(packed, code)
    if range_flags:
        raise SystemExit(f"the timed render left the operand range (status word {range_flags}): the number would be of NaN pixels")
    if rank == 0:
        total_samples = n_rays * N_SAMPLES * args.steps
        value = total_samples / elapsed
        peak = PEAK_F32 if args.precision == "fp32" else PEAK_BF16      # fp16 and bf16 MFMA rates are equal
        launch_samples = nr * N_SAMPLES
        achieved = launch_samples * FLOP_PER_SAMPLE / (kern_ms * 1e-3) / 1e12
        kern = RENDER_KERNEL[args.precision]
        traffic, traffic_src = pmc_traffic(kern, "render", args.precision) if not multi else (None, None)
        traffic_kind = ("HBM bytes per launch from the committed rocprofv3 PMC passes of this command "
                        "(WRITE_SIZE + 2 x FETCH_SIZE); not re-measured in this run")
        if not multi and not args.no_aux and os.environ.get("NERF_BENCH_LIVE_PMC", "1") == "1":
            live, note = live_traffic(kern, ["--precision", args.precision])
            if live is not None:
                stored = traffic
                traffic, traffic_src, traffic_kind = live, "live", note
                if stored:
                    traffic_kind += f"; committed summary: {stored:.0f} B ({(live / stored - 1) * 100:+.1f} % against it)"
            else:
                traffic_kind += f" (live pass unavailable: {note})"
        res = {
            "metric": "ray-samples/sec at 800x800x128", "value": value, "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.precision], "data": "synthetic",
            "config": {"workload": "lego-camera 800x800 single-view render, 128 samples/ray (BASELINE config 3)",
                       "rays": n_rays, "samples_per_ray": N_SAMPLES, "rays_per_launch": nr,
                       "launches_per_step": 1 if nws == 0 else 2,
                       "jitter": "device counter RNG", "weights": "synthetic_state_dict(0,'structured')",
                       "operands": {"fp16": "fp16 MFMA operands, fp32 accumulate (same dense peak as bf16; the 16-bit type that "
                                            "meets the 0.05 dB PSNR target on these weights: DESIGN.md section 2; bf16 is timed "
                                            "in aux.bf16 and misses that target: known gap)",
                                    "bf16": "bf16 MFMA operands, fp32 accumulate", "fp32": "exact-f32 MFMA"}[args.precision],
                       "parallelism": f"rays sharded x{world}" + (" + all_gather of [rgb,disp]" if multi else "")},
            "ranks": seen,
            "kernel_ms_per_rank": kern_ms_ranks,
            "collective_ms": max(coll_ms_ranks) if multi else 0.0,
            "collective_ms_per_rank": coll_ms_ranks if multi else [],
            "collective": "all_gather_into_tensor of the packed [rgb, disparity] pixels, events on the launch stream "
                          "from the end of the render kernel to the end of the gather" if multi else None,
            "roofline": {"bound": "mfma", "kernel": kern, "achieved": achieved, "peak": peak / 1e12,
                         "unit": "TFLOP/s", "frac": achieved * 1e12 / peak, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "traffic_kind": traffic_kind,
                         "algorithmic_hbm_bytes": nr * 40,
                         "kernel_ms": kern_ms, "flop_per_sample": FLOP_PER_SAMPLE,
                         "samples_per_launch": launch_samples},
        }
        res["by_dtype"] = {res["dtype"]: {"value": value, "unit": "ray-samples/s", "ms_per_step": elapsed / args.steps * 1e3,
                                          "steps": args.steps, "warmup": args.warmup, "kernel": kern, "kernel_ms": kern_ms,
                                          "tflops": achieved, "frac": achieved * 1e12 / peak, "traffic": traffic,
                                          "traffic_source": traffic_src, "headline": True}}
        if not multi and not args.no_aux:
            # the other 16-bit operand type with the SAME number of timed steps: both types carry equal statistical weight
            # (fp16 is the default because it meets the 0.05 dB criterion on every weight set tried; bf16 is the type
            # BASELINE config 3 names -- DESIGN.md section 2)
            other = {"fp16": "bf16", "bf16": "fp16"}.get(args.precision)
            if other:
                res["by_dtype"][{"bf16": "bf16", "fp16": "f16"}[other]] = render_leg(dev, sd, rays, other, args.steps, args.warmup)
            res["aux"] = aux_configs(dev, sd, rays)
            try:
                res["aux"]["shard8"] = shard8_bound(dev, sd, rays, args.precision, elapsed / args.steps * 1e3)
            except Exception as e:                       # a rehearsal: never take the headline line down with it
                res["aux"]["shard8"] = {"error": f"{type(e).__name__}: {e}"}
        if not multi and not args.no_cpu_baseline:
            base, (crays, cu, cout, O) = cpu_baseline(sd, rays_cpu, args.cpu_rays)
            with torch.no_grad():
                g = render_nerf(crays.to(dev), net, N_SAMPLES, u=cu.to(dev))
                # BASELINE's PSNR criterion on the weights that were timed: |PSNR(GPU,T) - PSNR(CPU,T)| against
                # the CPU render T of a perturbed teacher, reference PSNR formula
                teacher = synthetic.perturbed_state_dict(sd, seed=1, rel=0.02)
                T = torch.clip(O.render_nerf(crays, teacher, N_SAMPLES, u=cu)[0], 0, 1)
            gpu_rgb, cpu_rgb = torch.clip(g[0].cpu(), 0, 1), torch.clip(cout[0], 0, 1)
            base["psnr_gpu_vs_cpu_db"] = float(O.img_psnr(cpu_rgb, gpu_rgb))
            base["psnr_cpu_vs_teacher_db"] = float(O.img_psnr(T, cpu_rgb))
            base["psnr_delta_vs_teacher_db"] = float(O.img_psnr(T, gpu_rgb)) - base["psnr_cpu_vs_teacher_db"]
            base["max_abs_rgb_err"] = float((g[0].cpu() - cout[0]).abs().max())
            base["gpu_over_cpu"] = value / base["value"]
            res["cpu_baseline"] = base
            for key, prec in (("f16", "fp16"), ("bf16", "bf16"), ("f32", "fp32")):
                if key not in res["by_dtype"]:
                    continue
                with torch.no_grad():
                    gb = g if prec == args.precision else render_nerf(crays.to(dev), net, N_SAMPLES, u=cu.to(dev), precision=prec)
                p_rgb = torch.clip(gb[0].cpu(), 0, 1)
                d = float(O.img_psnr(T, p_rgb)) - base["psnr_cpu_vs_teacher_db"]
                res["by_dtype"][key].update({"psnr_gpu_vs_cpu_db": float(O.img_psnr(cpu_rgb, p_rgb)), "psnr_delta_vs_teacher_db": d,
                                             "meets_0.05_db": abs(d) <= 0.05})
            if "bf16" in res["by_dtype"] and not res["by_dtype"]["bf16"].get("meets_0.05_db", True):
                res["by_dtype"]["bf16"]["note"] = ("bf16's 8-bit weight mantissa shifts the image systematically on these high-gain "
                                                   "weights (DESIGN.md section 2): the 0.05 dB criterion is a known gap of the bf16 "
                                                   "operand mode on them, which is why fp16 operands are the default")
        emit(res)
    if multi:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# train mode (BASELINE config 5)
# ----------------------------------------------------------------------------------------------
def run_train(args):
    world, rank, dev, dist, backend = init_rank(args)
    multi = dist is not None
    from nerf_simple_amd import _lib, parallel
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep, lr_decay_factor
    from nerf_simple_amd.utils import synthetic
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()
    B, N = TRAIN_RAYS, TRAIN_SAMPLES
    P = B * N
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    parallel.broadcast_parameters(net)
    opt = FusedAdam(net, lr=5e-4)
    # the reference's training set shape, resident in HBM: 16 M rays + colours per rank (its own cameras and targets)
    rg = synthetic_ray_table(dev, rank)
    n_table = int(rg.rays_dataset["train"].shape[0])
    # the whole iteration inside the replayed graphs: a fresh batch of 4096 rays selected from the table (rg.select + the
    # colour gather, train.py:47-49) and fresh stratified jitter every step, both from the counter RNG (seed + step count
    # in device memory: the replayed graphs carry no per-step argument); ray_id0 offsets each rank's draws
    stepper = GraphedTrainStep(net, opt, B, N, group=(dist.group.WORLD if multi else None), timing=multi,
                               buckets=int(os.environ.get("NERF_BENCH_BUCKETS", "1")), device_rng=True, seed=1234,
                               ray_id0=rank * B, rays_from=rg, storage=args.storage)
    e4m3 = args.storage == "e4m3"
    dw_kernel = "dw_gemm_e4m3_kernel(" if e4m3 else "dw_gemm_kernel("
    dw_bytes = DW_BYTES_PER_POINT_E4M3 if e4m3 else DW_BYTES_PER_POINT
    decay = lr_decay_factor(5e-4, 5e-5, 10000)              # reference configs/lego.yaml lr_init / lr_final shape
    it = [0]

    timed = [False]

    def step(record):
        if record and not timed[0]:
            timed[0] = True
            stepper.reset_timing()                  # the exchange times are of the timed steps only
        stepper.step(decay=decay)
        it[0] += 1

    elapsed = timed_loop(step, args, dist, dev, world)
    coll = stepper.collective_times() if multi else None      # (span, exposed) ms per timed step
    # the other storage form of the saved tensors, timed alike in the same process (N = 1): a second module from the same
    # initial weights, its own optimizer and graphs, the same table and seeds
    other = None
    if not multi and not args.no_aux:
        o_storage = "bf16" if e4m3 else "e4m3"
        onet = Nerf(precision="bf16").to(dev)
        onet.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        ost = GraphedTrainStep(onet, FusedAdam(onet, lr=5e-4), B, N, device_rng=True, seed=1234, ray_id0=rank * B, rays_from=rg,
                               storage=o_storage)
        for _ in range(args.warmup):
            ost.step(decay=decay)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ost.step(decay=decay)
        torch.cuda.synchronize(dev)
        o_ms = (time.perf_counter() - t0) / args.steps * 1e3
        other = {"storage": o_storage, "ms_per_step": o_ms, "value": P / (o_ms * 1e-3), "steps": args.steps, "warmup": args.warmup,
                 "final_loss": float(ost.loss)}
        del ost, onet
    seen = ranks_seen(dist, dev, world)
    loss = float(stepper.loss)
    # duration of the dominant kernel (the dW products: nerf_amd_param_gradients_finish*): 20 more launches on the same
    # buffers right after the timed steps, each bracketed by events on the launch stream.  What precedes it in a step -- the
    # zero fill of the gradient vector and the d_raw pack (_begin; the 8-bit form: the conversion of the narrow operands) --
    # is launched in front of the first event: the products add into the vector with atomics.
    st = _lib.stream_ptr(dev)
    image = net.packed_weights(_lib.BF16_BWD)
    evs = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.check(lib.nerf_amd_param_gradients_begin(_lib.ptr(stepper.d_raw), _lib.ptr(stepper.scratch), _lib.ptr(stepper.grads), P, st),
                   "nerf_amd_param_gradients_begin")
        if e4m3:
            _lib.check(lib.nerf_amd_param_gradients_convert_e4m3(_lib.ptr(stepper.posx), _lib.ptr(stepper.posd), _lib.ptr(stepper.scratch),
                                                                 _lib.ptr(stepper.scratch8), P, 3, st), "nerf_amd_param_gradients_convert_e4m3")
            e0.record()
            _lib.check(lib.nerf_amd_param_gradients_finish_e4m3(_lib.ptr(stepper.acts), _lib.ptr(stepper.dys), _lib.ptr(stepper.scratch8),
                                                                _lib.ptr(stepper.grads), P, 0, st), "nerf_amd_param_gradients_finish_e4m3")
        else:
            e0.record()
            _lib.check(lib.nerf_amd_param_gradients_finish(_lib.ptr(stepper.acts), _lib.ptr(stepper.dys), _lib.ptr(stepper.posx),
                                                           _lib.ptr(stepper.posd), _lib.ptr(stepper.scratch), _lib.ptr(stepper.grads), P, st),
                       "nerf_amd_param_gradients_finish")
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize(dev)
    dw_ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    dw_ms_ranks = gather_floats(dist, dev, world, dw_ms)
    span_ranks = gather_floats(dist, dev, world, coll[0] if coll else 0.0)
    exposed_ranks = gather_floats(dist, dev, world, coll[1] if coll else 0.0)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * P * args.steps / elapsed
        achieved = dw_bytes * P / (dw_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dw_kernel, "train") if not multi else (None, None)
        traffic_kind = "HBM bytes per launch from the committed rocprofv3 PMC passes of this command; not re-measured in this run"
        if not multi and not args.no_aux and os.environ.get("NERF_BENCH_LIVE_PMC", "1") == "1":
            live, note = live_traffic(dw_kernel, ["--mode", "train", "--storage", args.storage])
            if live is not None:
                stored = traffic
                traffic, traffic_src, traffic_kind = live, "live", note
                if stored:
                    traffic_kind += f"; committed summary: {stored:.0f} B ({(live / stored - 1) * 100:+.1f} % against it)"
            else:
                traffic_kind += f" (live pass unavailable: {note})"
        res = {
            "metric": "training ray-samples/sec (forward + backward + Adam) at 4096 rays x 64 samples per GPU",
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "train.py step: 4096 rays x 64 samples per GPU, bf16, FusedAdam, hipGraph replay "
                                   "(BASELINE config 5)",
                       "storage": ("saved activations and dY as e4m3 + one exponent per 32 features x 32 points (the dW products on the "
                                   "block-scaled 8-bit MFMA); forward, loss and dX chain compute in bf16 as before" if e4m3 else
                                   "saved activations and dY as bf16"),
                       "jitter": "fresh per step, device counter RNG inside the timed step",
                       "batch": f"fresh per step: rg.select + colour gather (train.py:47-49) from a {n_table}-ray table in HBM, "
                                "inside the timed step (nodes of the replayed graph: each replay selects the next step's batch beside its dX chain)",
                       "table_rays": n_table,
                       "rays_per_gpu": B, "samples_per_ray": N, "global_batch_rays": B * world,
                       "parallelism": f"data-parallel x{world}" + (" + all_reduce of the flat 2.38 MB gradient" if multi else "")},
            "ranks": seen, "final_loss": loss,
            "by_storage": (None if other is None else {
                args.storage: {"headline": True, "ms_per_step": ms, "value": value, "steps": args.steps, "warmup": args.warmup,
                               "final_loss": loss},
                other["storage"]: {k: v for k, v in other.items() if k != "storage"},
                "criterion": "both forms pass the same gradient criterion (every tensor's error <= half the reference's minibatch "
                             "deviation, fixture G6c) and the 60-iteration trajectory bands: tests/test_gpu_trajectory.py"}),
            "kernel_ms_per_rank": dw_ms_ranks,
            "collective_ms": max(span_ranks) if multi else 0.0,
            "collective_exposed_ms": max(exposed_ranks) if multi else 0.0,
            "collective_ms_per_rank": span_ranks if multi else [],
            "collective": ("one all-reduce of the flat 2.38 MB gradient between the two graphs of the step (events on the "
                           "launch stream); fully exposed" if not stepper.bucketed else
                           "two all-reduces of the flat gradient (1.27 MB, then 1.12 MB); collective_ms = from the end of the "
                           "late-layer gradient launch to both reduced, collective_exposed_ms = the part behind the end of the "
                           "head-gradient launch that runs beside the first exchange") if multi else None,
            "roofline": {"bound": "hbm", "kernel": dw_kernel.rstrip("(") + " (the one launch of nerf_amd_param_gradients_finish" +
                                   ("_e4m3" if e4m3 else "") + ": all 14 products + the bias sums)",
                         "achieved": achieved, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": achieved * 1e9 / PEAK_HBM,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_kind": traffic_kind, "kernel_ms": dw_ms,
                         "algorithmic_bytes_per_point": dw_bytes,
                         "kernel_ms_note": "median of 20 extra launches on the step's own buffers after the timed region",
                         "step_mfma_frac": 3 * FLOP_PER_SAMPLE * P / (ms * 1e-3) / PEAK_BF16},
        }
        emit(res)
    if multi:
        dist.destroy_process_group()


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus)                           # never returns
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    if args.mode == "train":
        run_train(args)
    else:
        run_render(args)


if __name__ == "__main__":
    main()
