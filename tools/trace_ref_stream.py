"""30 graphed training iterations with the reference's RNG stream (selection + jitter from torch's CPU generator, continued on
the device), for a kernel trace:  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/trace_ref_stream.py
then  python3 tools/trace_ref_stream.py --analyse OUT  prints the timeline of one steady-state iteration.
(--counter: the counter-RNG form; --e4m3: the 8-bit storage form of the saved tensors.)"""
import csv, glob, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:46]))
    rows.sort()
    key = "adam_hyper_kernel" if "--counter" in sys.argv else "mt19937_uniform_kernel<unsigned"     # what opens / closes an iteration
    starts = [i for i, r in enumerate(rows) if key in r[2]]
    a, b = starts[-4], starts[-3]
    t0 = rows[a][0]
    prev_end = rows[a - 1][1]
    print(f"iteration: {(rows[b][0] - rows[a][0]) / 1e3:.1f} us; idle before its first kernel: {(rows[a][0] - prev_end) / 1e3:.1f} us")
    last = prev_end
    for s, e, n in rows[a:b]:
        print(f"  +{(s - t0) / 1e3:8.1f} us  gap {(s - last) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {n}")
        last = max(last, e)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_simple_amd.utils import synthetic
from nerf_simple_amd.utils.nets import Nerf
from nerf_simple_amd.optim import FusedAdam
from nerf_simple_amd.training import GraphedTrainStep
dev = torch.device("cuda:0")
rg = bench.synthetic_ray_table(dev)
net = Nerf(precision="bf16").to(dev)
net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
st = GraphedTrainStep(net, FusedAdam(net, lr=5e-4), 4096, 64, rays_from=rg, device_rng="--counter" in sys.argv, seed=3,
                      storage="e4m3" if "--e4m3" in sys.argv else "bf16")
torch.manual_seed(1)
for _ in range(60):
    st.step()
torch.cuda.synchronize()
