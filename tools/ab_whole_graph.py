"""One-process A/B of the graphed training iteration replayed as ONE hipGraph (forward ... dW, Adam, re-pack: what
GraphedTrainStep does without an exchange) against the two graphs of a data-parallel step replayed back to back, in both
storage forms of the saved tensors.  2000 iterations per arm, three alternations.
Round 4 on one MI355X: e4m3 0.901-0.908 vs 0.909-0.915 ms (-7 us), bf16 1.142-1.149 either way."""
import sys, time, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from nerf_simple_amd.utils import synthetic
from nerf_simple_amd.utils.nets import Nerf
from nerf_simple_amd.optim import FusedAdam
from nerf_simple_amd.training import GraphedTrainStep
dev = torch.device("cuda:0")
rg = bench.synthetic_ray_table(dev)
def make(whole, storage):
    net = Nerf(precision="bf16").to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
    st = GraphedTrainStep(net, FusedAdam(net, lr=5e-4), 4096, 64, rays_from=rg, device_rng=True, seed=3, storage=storage)
    if not whole:
        st.graph_ab = None
    return st
for storage in ("e4m3", "bf16"):
    sts = {w: make(w, storage) for w in (True, False)}
    for rep in range(3):
        for w, st in sts.items():
            for _ in range(100): st.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2000): st.step()
            torch.cuda.synchronize()
            print(storage, "one graph" if w else "two graphs", f"{(time.perf_counter() - t0) / 2000 * 1e3:.4f} ms", flush=True)
