"""Duration of nerf_amd_select_rays (both launches) by HIP events: counter RNG, B rows from a table of n rays."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_simple_amd.utils.dataload import RayGenerator
dev = torch.device("cuda:0")
for n, B in ((16_000_000, 4096), (16_000_000, 1024), (8192, 4096), (16_000_000, 16384)):
    rg = RayGenerator({"train": torch.rand(n, 6, device=dev)}, {"train": torch.rand(n, 3, device=dev)})
    out = (torch.empty(B, 6, device=dev), torch.empty(B, 3, device=dev), torch.empty(B, dtype=torch.int64, device=dev))
    for _ in range(20):
        rg.select_batch("train", B, device_rng=True, seed=3, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(500):
        rg.select_batch("train", B, device_rng=True, seed=k, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"n {n:9d}  B {B:6d}: {e0.elapsed_time(e1) / 500 * 1e3:7.2f} us per select (scan + gather, eager launches back to back)")
