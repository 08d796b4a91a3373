"""Time nerf_amd_param_gradients alone (HIP events) for a given point count.

usage: python tools/dw_sweep.py [P] [path/to/libnerf_amd.so]     (a second build to compare against, as tools/ab_bench.py)
(profiles/r01e_dw_sweep.jsonl holds the sweep of the per-slab cost term that chose the
workgroup split now fixed in dw_gemm.hip; it was run with a temporary environment knob.)
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_simple_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 64
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
lib = _lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
acts = (torch.randn(10, P, 256, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dys = (torch.randn(10, P, 256, device=dev, generator=g) * 0.01).to(torch.bfloat16)
posx = torch.randn(P, 64, device=dev, generator=g).to(torch.bfloat16)
posd = torch.randn(P, 32, device=dev, generator=g).to(torch.bfloat16)
draw = torch.randn(P, 4, device=dev, generator=g)
grads = torch.empty(int(lib.nerf_amd_param_count()), device=dev)
scratch = torch.empty(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), dtype=torch.uint8, device=dev)
s = _lib.stream_ptr(dev)


def run():
    _lib.check(lib.nerf_amd_param_gradients(_lib.ptr(draw), _lib.ptr(acts), _lib.ptr(dys), _lib.ptr(posx),
                                            _lib.ptr(posd), _lib.ptr(scratch), _lib.ptr(grads), P, s),
               "nerf_amd_param_gradients")


for _ in range(5):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
bytes_per_point = 11456
print(json.dumps({"lib": os.path.relpath(_lib.LIB_PATH), "P": P, "ms_min": round(ts[0], 4),
                  "ms_med": round(ts[len(ts) // 2], 4), "TBps_med": round(P * bytes_per_point / ts[len(ts) // 2] / 1e9, 3)}))
