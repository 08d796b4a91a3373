"""Determinism stress of the training kernels: each of forward_train / mlp_backward is run REPS
times on identical inputs and every output byte is compared with the first run (they are
deterministic kernels: any difference is a race); param_gradients (float atomics: order varies)
is compared with a tolerance.  usage: python tools/stress_train.py [REPS] [B] [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_simple_amd import _lib
from nerf_simple_amd.utils import synthetic
from nerf_simple_amd.utils.nets import Nerf
from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lib = _lib.lib(); dev = torch.device("cuda:0")
net = Nerf().to(dev)
net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
rays = camera_rays([pose], [64, 64, synthetic.focal_from_fov(64)]).float().contiguous().to(dev)[:B].contiguous()
P = B * N
u = torch.rand(B, N, generator=torch.Generator().manual_seed(4)).to(dev)
tbins = torch.linspace(2, 6, N + 1).to(dev)
packed = net.packed_weights(_lib.BF16); image = net.packed_weights(_lib.BF16_BWD)
st = _lib.stream_ptr(dev)
nb = int(lib.nerf_amd_train_activation_bytes(P))
g = torch.randn(P, 4, device=dev) * 1e-3
posx = torch.empty(P, 64, dtype=torch.bfloat16, device=dev); posd = torch.empty(P, 32, dtype=torch.bfloat16, device=dev)
ts = torch.empty(B, N, device=dev)
_lib.check(lib.nerf_amd_sample_encode_bf16(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), 0, 0, 0, _lib.ptr(posx), _lib.ptr(posd),
                                           _lib.ptr(ts), B, N, st), "enc")
scratch = torch.empty(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), dtype=torch.uint8, device=dev)

def fwd():
    raw = torch.zeros(B, N, 4, device=dev); t2 = torch.zeros(B, N, device=dev)
    acts = torch.zeros(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.nerf_amd_mlp_forward_train(_lib.ptr(rays), _lib.ptr(u), _lib.ptr(tbins), _lib.ptr(packed), 0, 0, 0,
                                              _lib.ptr(raw), _lib.ptr(t2), _lib.ptr(acts), B, N, st), "fwd")
    return raw, acts

def bwd(acts):
    dys = torch.zeros(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.nerf_amd_mlp_backward(_lib.ptr(g), _lib.ptr(image), _lib.ptr(acts), _lib.ptr(dys), P, st), "bwd")
    return dys

def dw(acts, dys):
    flat = torch.empty(int(lib.nerf_amd_param_count()), device=dev)
    _lib.check(lib.nerf_amd_param_gradients(_lib.ptr(g), _lib.ptr(acts), _lib.ptr(dys), _lib.ptr(posx), _lib.ptr(posd),
                                            _lib.ptr(scratch), _lib.ptr(flat), P, st), "dw")
    return flat

raw0, acts0 = fwd(); dys0 = bwd(acts0); flat0 = dw(acts0, dys0)
torch.cuda.synchronize()
bad = {"fwd_raw": 0, "fwd_acts": 0, "bwd": 0, "dw": 0}
worst_dw = 0.0
for r in range(REPS):
    raw, acts = fwd()
    if not torch.equal(raw, raw0): bad["fwd_raw"] += 1
    if not torch.equal(acts, acts0):
        bad["fwd_acts"] += 1
        if bad["fwd_acts"] <= 3:
            d = (acts != acts0).nonzero().flatten()
            print("fwd acts diff bytes", d.numel(), "first", d[:6].tolist(), "last", d[-3:].tolist())
    dys = bwd(acts0)
    if not torch.equal(dys, dys0):
        bad["bwd"] += 1
        if bad["bwd"] <= 4:
            a16 = dys.view(torch.int16); b16 = dys0.view(torch.int16)
            d = (a16 != b16).nonzero().flatten()
            ntl = (P + 255) // 256
            rows = []
            for e in d[:400].tolist():
                byte = e * 2
                L, r = divmod(byte, ntl * 131072); tile, r = divmod(r, 131072); ch, r = divmod(r, 4096); pt, r = divmod(r, 16)
                va = float(a16[e:e + 1].view(torch.bfloat16).float()); vb = float(b16[e:e + 1].view(torch.bfloat16).float())
                rows.append((L, tile, pt, ch * 8 + r // 2, round(va, 8), round(vb, 8)))
            print("bwd diff elements", d.numel(), "(layer, tile, point, feature, now, first):")
            for k in range(0, min(len(rows), 36)): print("   ", rows[k])
            import collections
            print("   layers", collections.Counter(r[0] for r in rows), "features", collections.Counter(r[3] for r in rows).most_common(6),
                  "tiles", collections.Counter(r[1] for r in rows).most_common(6))
    flat = dw(acts0, dys0)
    e = float((flat - flat0).abs().max() / flat0.abs().max())
    worst_dw = max(worst_dw, e)
    if e > 1e-3:
        bad["dw"] += 1
        if bad["dw"] <= 3:
            d = ((flat - flat0).abs() > 1e-3 * flat0.abs().max()).nonzero().flatten()
            print("dw diff entries", d.numel(), "first", d[:6].tolist(), "last", d[-3:].tolist(), "rel", e)
torch.cuda.synchronize()
print("REPS", REPS, "P", P, "mismatching runs:", bad, "worst dw rel diff", worst_dw)
