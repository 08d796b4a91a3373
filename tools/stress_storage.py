"""Repeat the 8-bit storage form of the training chain (forward, compositor backward, dX chain, conversions, dW: the C ABI calls of
tests/test_gpu_storage.py run_chain) on the same inputs and compare every buffer bit for bit with the first run -- a hazard in
the hand-scheduled store path (DPP block, conversions and MFMAs in inline asm, where the compiler's hazard recogniser does
not look) would show as stored bytes that move.  Gradients: float atomics in another order, so 1e-5 of the largest entry.
Round 4: 262,144 / 25,641 / 4,096 points, 12 / 40 / 40 repetitions: 0 findings."""
import importlib.util, os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
spec = importlib.util.spec_from_file_location("storage_tests", os.path.join(ROOT, "tests", "test_gpu_storage.py"))
T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
from nerf_simple_amd.utils import synthetic
dev = torch.device("cuda:0")
bad = 0
for (B, N, kind) in ((4096, 64, "default"), (777, 33, "structured"), (64, 64, "default")):
    ref = T.run_chain(dev, synthetic, B, N, True, kind, seed=9)
    for rep in range(40 if B < 4096 else 12):
        cur = T.run_chain(dev, synthetic, B, N, True, kind, seed=9)
        for k in ("acts", "dys", "scratch8", "raw", "d_raw"):
            if not np.array_equal(ref[k], cur[k]):
                bad += 1
                print("NONDETERMINISTIC", B, N, kind, rep, k, int((ref[k] != cur[k]).sum()))
        g = np.abs(ref["grads"] - cur["grads"]).max() / np.abs(ref["grads"]).max()
        if g > 1e-5:
            bad += 1
            print("GRADS MOVE", B, N, rep, g)
    print("shape done", B, N, flush=True)
print("stress done, findings:", bad)
