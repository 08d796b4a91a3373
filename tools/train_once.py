"""A few fused training steps at 4096 rays x 64 samples (config 5) for profiling runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_simple_amd.utils import synthetic
from nerf_simple_amd.utils.nets import Nerf
from nerf_simple_amd.utils.rendering import generate_rays
from nerf_simple_amd.utils.xyz import spherical_to_pose
from nerf_simple_amd.optim import FusedAdam
from nerf_simple_amd.training import train_step

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = Nerf(precision="bf16").to(dev)
net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"))
opt = FusedAdam(net, lr=5e-4)
pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
rays = generate_rays(pose, [800, 800, synthetic.focal_from_fov(800)], dev)[:4096].contiguous()
gt = torch.rand(4096, 3, device=dev)
for i in range(6):
    train_step(net, opt, rays, gt, N, device_rng=True, seed=i)
torch.cuda.synchronize()
