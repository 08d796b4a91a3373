"""CPU ORACLE -- test infrastructure, NOT product code.

A from-scratch PyTorch-CPU restatement of the reference hot path
(UCSD-Comp-Imaging/Nerf-Simple), functional style over a plain state dict.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the shipped render path never does (it
fails loudly when the HIP library is missing instead of falling back here).

Parity status: PINNED.  Every function below is checked bit-for-bit (same
torch build, CPU) against outputs of the reference source imported in the
build container -- fixtures under tests/golden/, produced by
tests/golden/make_golden.py (which holds no reference code).  The reference
pins torch==1.11.0 (requirements.txt:2); only torch 2.10.0 exists offline, so
the goldens are "reference source on torch 2.10 CPU" (SURVEY.md section 8c).

EXCEPTION -- parity UNPINNED: ``sample_pdf`` / ``render_hierarchical`` (BASELINE
config 4's fine pass).  Hierarchical sampling does not exist in the reference
(README.md:3, configs/lego.yaml:7, utils/nets.py:45-49), so there is nothing to
capture goldens from; they restate the NeRF paper and only the two constituent
render passes are pinned (through render_nerf with explicit ts).

Each function cites the reference file:line it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# positional encoding  (reference utils/xyz.py:6-36)
# --------------------------------------------------------------------------
def gamma(x, L=4):
    """[sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] along dim 1
    (utils/xyz.py:6-14).  No pi factor; the |x|>1 warning is not reproduced
    (it is not part of the results)."""
    cols = []
    for i in range(L):
        cols.append(torch.sin(2 ** i * x))
        cols.append(torch.cos(2 ** i * x))
    return torch.cat(cols, dim=1)


def positional_encoder(vec, Lp=10, Ld=4):
    """vec [P,6] -> posx [P,3+6Lp], posd [P,3+6Ld]; grouped per coordinate:
    [x,y,z, g(x), g(y), g(z)]  (utils/xyz.py:16-36)."""
    c = [vec[:, i:i + 1] for i in range(6)]
    posx = torch.cat(c[0:3] + [gamma(c[i], Lp) for i in range(3)], dim=1)
    posd = torch.cat(c[3:6] + [gamma(c[i], Ld) for i in range(3, 6)], dim=1)
    return posx, posd


# --------------------------------------------------------------------------
# the MLP  (reference utils/nets.py:8-43)
# --------------------------------------------------------------------------
def nerf_forward(sd, v, Lp=10, Ld=4, return_hidden=False):
    """v [P,6] -> [P,4] = [r,g,b (raw, no sigmoid), sigma (raw)].

    Data-flow of utils/nets.py:34-43: five ReLU'd layers, skip-concat [h ; x]
    (h first), two more ReLU'd layers, sigma head on the post-ReLU features,
    a linear 256->256 WITHOUT activation, colour head on [h ; d]."""
    x, d = positional_encoder(v, Lp, Ld)
    h = x
    for i in (0, 2, 4, 6, 8):
        h = F.relu(F.linear(h, sd[f"layers_0.{i}.weight"], sd[f"layers_0.{i}.bias"]))
    h5 = h
    h = F.relu(F.linear(torch.cat([h, x], dim=1),
                        sd["skip_conn_layer.0.weight"], sd["skip_conn_layer.0.bias"]))
    for i in (0, 2):
        h = F.relu(F.linear(h, sd[f"layers_1.{i}.weight"], sd[f"layers_1.{i}.bias"]))
    h8 = h
    sigma = F.linear(h8, sd["sigma_fc.0.weight"], sd["sigma_fc.0.bias"])
    h9 = F.linear(h8, sd["layers_2.weight"], sd["layers_2.bias"])
    c = F.relu(F.linear(torch.cat([h9, d], dim=1),
                        sd["color_fc.0.weight"], sd["color_fc.0.bias"]))
    rgb = F.linear(c, sd["color_fc.2.weight"], sd["color_fc.2.bias"])
    out = torch.cat([rgb, sigma], dim=1)
    if return_hidden:
        return out, {"h5": h5, "h8": h8, "h9": h9}
    return out


# --------------------------------------------------------------------------
# sampling + compositing  (reference utils/rendering.py:13-85)
# --------------------------------------------------------------------------
def sample_ts(u, tn=2, tf=6):
    """Stratified sample positions from jitter u [B,N] in [0,1)
    (utils/rendering.py:25-29): ts = (t_bins[1]-t_bins[0]) * u + t_bins[:-1]."""
    N = u.shape[1]
    t_bins = torch.linspace(tn, tf, N + 1)
    return (t_bins[1] - t_bins[0]) * u + t_bins[:-1]


def query_points(rays, ts):
    """rays [B,6], ts [B,N] -> (query_pts [B*N,6], unit dirs [B,3])
    (utils/rendering.py:31-40).  Positions use the UN-normalised direction;
    the network input and the compositor get the normalised one."""
    o, d = rays[:, :3], rays[:, 3:]
    N = ts.shape[1]
    locs = o.unsqueeze(-1) + d.unsqueeze(-1) * ts.unsqueeze(1)       # [B,3,N]
    dn = d / torch.norm(d, dim=1, keepdim=True)
    q = torch.cat((locs, dn.unsqueeze(-1).expand(-1, -1, N)), dim=1)  # [B,6,N]
    return q.permute(0, 2, 1).reshape(-1, 6), dn


def volume_render(nerf_outs, ts, dirs):
    """nerf_outs [B,N,4], ts [B,N], dirs [B,3] -> (rgb, disp, alpha, acc, w)
    (utils/rendering.py:47-85).  softplus(beta=1, threshold=20); last delta
    1e10; transmittance = exclusive cumprod of (1 - alpha + 1e-10); the second
    output is DISPARITY 1/max(1e-10, depth/acc) (NaN when acc == 0)."""
    deltas = ts[:, 1:] - ts[:, :-1]
    deltas = torch.cat((deltas, 1e10 * torch.ones_like(deltas[:, :1])), dim=1)
    deltas = deltas * torch.norm(dirs[..., None, :], dim=-1)
    sigma = nerf_outs[..., 3]
    alpha = 1 - torch.exp(-F.softplus(sigma) * deltas)
    ones = torch.ones((alpha.shape[0], 1))
    w = alpha * torch.cumprod(torch.cat([ones, 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb = torch.sum(w.unsqueeze(-1) * nerf_outs[..., :3], dim=1)
    depth = torch.sum(w * ts, dim=-1)
    acc = torch.sum(w, dim=-1)
    disp = torch.max(1e-10 * torch.ones_like(depth), depth / torch.sum(w, dim=-1))
    disp = 1. / disp
    return rgb, disp, alpha, acc, w


def render_nerf(rays, sd, N, tn=2, tf=6, u=None, ts=None):
    """The reference render_nerf(rays, net, N, tn, tf) (utils/rendering.py:13-45)
    with the net given as a state dict.  ``u`` [B,N]: explicit jitter; when
    both u and ts are None one torch.rand(B,N) is drawn from the CPU default
    generator exactly as the reference does (:28)."""
    B = rays.shape[0]
    if ts is None:
        if u is None:
            u = torch.rand(B, N)
        ts = sample_ts(u, tn, tf)
    q, dn = query_points(rays, ts)
    out = nerf_forward(sd, q).reshape(B, N, 4)
    return volume_render(out, ts, dn)


# --------------------------------------------------------------------------
# hierarchical sampling -- NOT in the reference (README.md:3; configs/lego.yaml:7;
# utils/nets.py:45-49): PARITY UNPINNED.  Restates the NeRF paper's sample_pdf
# (Mildenhall et al. 2020, section 5.2) so the HIP sampler has a checker.
# --------------------------------------------------------------------------
def sample_pdf(ts, w, u):
    """ts [B,Nc] coarse positions, w [B,Nc] coarse weights, u [B,Nf] in [0,1)
    -> sorted [B,Nc+Nf]: Nf inverse-CDF samples of the interior bins
    (mids of ts, weights w[1:-1] + 1e-5) merged with ts."""
    bins = 0.5 * (ts[:, 1:] + ts[:, :-1])
    wt = w[:, 1:-1] + 1e-5
    pdf = wt / torch.sum(wt, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    z = b0 + (u - c0) / denom * (b1 - b0)
    return torch.sort(torch.cat([ts, z], -1), -1).values


def render_hierarchical(rays, sd_coarse, sd_fine, Nc, Nf, u_c, u_f, tn=2, tf=6):
    """Coarse pass (Nc stratified samples) -> sample_pdf -> fine pass on the
    Nc+Nf merged positions.  Returns (fine 5-tuple, coarse 5-tuple, ts_fine)."""
    ts_c = sample_ts(u_c, tn, tf)
    coarse = render_nerf(rays, sd_coarse, Nc, tn, tf, ts=ts_c)
    ts_f = sample_pdf(ts_c, coarse[4], u_f)
    fine = render_nerf(rays, sd_fine, Nc + Nf, tn, tf, ts=ts_f)
    return fine, coarse, ts_f


# --------------------------------------------------------------------------
# cameras  (reference utils/xyz.py:38-91, utils/rendering.py:129-134)
# --------------------------------------------------------------------------
def rays_single_cam(cam_params):
    """[H,W,f] -> camera-frame directions [3, H*W], row-major h*W+w:
    dir(h,w) = ((w - W//2)/f, -(h - H//2)/f, -1)  (utils/xyz.py:38-52)."""
    H, W, f = cam_params
    hh = (torch.arange(H) - H // 2).reshape(H, 1).expand(H, W)
    ww = (torch.arange(W) - W // 2).reshape(1, W).expand(H, W)
    d = torch.stack((ww / f, -hh / f, -torch.ones_like(ww))).float()
    return d.reshape(3, -1)


def spherical_to_pose(r, theta, phi):
    """Camera-to-world 4x4 for spherical (r, theta deg, phi deg):
    Rz(phi) . Rx(theta) . T(0,0,r)   (utils/xyz.py:55-81)."""
    th, ph = np.radians(theta), np.radians(phi)
    T = np.eye(4)
    T[2, 3] = r
    Rt = np.array([[1, 0, 0, 0],
                   [0, np.cos(th), np.sin(th), 0],
                   [0, -np.sin(th), np.cos(th), 0],
                   [0, 0, 0, 1.]])
    Rp = np.array([[np.cos(ph), np.sin(ph), 0, 0],
                   [-np.sin(ph), np.cos(ph), 0, 0],
                   [0, 0, 1, 0],
                   [0, 0, 0, 1.]])
    return Rp @ Rt @ T


def poses_to_render(r, theta, n_phi=40):
    """n_phi poses on a circle of azimuths linspace(0,360,n_phi) (utils/xyz.py:83-91)."""
    return [torch.from_numpy(spherical_to_pose(r, theta, p)).float()
            for p in np.linspace(0, 360.0, n_phi)]


def camera_rays(pose, cam_params):
    """pose [4,4] float32 tensor -> rays [H*W,6] = [origin, R @ dir]
    (utils/rendering.py:129-134)."""
    d = rays_single_cam(cam_params)
    rd = torch.matmul(pose[:3, :3], d)                      # [3,HW]
    o = pose[:3, 3:].expand(3, d.shape[1])
    return torch.cat((o, rd), dim=0).permute(1, 0).reshape(-1, 6)


def render_image(sd, rays, batch_size, N=128, tn=2, tf=6, u=None):
    """Batched full-image render with the reference's caller semantics
    (utils/rendering.py:139-151): per batch render_nerf, clip rgb to [0,1]
    AFTER compositing, disparity un-clipped.  Unlike the reference the tail
    batch is rendered too.  ``u`` [B,N] optional explicit jitter; otherwise one
    torch.rand(batch,N) per batch in batch order."""
    rgbs, disps = [], []
    with torch.no_grad():
        for s in range(0, rays.shape[0], batch_size):
            r = rays[s:s + batch_size]
            ub = None if u is None else u[s:s + batch_size]
            rgb, disp, _, _, _ = render_nerf(r, sd, N, tn, tf, u=ub)
            rgbs.append(torch.clip(rgb, 0., 1.))
            disps.append(disp)
    return torch.cat(rgbs), torch.cat(disps)


# --------------------------------------------------------------------------
# loss / PSNR / one optimizer step  (reference train.py:16-26, 41-57)
# --------------------------------------------------------------------------
def img_mse(gt, pred):
    return torch.mean((pred - gt) ** 2)


def img_psnr(gt, pred):
    """20 log10(max(gt)) - 10 log10(mse): peak is max(gt), not 1 (train.py:21-26)."""
    ten = torch.tensor(10)
    return 20 * torch.log(torch.max(gt)) / torch.log(ten) \
        - 10 * torch.log(img_mse(gt, pred)) / torch.log(ten)


def train_step_grads(sd, rays, u, gt, N, tn=2, tf=6):
    """loss = MSELoss(rgb, gt) (mean over B*3) and d loss / d every parameter
    for fixed (rays, u, gt)  (train.py:51-54)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rgb, _, _, _, _ = render_nerf(rays, params, N, tn, tf, u=u)
    loss = F.mse_loss(rgb, gt)
    loss.backward()
    return loss.detach(), {k: p.grad.detach() for k, p in params.items()}


def adam_step(sd, grads, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, step=1, state=None):
    """torch.optim.Adam defaults (train.py:43,55), restated explicitly.
    Returns (new_sd, new_state)."""
    b1, b2 = betas
    new_sd, new_state = {}, {}
    for k, p in sd.items():
        g = grads[k]
        m, v = (state[k] if state is not None else (torch.zeros_like(p), torch.zeros_like(p)))
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        mhat = m / (1 - b1 ** step)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)) + eps
        new_sd[k] = p - lr * mhat / denom
        new_state[k] = (m, v)
    return new_sd, new_state


def train_loop(sd, rays_table, gt_table, batch_size, N, num_iters, lr_init, lr_final, seed, decay_iters=None,
               checkpoints=(), on_checkpoint=None):
    """The loop body of train.py:45-57 on a ray table: per iteration ``randperm(n)[:batch_size]`` (RayGenerator.select,
    utils/dataload.py:150-153), render_nerf at N samples (its one torch.rand(B,N) from the same CPU stream), MSELoss,
    backward, torch.optim.Adam(lr=5e-4 hard-coded, train.py:43) step, lr *= decay with
    decay = exp(log(lr_final / lr_init) / decay_iters) (train.py:36-39).  torch's CPU generator is seeded once.
    Returns (losses [num_iters], final params); ``on_checkpoint(i, params)`` is called after iteration i in checkpoints."""
    decay = np.exp(np.log(lr_final / lr_init) / (decay_iters or num_iters))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-4)
    losses = []
    torch.manual_seed(seed)
    for i in range(num_iters):
        ray_ids = torch.randperm(rays_table.size(0))[:batch_size]
        rays, gt = rays_table[ray_ids, :], gt_table[ray_ids, :]
        opt.zero_grad()
        rgb, _, _, _, _ = render_nerf(rays, params, N)
        loss = F.mse_loss(rgb, gt)
        loss.backward()
        opt.step()
        for pg in opt.param_groups:
            pg["lr"] = pg["lr"] * decay
        losses.append(loss.detach())
        if on_checkpoint is not None and i + 1 in checkpoints:
            on_checkpoint(i + 1, params)
    return torch.stack(losses), {k: p.detach() for k, p in params.items()}


# --------------------------------------------------------------------------
# torch's CPU uniform stream (the jitter of reference utils/rendering.py:28-30), restated
# --------------------------------------------------------------------------
def mt19937_uniform(words, nxt, n, raw=False):
    """Continue at::mt19937 (ATen/core/MT19937RNGEngine.h) for n float32 uniforms
    (ATen/core/DistributionsHelper.h: one 32-bit output per draw, u = (y & 0xFFFFFF) * 2^-24).
    words: the 624 state words (uint32), nxt: first unread word of the current block (0..624).
    Returns (u float32 [n], state words afterwards); raw=True: the 32-bit outputs themselves (CPUGeneratorImpl::random(),
    what torch.randperm draws).  Pinned against torch.rand / torch.randperm themselves in
    tests/test_oracle_golden.py; checker for csrc/host_rng.hip."""
    import numpy as np
    N_, M_ = 624, 397
    D_ = N_ - M_
    mt = np.asarray(words, dtype=np.uint32).copy()

    def twist(u_, v_):
        y = (u_ & np.uint32(0x80000000)) | (v_ & np.uint32(0x7fffffff))
        return (y >> np.uint32(1)) ^ np.where(v_ & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0)).astype(np.uint32)

    def temper(y):
        y = y ^ (y >> np.uint32(11))
        y = y ^ ((y << np.uint32(7)) & np.uint32(0x9d2c5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xefc60000))
        return y ^ (y >> np.uint32(18))

    out = np.empty(n, dtype=np.uint32 if raw else np.float32)
    k = 0
    while k < n:
        if nxt >= N_:
            new = mt.copy()
            new[:D_] = mt[M_:] ^ twist(mt[:D_], mt[1:D_ + 1])
            for lo in range(D_, N_ - 1, D_):
                hi = min(lo + D_, N_ - 1)
                new[lo:hi] = new[lo - D_:hi - D_] ^ twist(mt[lo:hi], mt[lo + 1:hi + 1])
            new[N_ - 1] = new[M_ - 1] ^ twist(mt[N_ - 1:N_], new[0:1])[0]
            mt, nxt = new, 0
        take = min(N_ - nxt, n - k)
        y = temper(mt[nxt:nxt + take])
        out[k:k + take] = y if raw else (y & np.uint32(0xffffff)).astype(np.float32) * np.float32(2.0 ** -24)
        k += take
        nxt += take
    return out, mt


# --------------------------------------------------------------------------
# ray selection  (reference utils/dataload.py:141-153 RayGenerator.select, train.py:47-49)
# --------------------------------------------------------------------------
def randperm_prefix(n, B, draws):
    """``torch.randperm(n)[:B]`` from the generator's next 32-bit outputs ``draws`` (the first min(B, n-1) of the n - 1
    the call consumes).  The algorithm lives in torch, not in the reference (third-party: torch 2.10.0
    ATen/native/TensorFactories.cpp randperm_cpu, n < 2^32/20): r = [0..n-1]; for i in 0..n-2: swap(r[i], r[i + random() % (n-i)]).
    Sequential restatement over a sparse table of the touched positions; pinned against torch.randperm itself in
    tests/test_oracle_golden.py (which also pins the reference's call site: G6c / G8 store the ray_ids of reference runs).
    Checker for csrc/select.hip."""
    moved, out = {}, []
    for i in range(min(B, n)):
        j = i + int(draws[i]) % (n - i) if i < n - 1 else i
        vi, vj = moved.get(i, i), moved.get(j, j)
        out.append(vj)
        moved[j], moved[i] = vi, vj
    return np.asarray(out, dtype=np.int64)


def select(rays_table, colours, B, words, nxt):
    """rg.select(mode, N=B) + the ground-truth gather (dataload.py:150-153, train.py:49) from the generator state
    (words, nxt): (rays [B,6], gt [B,3], ray_ids [B] int64)."""
    n = int(rays_table.shape[0])
    first = min(B, max(n - 1, 0))
    draws, _ = mt19937_uniform(words, nxt, first, raw=True)
    ids = torch.from_numpy(randperm_prefix(n, B, draws))
    return rays_table[ids, :], colours[ids, :].float(), ids


def philox_word(seed, ctr):
    """Philox-4x32-10 (Salmon et al. 2011) as csrc/nerf_device.h keys it: counter (ctr lo, ctr hi, 'nerf', 'amd!'), key =
    seed; first output word.  numpy, vectorised over ctr.  The build's own counter RNG: no reference counterpart."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    m32 = np.uint64(0xffffffff)
    c0, c1 = ctr & m32, ctr >> np.uint64(32)
    c2 = np.full_like(c0, 0x6e657266)
    c3 = np.full_like(c0, 0x616d6421)
    k0, k1 = np.uint64(int(seed) & 0xffffffff), np.uint64((int(seed) >> 32) & 0xffffffff)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n2 = (p1 >> np.uint64(32)) ^ c1 ^ k0, (p0 >> np.uint64(32)) ^ c3 ^ k1
        c1, c3, c0, c2 = p1 & m32, p0 & m32, n0, n2
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return c0.astype(np.uint32)


SELECT_KEY = 0x73656c6563743a31


def select_ids_counter(n, B, seed, seed_offset=0):
    """The ids nerf_amd_select_rays draws from the counter RNG: the same forward Fisher-Yates prefix with
    z_i = philox_word((seed ^ SELECT_KEY) + seed_offset, i)."""
    key = ((int(seed) ^ SELECT_KEY) + int(seed_offset)) & 0xffffffffffffffff
    first = min(B, max(n - 1, 0))
    return randperm_prefix(n, B, philox_word(key, np.arange(first, dtype=np.uint64)))
