"""GPU tests of the 8-bit storage form of the training step (``GraphedTrainStep(storage='e4m3')``; csrc/nerf_layout.h,
include/nerf_amd.h "8-bit storage form"; reference train.py:51-54: loss.backward() as far as the parameter gradients).

Three layers of evidence, each through the C ABI:

  1. the producers: the buffers the training forward and the dX chain write with NERF_AMD_STORE_E4M3 decode (by the
     documented layout, on the host) to the bf16 buffers of the default form rounded to e4m3 under the block's exponent (one
     per group of four fragments = 128 features x 32 points) --
     element by element: |x8 - x16| <= half an e4m3 step at x16's magnitude (2^-4 relative; 2^-10 of the block scale in
     the subnormal range), the exponent puts the block's largest magnitude in [128, 256], and everything the chain itself
     produces (raw, ts, the ReLU masks) is bit for bit that of the bf16 form;
  2. the consumer: the 14 products and the bias sums from those very buffers equal the float64 products of the DECODED
     operands to the accuracy of the instruction's inner sum (2^-11 of the tensor's largest entry for a handful of points; ~1e-5 over
     thousands) -- the kernel adds no error of its own to what the storage form costs;
  3. the requirement: at the reference's real step shape (fixture G6c, 4096 x 128) every gradient tensor stays inside
     GRAD_NOISE_RATIO of the reference's own minibatch deviation, and the 60-iteration trajectory inside the same bands as
     the bf16 form (tests/test_gpu_trajectory.py, modes with ``storage='e4m3'``).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from nerf_simple_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


E4M3 = np.array([np.nan if (v & 0x7f) == 0x7f else
                 (-1.0 if v & 0x80 else 1.0) * ((v & 7) / 8.0 * 2.0 ** -6 if (v >> 3) & 15 == 0 else (1 + (v & 7) / 8.0) * 2.0 ** (((v >> 3) & 15) - 7))
                 for v in range(256)])


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def decode_bf16_layers(host, P):
    """[10][P, 256] float32 from the point-blocked bf16 buffer (nerf_layout.h act_elem_offset)."""
    nt = (P + 255) // 256
    out = []
    for L in range(10):
        blk = host[L * nt * 131072:(L + 1) * nt * 131072].view(np.uint16).reshape(nt, 32, 256, 8)
        out.append(bf16_to_f32(blk.transpose(0, 2, 1, 3).reshape(nt * 256, 256)[:P]))
    return out


def decode_e4m3_layers(host, P):
    """([10][P, 256] float64 values, [10][P, 256] bytes, [10][blocks, 8] exponent bytes) from the 8-bit buffer
    (nerf_layout.h f8_elem_offset / f8_scale_offset_bytes)."""
    nt = (P + 255) // 256
    vals, raws, exps = [], [], []
    scale0 = 10 * nt * 65536
    for L in range(10):
        blk = host[L * nt * 65536:(L + 1) * nt * 65536].reshape(nt, 16, 256, 16)
        b = blk.transpose(0, 2, 1, 3).reshape(nt * 256, 256)[:P]
        e = host[scale0 + L * nt * 64: scale0 + (L + 1) * nt * 64].reshape(nt * 8, 8)
        sc = 2.0 ** (e.astype(np.float64) - 127.0)                                    # [block, Q]
        per_elem = np.repeat(np.repeat(sc, 32, axis=0)[:P], 32, axis=1)               # [P, 256]
        vals.append(E4M3[b] * per_elem)
        raws.append(b)
        exps.append(e)
    return vals, raws, exps


def run_chain(dev, synthetic, B, N, e4m3, kind="default", seed=4, buckets=False):
    """Forward (saving), compositor backward, dX chain, dW through the C ABI in either storage form; returns the host
    copies of everything."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    lib = _lib.lib()
    net = Nerf().to(dev)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.synthetic_state_dict(5, kind).items()})
    side = int(np.ceil(np.sqrt(B)))
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [side, side, synthetic.focal_from_fov(side)]).float().contiguous()[:B].contiguous().to(dev)
    gen = torch.Generator().manual_seed(seed)
    u = torch.rand(B, N, generator=gen).to(dev)
    gt = torch.rand(B, 3, generator=gen).to(dev)
    P = B * N
    tbins = torch.linspace(2, 6, N + 1).to(dev)
    raw, ts = torch.empty(B, N, 4, device=dev), torch.empty(B, N, device=dev)
    rgb, d_raw = torch.empty(B, 3, device=dev), torch.empty(B, N, 4, device=dev)
    posx = torch.empty(P, 64, dtype=torch.bfloat16, device=dev)
    posd = torch.empty(P, 32, dtype=torch.bfloat16, device=dev)
    grads = torch.zeros(int(lib.nerf_amd_param_count()), device=dev)
    scratch = torch.empty(max(int(lib.nerf_amd_param_gradients_scratch_bytes(P)), 16), dtype=torch.uint8, device=dev)
    st, ptr, ck = _lib.stream_ptr(dev), _lib.ptr, _lib.check
    if e4m3:
        acts = torch.zeros(int(lib.nerf_amd_train_activation_bytes_e4m3(P)), dtype=torch.uint8, device=dev)
        dys = torch.zeros(int(lib.nerf_amd_train_gradient_bytes_e4m3(P)), dtype=torch.uint8, device=dev)
        scratch8 = torch.zeros(int(lib.nerf_amd_param_gradients_scratch_e4m3_bytes(P)), dtype=torch.uint8, device=dev)
    else:
        acts = torch.zeros(int(lib.nerf_amd_train_activation_bytes(P)), dtype=torch.uint8, device=dev)
        dys = torch.zeros_like(acts)
    packed, image = net.packed_weights(_lib.BF16), net.packed_weights(_lib.BF16_BWD)
    ck(lib.nerf_amd_sample_encode_bf16(ptr(rays), ptr(u), ptr(tbins), 0, 0, 0, ptr(posx), ptr(posd), None, B, N, st), "encode")
    ck(lib.nerf_amd_mlp_forward_train(ptr(rays), ptr(u), ptr(tbins), ptr(packed), _lib.FLAG_STORE_E4M3 if e4m3 else 0, 0, 0,
                                      ptr(raw), ptr(ts), ptr(acts), B, N, st), "forward")
    ck(lib.nerf_amd_volume_render_mse_backward(ptr(raw), ptr(ts), ptr(rays), ptr(gt), ptr(rgb), ptr(d_raw), B, N, st), "composite")
    ck(lib.nerf_amd_param_gradients_begin(ptr(d_raw), ptr(scratch), ptr(grads), P, st), "begin")
    if e4m3:
        ck(lib.nerf_amd_mlp_backward_e4m3(ptr(d_raw), ptr(image), ptr(acts), ptr(dys), P, st), "backward")
        ck(lib.nerf_amd_param_gradients_convert_e4m3(ptr(posx), ptr(posd), ptr(scratch), ptr(scratch8), P, 3, st), "convert")
        for bucket in ((1, 2) if buckets else (0,)):              # the data-parallel step's two launches, or everything at once
            ck(lib.nerf_amd_param_gradients_finish_e4m3(ptr(acts), ptr(dys), ptr(scratch8), ptr(grads), P, bucket, st), "finish")
    else:
        ck(lib.nerf_amd_mlp_backward(ptr(d_raw), ptr(image), ptr(acts), ptr(dys), P, st), "backward")
        ck(lib.nerf_amd_param_gradients_finish(ptr(acts), ptr(dys), ptr(posx), ptr(posd), ptr(scratch), ptr(grads), P, st), "finish")
    torch.cuda.synchronize()
    out = dict(raw=raw.cpu().numpy(), ts=ts.cpu().numpy(), d_raw=d_raw.cpu().numpy().reshape(P, 4), acts=acts.cpu().numpy(),
               dys=dys.cpu().numpy(), grads=grads.cpu().numpy(), posx=posx.float().cpu().numpy(), posd=posd.float().cpu().numpy(),
               scratch=scratch.cpu().numpy(), P=P)
    if e4m3:
        out["scratch8"] = scratch8.cpu().numpy()
    return out


def check_rounding(v8, raw8, exps, v16, width, tag):
    """Every stored element is its bf16 value rounded to e4m3 under the block's exponent; the exponent is the rule's."""
    P = v16.shape[0]
    v16 = v16[:, :width].astype(np.float64)
    v8, raw8 = v8[:, :width], raw8[:, :width]
    assert not np.isnan(v8).any(), tag
    nq = width // 32
    sc = 2.0 ** (exps[:, :nq].astype(np.float64) - 127.0)
    per = np.repeat(np.repeat(sc, 32, axis=0)[:P], 32, axis=1)
    scaled = np.abs(v16) / per
    # half a step of e4m3 at the scaled magnitude: 2^-4 relative for normals (>= 2^-6), 2^-10 absolute below
    tol = np.maximum(scaled * 2.0 ** -4, 2.0 ** -10) * per
    err = np.abs(v8 - v16)
    assert (err <= tol * (1 + 1e-12)).all(), (tag, float((err / tol).max()))
    assert (np.sign(v8) * np.sign(v16) >= 0).all(), tag
    # the exponent rule: ONE exponent per group of four fragments (128 features x 32 points: the producers pay the
    # cross-lane step once per group), under which the group's largest |bf16| lands in [128, 256] (groups of zeros / of
    # tiny values: byte 1)
    pad = (-P) % 32
    a = np.pad(np.abs(v16), ((0, pad), (0, 0))).reshape(-1, 32, nq, 32).max(axis=(1, 3))       # [block, Q]
    e = exps[:a.shape[0], :nq].astype(np.int64)
    ng = nq // 4
    eg = e.reshape(-1, ng, 4)
    assert (eg == eg[:, :, :1]).all(), tag                                                      # equal inside a group
    ag = a.reshape(-1, ng, 4).max(axis=2)
    lead = ag / 2.0 ** (eg[:, :, 0] - 127.0)
    # (the exponent comes from the fp32 values before their bf16 rounding, which can carry the largest one up to the next
    # power of two: exactly 256 then)
    assert (lead <= 256).all(), (tag, lead.max())
    # (a ragged last block: the wave's lanes past the end compute on the last point and take part in the maximum)
    full = np.zeros(ag.shape, dtype=bool)
    full[:P // 32] = True
    big = full & (ag >= 2.0 ** -118)
    assert (lead[big] >= 128).all(), (tag, lead[big].min())
    assert (eg[:, :, 0][full & ~big] >= 1).all(), tag
    return float((err / np.maximum(np.abs(v16), 1e-30))[np.abs(v16) > 0].mean())


@pytest.mark.parametrize("B,N", [(25, 24), (64, 64), (3, 7), (33, 9)])
def test_e4m3_buffers_are_the_bf16_ones_rounded(dev, synthetic, B, N):
    """Layer 1 of the evidence (module docstring): 600 points (two tiles and a ragged third) and 4096 points."""
    a16 = run_chain(dev, synthetic, B, N, False)
    a8 = run_chain(dev, synthetic, B, N, True)
    P = a16["P"]
    for k in ("raw", "ts", "d_raw", "posx", "posd"):
        assert np.array_equal(a16[k], a8[k]), k                     # the chain itself is untouched
    nt = (P + 255) // 256
    m16 = a16["acts"][10 * nt * 131072:]
    region8 = (10 * nt * 65536 + 10 * nt * 64 + 255) // 256 * 256
    m8 = a8["acts"][region8:]
    assert m16.size == m8.size == 10 * nt * 8192 and np.array_equal(m16, m8)       # the ReLU masks
    for name in ("acts", "dys"):
        ref = decode_bf16_layers(a16[name], P)
        vals, raws, exps = decode_e4m3_layers(a8[name], P)
        for L in range(10):
            width = 128 if L == 9 else 256
            mean_rel = check_rounding(vals[L], raws[L], exps[L], ref[L], width, (name, L))
            assert mean_rel < 2.0 ** -5, (name, L, mean_rel)
    # the encoder rows and the packed d_raw (nerf_amd_param_gradients_convert_e4m3)
    from_rows = {64: a16["posx"], 32: a16["posd"],
                 16: bf16_to_f32(a16["scratch"][:P * 64].view(np.uint16).reshape(P, 32))[:, :16]}
    off = 0
    for W in (64, 32, 16):
        data_bytes = nt * (W // 16) * 4096
        total = data_bytes + nt * 64
        buf = a8["scratch8"][off:off + total]
        blk = buf[:data_bytes].reshape(nt, W // 16, 256, 16).transpose(0, 2, 1, 3).reshape(nt * 256, W)[:P]
        e = buf[data_bytes:].reshape(nt * 8, 8)
        nq = (W + 31) // 32
        sc = 2.0 ** (e[:, :nq].astype(np.float64) - 127.0)
        per = np.repeat(np.repeat(sc, 32, axis=0)[:P], 32, axis=1)[:, :W]
        v8 = E4M3[blk] * per
        want = from_rows[W].astype(np.float64)
        tol = np.maximum(np.abs(want) / per * 2.0 ** -4, 2.0 ** -10) * per
        assert (np.abs(v8 - want) <= tol * (1 + 1e-12)).all(), W
        off += (total + 255) // 256 * 256


def expected_grads(a8, layout):
    """The float64 products of the decoded operands, in the flat vector's order."""
    P = a8["P"]
    nt = (P + 255) // 256
    X, _, _ = decode_e4m3_layers(a8["acts"], P)
    dY, _, _ = decode_e4m3_layers(a8["dys"], P)
    off = 0
    narrow = {}
    for W in (64, 32, 16):
        data_bytes = nt * (W // 16) * 4096
        total = data_bytes + nt * 64
        buf = a8["scratch8"][off:off + total]
        blk = buf[:data_bytes].reshape(nt, W // 16, 256, 16).transpose(0, 2, 1, 3).reshape(nt * 256, W)[:P]
        e = buf[data_bytes:].reshape(nt * 8, 8)
        nq = (W + 31) // 32
        per = np.repeat(np.repeat(2.0 ** (e[:, :nq].astype(np.float64) - 127.0), 32, axis=0)[:P], 32, axis=1)[:, :W]
        narrow[W] = E4M3[blk] * per
        off += (total + 255) // 256 * 256
    posx, posd, dsr = narrow[64][:, :63], narrow[32][:, :27], narrow[16]
    d_raw = a8["d_raw"].astype(np.float64)
    g = {}
    g["layers_0.0.weight"] = dY[0].T @ posx
    g["layers_0.0.bias"] = dY[0].sum(0)
    for i, l in enumerate((2, 4, 6, 8), start=1):
        g[f"layers_0.{l}.weight"] = dY[i].T @ X[i - 1]
        g[f"layers_0.{l}.bias"] = dY[i].sum(0)
    g["skip_conn_layer.0.weight"] = np.concatenate([dY[5].T @ X[4], dY[5].T @ posx], axis=1)
    g["skip_conn_layer.0.bias"] = dY[5].sum(0)
    g["layers_1.0.weight"] = dY[6].T @ X[5]
    g["layers_1.0.bias"] = dY[6].sum(0)
    g["layers_1.2.weight"] = dY[7].T @ X[6]
    g["layers_1.2.bias"] = dY[7].sum(0)
    g["sigma_fc.0.weight"] = dsr[:, 3:4].T @ X[7]
    g["sigma_fc.0.bias"] = d_raw[:, 3].sum(keepdims=True)              # head biases come from the fp32 d_raw
    g["layers_2.weight"] = dY[8].T @ X[7]
    g["layers_2.bias"] = dY[8].sum(0)
    g["color_fc.0.weight"] = np.concatenate([dY[9][:, :128].T @ X[8], dY[9][:, :128].T @ posd], axis=1)
    g["color_fc.0.bias"] = dY[9][:, :128].sum(0)
    g["color_fc.2.weight"] = dsr[:, :3].T @ X[9][:, :128]
    g["color_fc.2.bias"] = d_raw[:, :3].sum(0)
    return g


@pytest.mark.parametrize("B,N,kind", [(25, 24, "default"), (64, 64, "structured"), (301, 8, "default"), (3, 7, "default"),
                                      (257, 1, "structured"), (1, 64, "default")])
def test_e4m3_products_add_nothing_of_their_own(dev, synthetic, B, N, kind):
    """Layer 2: the kernel's gradients against float64 products of the decoded buffers (ragged sizes included)."""
    from nerf_simple_amd.utils.nets import Nerf
    a8 = run_chain(dev, synthetic, B, N, True, kind)
    want = expected_grads(a8, None)
    names = [k for k, _ in Nerf().named_parameters()]
    off = 0
    for k, p in Nerf().named_parameters():
        n = p.numel()
        got = a8["grads"][off:off + n].reshape(p.shape).astype(np.float64)
        off += n
        w = want[k].reshape(p.shape)
        scale = max(np.abs(w).max(), 1e-30)
        # fp32 accumulation + float atomics of the split-K partials, and the instruction's own inner sum: products 2^-14
        # below the largest of their group of eight are dropped (tools/micro/f8_probe.hip): up to seven of them, 7 x 2^-14 <
        # 2^-11 of the largest product -- that bounds it for a handful of points (observed up to 1.6e-4 at P = 16 on the
        # high-gain weights); over thousands of points the error is ~1e-5
        assert np.abs(got - w).max() <= 2.0 ** -11 * scale + 1e-12, (k, float(np.abs(got - w).max() / scale))
    assert off == a8["grads"].size and len(names) == 24


def test_e4m3_buckets_are_the_whole(dev, synthetic):
    """The two launches of the data-parallel step (bucket 1 = late layers, bucket 2 = layers_0.*) give the gradients of the one
    launch: the same products, each workgroup share re-balanced per launch, so float atomics in another order."""
    a, b = run_chain(dev, synthetic, 64, 64, True), run_chain(dev, synthetic, 64, 64, True, buckets=True)
    assert np.array_equal(a["acts"], b["acts"]) and np.array_equal(a["dys"], b["dys"])
    scale = np.abs(a["grads"]).max()
    assert np.abs(a["grads"] - b["grads"]).max() <= 2e-5 * scale
    assert np.count_nonzero(b["grads"]) > 0.5 * b["grads"].size        # (dead units have exactly zero rows)


def test_e4m3_against_bf16_gradients(dev, synthetic):
    """The two storage forms side by side on 4096 points: every tensor's gradient within 2^-5 relative L2 of the bf16
    form's (each operand element is within 2^-4; the errors of a sum over the points are far smaller -- the REQUIREMENT
    is layer 3, in tests/test_gpu_trajectory.py)."""
    from nerf_simple_amd.utils.nets import Nerf
    a16 = run_chain(dev, synthetic, 64, 64, False)
    a8 = run_chain(dev, synthetic, 64, 64, True)
    off = 0
    for k, p in Nerf().named_parameters():
        n = p.numel()
        g16, g8 = a16["grads"][off:off + n].astype(np.float64), a8["grads"][off:off + n].astype(np.float64)
        off += n
        rel = np.linalg.norm(g8 - g16) / max(np.linalg.norm(g16), 1e-30)
        assert rel <= 2.0 ** -5, (k, rel)


def test_graphed_step_takes_the_storage_keyword(dev, synthetic):
    """GraphedTrainStep(storage='e4m3'): the same loss as the default to the last bit (the forward is untouched), gradients
    close to it, half-size buffers; an unknown storage is refused."""
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.optim import FusedAdam
    from nerf_simple_amd.training import GraphedTrainStep
    gen = torch.Generator().manual_seed(3)
    B, N = 128, 64
    from nerf_simple_amd.utils.xyz import camera_rays, spherical_to_pose
    pose = torch.from_numpy(spherical_to_pose(4, -30, 0)).float()
    rays = camera_rays([pose], [16, 16, synthetic.focal_from_fov(16)]).float()[:B].contiguous().to(dev)
    gt, u = torch.rand(B, 3, generator=gen).to(dev), torch.rand(B, N, generator=gen).to(dev)
    res = {}
    for storage in ("bf16", "e4m3"):
        net = Nerf(precision="bf16").to(dev)
        net.load_state_dict(synthetic.synthetic_state_dict(0, "default"))
        stepper = GraphedTrainStep(net, FusedAdam(net, lr=5e-4), B, N, storage=storage)
        losses = [float(stepper.step(rays, gt, u=u)) for _ in range(3)]
        res[storage] = (losses, stepper.grads.clone(), stepper.acts.numel() + stepper.dys.numel())
    assert res["bf16"][0][0] == res["e4m3"][0][0]                 # step 1: identical weights, identical forward
    assert abs(res["bf16"][0][2] - res["e4m3"][0][2]) <= 2e-3 * abs(res["bf16"][0][2])
    assert res["e4m3"][2] < 0.56 * res["bf16"][2]
    rel = float((res["e4m3"][1] - res["bf16"][1]).norm() / res["bf16"][1].norm())
    assert rel < 5e-2, rel
    with pytest.raises(ValueError):
        GraphedTrainStep(net, FusedAdam(net, lr=5e-4), B, N, storage="fp4")
