"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every
symbol include/nerf_amd.h declares, and its host-side layout math is sane.
No compute entry point is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nerf_amd.h")


@pytest.fixture(scope="module")
def lib():
    from nerf_simple_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_amd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from nerf_simple_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 13
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/nerf_amd.h but not exported"
    # the Python binding covers exactly the header
    assert sorted(_lib.EXPORTS) == syms


def test_introspection(lib):
    assert lib.nerf_amd_abi_version() == 5
    assert lib.nerf_amd_param_count() == 595844
    # 16-bit images: weights, bias table, 256-byte status block (the fp16 range guard's sticky flags)
    assert lib.nerf_amd_packed_bytes(1) == lib.nerf_amd_packed_bytes(2) == 1172 * 1024 + 2464 * 4 + 256
    assert lib.nerf_amd_packed_status_offset(1) == lib.nerf_amd_packed_status_offset(2) == 1172 * 1024 + 2464 * 4
    assert lib.nerf_amd_packed_status_offset(0) == -1 and lib.nerf_amd_packed_status_offset(3) == -1
    first, count = ctypes.c_int64(), ctypes.c_int64()
    ranges = []
    for bucket in (0, 1, 2):
        assert lib.nerf_amd_grad_bucket_range(bucket, ctypes.byref(first), ctypes.byref(count)) == 0
        ranges.append((first.value, count.value))
    assert ranges[0] == (0, 595844)
    # bucket 2 = layers_0.* (the head of the flat vector), bucket 1 = everything behind it
    head = 256 * 63 + 256 + 4 * (256 * 256 + 256)
    assert ranges[2] == (0, head) and ranges[1] == (head, 595844 - head)
    assert lib.nerf_amd_grad_bucket_range(3, ctypes.byref(first), ctypes.byref(count)) < 0
    assert lib.nerf_amd_packed_bytes(0) == 2360 * 1024 + 154 * 16 * 4
    assert lib.nerf_amd_packed_bytes(7) < 0
    for prec in (0, 1, 2):                                                              # one fused launch: no workspace
        assert lib.nerf_amd_render_workspace_bytes(prec, 16000, 128) == 0
    assert lib.nerf_amd_render_workspace_bytes(2, 16000, 768) == 0
    assert lib.nerf_amd_render_workspace_bytes(1, 100, 769) >= 100 * 769 * 20           # ray longer than the LDS ring
    assert lib.nerf_amd_render_workspace_bytes(1, -1, 128) < 0
    assert lib.nerf_amd_render_image_workspace_bytes(1, 1000, 128) == 24064             # the ray table only


def test_layout_selfcheck(lib):
    assert lib.nerf_amd_layout_selfcheck() == 0


def test_layout_maps(lib):
    """Spot-check the k-permutations against their definitions (csrc/nerf_layout.h)."""
    # 16-bit chain order (bf16 and fp16 share it): element j of lane group g in k-step q is feature
    # 32q + 16(j>>2) + 4g + (j&3)  (rows of two stacked 16-row accumulator tiles of mfma 16x16x32)
    for prec in (1, 2):
        for q in range(8):
            for g in range(4):
                for j in range(8):
                    assert lib.nerf_amd_layout_src_col(prec, 1, q, g, j) == 32 * q + 16 * (j >> 2) + 4 * g + (j & 3)
    # skip layer: chain part then posx slots offset by 256 ([h ; x], h first)
    cols = sorted(lib.nerf_amd_layout_src_col(1, 5, s, g, j)
                  for s in range(8, 10) for g in range(4) for j in range(8))
    assert cols == [-1] + list(range(256, 319))
    # layer 0 covers the 63 posx columns exactly once
    cols = sorted(lib.nerf_amd_layout_src_col(1, 0, s, g, j)
                  for s in range(2) for g in range(4) for j in range(8))
    assert cols == [-1] + list(range(63))
    assert lib.nerf_amd_layout_src_col(1, 1, 8, 0, 0) == -2 and lib.nerf_amd_layout_src_col(1, 1, 0, 4, 0) == -2
    # f32 chain order: register i of tile t in lane group g is feature 16t + 4g + i
    for s in range(64):
        for g in range(4):
            assert lib.nerf_amd_layout_src_col(0, 2, s, g, 0) == 16 * (s >> 2) + 4 * g + (s & 3)
    cols = sorted(lib.nerf_amd_layout_src_col(0, 9, s, g, 0) for s in range(64, 72) for g in range(4))
    assert cols == [-1] * 5 + list(range(256, 283))


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of computing on the host."""
    import torch
    from nerf_simple_amd.utils import rendering, xyz, nets
    with pytest.raises(RuntimeError):
        xyz.positional_encoder(torch.zeros(4, 6))
    with pytest.raises(RuntimeError):
        rendering.volume_render(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    net = nets.Nerf()                       # (its nn.Linear initialisers draw from the CPU generator)
    state = torch.get_rng_state()
    with pytest.raises(RuntimeError):
        rendering.render_nerf(torch.zeros(2, 6), net, 8)
    assert torch.equal(torch.get_rng_state(), state), "a refused call must not consume the CPU generator"
    with pytest.raises(AssertionError):
        xyz.gamma([1.0, 2.0])
    # training precision contract, checked without a GPU: the fused (graphed) step is bf16; fp32 trains layer by layer
    from nerf_simple_amd import training
    with pytest.raises(RuntimeError, match="the fused training step is bf16"):
        training._check_fused_trainable("fp32")
    training._check_fused_trainable("fp16")
    training._check_fused_trainable("bf16")


def test_state_dict_contract(synthetic):
    """24 keys / shapes of the reference checkpoint format."""
    from nerf_simple_amd.utils.nets import Nerf
    net = Nerf()
    sd = net.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, s) for k, s in synthetic.PARAM_SPECS]
    net.load_state_dict(synthetic.synthetic_state_dict(0, "structured"), strict=True)
    assert sum(p.numel() for p in net.parameters()) == 595844


def test_network_sizes_follow_the_reference_constructor(oracle):
    """Nerf(Lp, Ld, H) takes any sizes, as the reference's does (utils/nets.py:9-32): the same 24 state-dict keys with
    the reference's shapes, loadable into the oracle's forward.  Sizes other than the default have no packed weight
    image (they run on the layer-by-layer path, GPU tests) and say so; nonsense sizes raise at construction."""
    import torch
    from nerf_simple_amd.utils.nets import Nerf
    assert Nerf(10, 4, 256)._fused_ok()
    for Lp, Ld, H in ((6, 4, 256), (10, 2, 256), (10, 4, 128), (1, 1, 2), (3, 1, 40)):
        net = Nerf(Lp, Ld, H)
        assert not net._fused_ok()
        sd = net.state_dict()
        cx, cd = 3 + 6 * Lp, 3 + 6 * Ld
        assert len(sd) == 24
        assert sd["layers_0.0.weight"].shape == (H, cx) and sd["skip_conn_layer.0.weight"].shape == (H, H + cx)
        assert sd["color_fc.0.weight"].shape == (H // 2, H + cd) and sd["color_fc.2.weight"].shape == (3, H // 2)
        assert sd["sigma_fc.0.weight"].shape == (1, H)
        with torch.no_grad():
            out = oracle.nerf_forward(sd, torch.randn(5, 6), Lp, Ld)
        assert out.shape == (5, 4)
        with pytest.raises(RuntimeError, match="GPU|no packed weight image"):
            net.packed_weights()
    for bad in ((0, 4, 256), (10, 0, 256), (10, 4, 1), (10.5, 4, 256)):
        with pytest.raises(RuntimeError, match="sizes must be integers"):
            Nerf(*bad)


def test_oracle_single_sample_is_the_reference_degenerate_case(oracle, synthetic):
    """N = 1 in the reference (verified by running it: utils/rendering.py:60-61 builds deltas from an empty
    difference): empty sample axis, rgb = acc = 0, disparity NaN.  The oracle issues the same torch ops; the GPU
    tests hold the kernels to this."""
    import torch
    rays = torch.tensor([[0., 0, 4, 0.1, 0.2, -1.0], [0., 0, 4, -0.1, 0.0, -1.0]])
    with torch.no_grad():
        rgb, disp, alpha, acc, w = oracle.render_nerf(rays, synthetic.synthetic_state_dict(0, "default"), 1)
    assert alpha.shape == (2, 0) and w.shape == (2, 0)
    assert float(rgb.abs().max()) == 0 and float(acc.abs().max()) == 0 and torch.isnan(disp).all()


def test_module_copies_leave_the_packed_cache_behind(synthetic):
    """copy.deepcopy / pickle of a Nerf carry the 24 parameters and the precision, not the derived weight images."""
    import copy
    import pickle
    import torch
    from nerf_simple_amd.utils.nets import Nerf, _Packed
    a = Nerf(precision="bf16")
    a.load_state_dict(synthetic.synthetic_state_dict(2, "default"))
    a._packed[("fake", 1)] = _Packed((), torch.zeros(4, dtype=torch.uint8))      # stands in for a device image
    for b in (copy.deepcopy(a), pickle.loads(pickle.dumps(a))):
        assert b._packed == {} and b.precision == "bf16"
        assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    assert len(a._packed) == 1


def test_host_camera_helpers(golden):
    import torch
    import numpy as np
    from nerf_simple_amd.utils import xyz
    g = golden("camera.npz")
    assert np.array_equal(xyz.rays_single_cam([100, 100, float(g["f"])]).numpy(), g["dirs100"])
    assert np.array_equal(xyz.rays_single_cam([6, 10, 7.5]).numpy(), g["dirs_6x10"])
    assert np.array_equal(xyz.spherical_to_pose(4, -30, 40), g["pose_4_m30_40"])
    assert np.array_equal(torch.stack(xyz.poses_to_render(4, -30, 5)).numpy(), g["poses5"])
    pose = torch.from_numpy(xyz.spherical_to_pose(4, -30, 40)).float()
    assert np.array_equal(xyz.camera_rays([pose], [100, 100, float(g["f"])]).numpy(), g["rays100_phi40"])


def test_abi_argument_errors(lib):
    """Error behaviour of the C ABI (include/nerf_amd.h): bad arguments return
    NERF_AMD_EINVAL / EUNSUP before anything is launched -- no exception, no exit, and
    no GPU needed to observe it."""
    EINVAL, EUNSUP = -1, -2
    null = None
    one = ctypes.c_void_p(16)                     # a non-null dummy: never dereferenced on these paths
    assert lib.nerf_amd_pack_weights(null, one, 1, null) == EINVAL
    assert lib.nerf_amd_pack_weights(one, one, 9, null) == EINVAL
    assert lib.nerf_amd_gamma(one, 1, one, -1, 4, null) == EINVAL
    assert lib.nerf_amd_gamma(null, 1, null, 0, 4, null) == 0          # empty input: nothing to do
    assert lib.nerf_amd_positional_encoder(null, one, one, 5, 10, 4, null) == EINVAL
    assert lib.nerf_amd_mlp_forward(one, one, one, 8, 7, null) == EINVAL          # unknown precision
    assert lib.nerf_amd_mlp_forward(null, one, one, 8, 1, null) == EINVAL
    assert lib.nerf_amd_mlp_forward(null, null, null, 0, 1, null) == 0
    assert lib.nerf_amd_volume_render(one, one, one, 2, one, one, null, one, null, 4, 8, null) == EINVAL  # stride < 3
    assert lib.nerf_amd_volume_render(one, one, one, 3, one, one, null, one, null, 4, 0, null) == EINVAL  # N = 0
    assert lib.nerf_amd_volume_render_backward(one, one, one, 3, one, null, null, null, null, one, 4, 1024, null) == EUNSUP
    assert lib.nerf_amd_render_forward(one, null, one, one, 1, 0, 0, 0, one, one, null, one, null, one, 4, 8, null) == EINVAL  # no jitter
    assert lib.nerf_amd_render_forward(one, one, null, one, 1, 0, 0, 0, one, one, null, one, null, one, 4, 8, null) == EINVAL  # no tbins
    assert lib.nerf_amd_render_forward(one, one, one, one, 0, 0, 0, 0, one, one, null, one, null, null, 4, 800, null) == EINVAL  # no workspace on the two-launch path (N > 768)
    assert lib.nerf_amd_sample_pdf(one, one, one, 0, 0, 0, one, 4, 2, 8, null) == EUNSUP      # Nc < 3
    assert lib.nerf_amd_sample_pdf(one, one, one, 0, 0, 0, one, 4, 300, 8, null) == EUNSUP    # Nc > 256
    assert lib.nerf_amd_generate_rays(one, 10, 10, ctypes.c_float(5.0), 90, 20, one, null) == EINVAL   # past the image
    assert lib.nerf_amd_generate_rays(one, 10, 10, ctypes.c_float(0.0), 0, 10, one, null) == EINVAL    # f <= 0
    assert lib.nerf_amd_mlp_backward(null, one, one, one, 16, null) == EINVAL
    assert lib.nerf_amd_param_gradients(one, one, one, one, one, one, null, 16, null) == EINVAL
    # point-blocked bf16 activations (10 layers x ceil(P/256) tiles x 128 KiB) + ReLU mask bits (.. x 8 KiB)
    assert lib.nerf_amd_train_activation_bytes(1000) == 10 * 4 * (131072 + 8192)
    assert lib.nerf_amd_packed_bytes(3) == 1112 * 1024
    assert lib.nerf_amd_packed_bytes(2) == 1172 * 1024 + 2464 * 4 + 256
    assert lib.nerf_amd_query_points(one, null, one, 0, 0, 0, one, null, 4, 8, null) == EINVAL        # no jitter
    assert lib.nerf_amd_query_points(null, null, null, 0, 0, 0, null, null, 0, 8, null) == 0           # no rays: nothing to do
    assert lib.nerf_amd_param_gradients_finish_bucket(one, one, one, one, one, one, 16, 3, null) == EINVAL   # buckets are 0, 1, 2
    # flags: unknown bits, and the seed-in-memory form (its address must be 8-byte aligned; the hierarchical entry has no such form)
    assert lib.nerf_amd_render_forward(one, one, one, one, 1, 8, 0, 0, one, one, null, one, null, one, 4, 8, null) == EINVAL
    assert lib.nerf_amd_render_forward(one, ctypes.c_void_p(20), one, one, 1, 6, 0, 0, one, one, null, one, null, one, 4, 8, null) == EINVAL
    assert lib.nerf_amd_render_hierarchical_forward(one, 8, 8, ctypes.c_float(5.0), 0, 64, null, null, one, one, one, 1, 6, 0,
                                                    one, one, 16, 16, null) == EINVAL
    assert lib.nerf_amd_render_hierarchical_forward(one, 8, 8, ctypes.c_float(5.0), 0, 64, one, one, one, one, one, 1, 8, 0,
                                                    one, one, 16, 16, null) == EINVAL
    assert lib.nerf_amd_sample_pdf(one, one, one, 4, 0, 0, one, 4, 16, 8, null) == EINVAL
    # ray selection (RayGenerator.select): B <= n < 2^32 / 20, aligned tables
    assert lib.nerf_amd_select_rays(null, 0, null, 10, 11, one, one, one, one, one, one, null) == EINVAL       # B > n
    assert lib.nerf_amd_select_rays(null, 0, null, 2 ** 32 // 20, 16, one, one, one, one, one, one, null) == EUNSUP
    assert lib.nerf_amd_select_rays(null, 0, null, 10, 0, null, null, null, null, null, null, null) == 0       # nothing to select
    assert lib.nerf_amd_select_rays(null, 0, null, 100, 16, one, one, one, one, one, null, null) == EINVAL     # no workspace
    assert lib.nerf_amd_select_rays(null, 0, null, 100, 16, ctypes.c_void_p(20), one, one, one, one, one, null) == EINVAL   # rows are read 8 bytes at a time
    assert lib.nerf_amd_select_rays(null, 0, null, 100, 16, null, one, one, one, one, one, null) == EINVAL     # rays out without a table
    assert lib.nerf_amd_select_workspace_bytes(4096) == 49408
    assert lib.nerf_amd_mt19937_raw(null, 0, one, 4, null, null) == EINVAL
    assert lib.nerf_amd_mt19937_advance(one, one, one, null) == EINVAL                                          # in place
    assert lib.nerf_amd_mt19937_jump_poly(-1, one, one) == EINVAL
    # the 8-bit storage form of the training step: sizes (10 layers x tiles x 64 KiB + 64 exponent bytes per layer and tile,
    # acts: + the mask planes), flag accepted by the training forward only, conversions name what they convert
    assert lib.nerf_amd_train_gradient_bytes_e4m3(1000) == 10 * 4 * (65536 + 64)
    assert lib.nerf_amd_train_activation_bytes_e4m3(1000) == 10 * 4 * (65536 + 64) + 10 * 4 * 8192
    assert lib.nerf_amd_train_activation_bytes_e4m3(-1) == EINVAL and lib.nerf_amd_param_gradients_scratch_e4m3_bytes(-1) == EINVAL
    assert lib.nerf_amd_param_gradients_scratch_e4m3_bytes(1000) == 4 * (4 + 2 + 1) * 4096 + 3 * 4 * 64
    assert lib.nerf_amd_render_forward(one, one, one, one, 1, 8 | 0, 0, 0, one, one, null, one, null, one, 4, 8, null) == EINVAL   # STORE_E4M3 elsewhere
    assert lib.nerf_amd_mlp_forward_train(one, one, one, one, 8 | 16, 0, 0, one, one, one, 4, 8, null) == EINVAL                 # unknown bit beside it
    assert lib.nerf_amd_mlp_backward_e4m3(one, one, null, one, 16, null) == EINVAL
    assert lib.nerf_amd_param_gradients_convert_e4m3(one, one, one, one, 16, 0, null) == EINVAL                                 # nothing named
    assert lib.nerf_amd_param_gradients_convert_e4m3(null, one, one, one, 16, 1, null) == EINVAL                                # encoder rows missing
    assert lib.nerf_amd_param_gradients_convert_e4m3(null, null, null, one, 16, 2, null) == EINVAL                              # packed d_raw missing
    assert lib.nerf_amd_param_gradients_convert_e4m3(null, null, null, null, 0, 3, null) == 0
    assert lib.nerf_amd_param_gradients_finish_e4m3(one, one, one, one, 16, 3, null) == EINVAL
    assert lib.nerf_amd_param_gradients_finish_e4m3(one, one, null, one, 16, 0, null) == EINVAL
    assert lib.nerf_amd_hyper_fetch(null, 16, one, one, null) == EINVAL and lib.nerf_amd_hyper_fetch(one, 0, one, one, null) == EINVAL
    assert lib.nerf_amd_pinned_device_address(null) == EINVAL


def test_jump_polynomial_on_the_host(lib):
    """nerf_amd_mt19937_jump_poly (C++, host): x^(624 q) mod phi equals tools/make_mt_jump.py's big-integer
    square-and-multiply for small and table-sized q, a wrong phi is refused, and the polynomial does what it is for:
    the GF(2) convolution over the raw word sequence lands on the state q blocks later (sequential generation)."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("make_mt_jump", os.path.join(ROOT, "tools", "make_mt_jump.py"))
    J = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(J)
    z = np.load(os.path.join(ROOT, "nerf-simple_amd", "utils", "mt19937_jump.npz"))
    phi_words = np.ascontiguousarray(z["phi"].astype(np.uint32))
    phi = int.from_bytes(phi_words.astype("<u4").tobytes(), "little")

    def poly(q):
        out = np.zeros(624, dtype=np.uint32)
        assert lib.nerf_amd_mt19937_jump_poly(q, phi_words.ctypes.data, out.ctypes.data) == 0
        return out

    assert poly(0)[0] == 1 and not poly(0)[1:].any()
    for q in (1, 2, 31, 32, 33, 300, (16_000_000 - 1) // 624):
        assert np.array_equal(poly(q), J.to_words(J.x_pow_mod(624 * q, phi))), q
    t0 = time.time()
    poly((64_000_000 - 1) // 624)
    assert time.time() - t0 < 5.0
    bad = phi_words.copy()
    bad[623] = 0                                           # degree < 19937: not the characteristic polynomial
    assert lib.nerf_amd_mt19937_jump_poly(5, bad.ctypes.data, np.zeros(624, dtype=np.uint32).ctypes.data) == -1
    rng = np.random.default_rng(4)
    s = rng.integers(0, 2 ** 32, size=624, dtype=np.uint64).astype(np.uint32)
    seq = J.raw_words(s, 41)
    got = J.apply_jump(seq[624:2 * 624], poly(39))         # skip = 1: from the block AFTER the state, every word generated
    assert np.array_equal(got, seq[40 * 624:41 * 624])     # all 32 bits of all 624 words (nerf_amd_mt19937_advance)


def test_checkpoint_roundtrip(tmp_path, synthetic):
    """N4: checkpoints are the reference's format (24-key state_dict, strict load)."""
    import torch
    from nerf_simple_amd.utils.nets import Nerf
    from nerf_simple_amd.utils import checkpoint
    a = Nerf()
    a.load_state_dict(synthetic.synthetic_state_dict(3, "default"))
    path = checkpoint.save_checkpoint(a, str(tmp_path / "ckpt.pth"))
    raw = torch.load(path, weights_only=True)
    assert list(raw.keys()) == [k for k, _ in synthetic.PARAM_SPECS]
    b = checkpoint.load_checkpoint(Nerf(), path)
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    bad = dict(raw)
    bad.pop("sigma_fc.0.bias")
    torch.save(bad, str(tmp_path / "bad.pth"))
    with pytest.raises(KeyError):
        checkpoint.load_checkpoint(Nerf(), str(tmp_path / "bad.pth"))
    # a module of another size: its own checkpoint loads, the default one is refused with the offending tensor named
    small = Nerf(6, 2, 128)
    p2 = checkpoint.save_checkpoint(small, str(tmp_path / "small.pth"))
    again = checkpoint.load_checkpoint(Nerf(6, 2, 128), p2)
    assert all(torch.equal(x, y) for x, y in zip(small.state_dict().values(), again.state_dict().values()))
    with pytest.raises(ValueError, match="layers_0.0.weight"):
        checkpoint.load_checkpoint(Nerf(6, 2, 128), path)
    with pytest.raises(ValueError, match="layers_0.0.weight"):
        checkpoint.load_checkpoint(Nerf(), p2)


def test_counted_vmcnt_waits():
    """Each chunk barrier of the MLP kernels waits for vmcnt(N), N = the vector-memory
    instructions issued after the chunk's LDS-DMA pieces (csrc/nerf_device.h chunk_barrier).
    The count is a compile-time formula; this checks it against the generated gfx950 ISA
    (too small an N would be a race on the weight buffer)."""
    import shutil
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import check_vmcnt
    finally:
        sys.path.pop(0)
    for src, kernels, waits in (("mlp_bf16_16.hip", 5, 39), ("mlp_bwd_16.hip", 2, 39)):     # with the two e4m3-storage kernels
        asm = check_vmcnt.assemble(os.path.join(root, "nerf-simple_amd", "csrc", src))
        res = {k: check_vmcnt.check_kernel(v) for k, v in check_vmcnt.kernels_of(asm).items()}
        assert len(res) == kernels, (src, list(res))
        for name, (checked, bad) in res.items():
            assert checked == waits, (src, name, checked)
            assert not bad, (src, name, bad[:5])
        # and no 16-byte store has its data registers overwritten by the next instruction (the
        # store-data hazard that put NaNs into dY9: csrc/nerf_device.h store_granule)
        stores = {k: check_vmcnt.check_store_data_hazard(v) for k, v in check_vmcnt.kernels_of(asm).items()}
        assert sum(n for n, _ in stores.values()) >= (76 if src == "mlp_bf16_16.hip" else 152), (src, stores)
        for name, (n, offenders) in stores.items():
            assert not offenders, (src, name, offenders[:3])


def test_inference_kernels_have_few_hazard_nops():
    """csrc/mlp_bf16_16.hip chunk_step keeps a weight fragment and an accumulator alive a little past
    their last MFMA so that the registers coming free are not operands of the MFMA just issued: the
    hazard recognizer then has no write-after-read wait states to insert (617 s_nop per tile before,
    74 now).  A compiler or source change that brings them back costs 1-2 %."""
    import re
    import shutil
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import check_vmcnt
    finally:
        sys.path.pop(0)
    asm = check_vmcnt.assemble(os.path.join(root, "nerf-simple_amd", "csrc", "mlp_bf16_16.hip"))
    seen = 0
    for name, body in check_vmcnt.kernels_of(asm).items():
        if "ILb1ELb1ELb0EE" in name or "train_e4m3" in name:     # the training forwards: their stores bring their own
            continue
        lines = body if isinstance(body, list) else body.split("\n")
        mfma = [i for i, ln in enumerate(lines) if re.match(r"\s+v_mfma", ln)]
        assert len(mfma) == 2344, (name, len(mfma))
        nops = [ln for ln in lines[mfma[0]:mfma[-1] + 1] if re.match(r"\s+s_nop", ln)]
        assert len(nops) <= 150, (name, len(nops))
        seen += 1
    assert seen == 3


def test_stamp_tool_insertion_points():
    """tools/stamp_tiles.py (the diagnostic build behind DESIGN.md section 5's cycles-per-phase table) matches
    its insertion points in csrc/mlp_bf16_16.hip literally; this keeps them in step with the kernel source."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stamp_tiles", os.path.join(root, "tools", "stamp_tiles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = mod.stamped_source()                         # SystemExit when a pattern is missing or ambiguous
    assert src.count("STAMP(") == 19                   # the macro + 18 stamps
    shipped = open(os.path.join(root, "nerf-simple_amd", "csrc", "mlp_bf16_16.hip")).read()
    assert "s_memtime" not in shipped and "STAMP" not in shipped     # the shipped kernel carries none


def test_header_compiles_as_c_and_cxx(tmp_path):
    """include/nerf_amd.h is the contract a C caller binds: it must compile as plain C99 and as C++ without warnings,
    and every declared function must link against the shared library by its declared prototype."""
    import shutil
    import subprocess
    from nerf_simple_amd import _lib
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("gcc / g++ not available")
    inc = os.path.join(ROOT, "include")
    syms = declared_symbols()
    body = "#include \"nerf_amd.h\"\n#include <stdio.h>\nint main(void) {\n  void* p[] = {" + \
           ", ".join(f"(void*){s}" for s in syms) + "};\n  printf(\"%d %d\\n\", (int)(sizeof p / sizeof p[0]), nerf_amd_abi_version());\n  return 0;\n}\n"
    c_file, cxx_file = tmp_path / "abi.c", tmp_path / "abi.cc"
    c_file.write_text(body)
    cxx_file.write_text(body)
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wno-pedantic", "-I", inc, "-c", str(c_file), "-o",
                    str(tmp_path / "abi_c.o")], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(cxx_file), "-o", str(tmp_path / "abi_cc.o")],
                   check=True)
    # link the C object against the library: unresolved prototypes would fail here (no GPU needed to load it)
    exe = tmp_path / "abi"
    torch_lib = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")
    r = subprocess.run(["gcc", str(tmp_path / "abi_c.o"), "-o", str(exe), "-L", libdir, "-lnerf_amd", f"-Wl,-rpath,{libdir}",
                        f"-Wl,-rpath-link,{torch_lib}", f"-Wl,-rpath,{torch_lib}", "-L", torch_lib, "-Wl,--allow-shlib-undefined"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    n, ver = out.stdout.decode().split()
    assert int(n) == len(syms) and int(ver) == 5


def test_reference_jitter_segment_plan():
    """How utils/host_rng cuts a torch.rand(B, N) draw into workgroups (no GPU needed): one workgroup up to 39,936 new
    numbers behind the block's unread words, the one-launch jump form up to 64 short segments (training batches, the
    reference's 16,000-ray test batch), the doubling tree of 638,976-number segments beyond (images); and the count
    agrees with the library's own (nerf_amd_mt19937_segments)."""
    from nerf_simple_amd import _lib
    from nerf_simple_amd.utils.host_rng import segment_plan, _segments
    lib = _lib.lib()
    long_words, short_words, levels, n_short = 1024 * 624, 64 * 624, 10, 63
    for next0 in (0, 1, 333, 624):
        avail = 624 - next0
        for n in (0, 1, avail, avail + short_words, avail + short_words + 1, 4096 * 64, 4096 * 128, 16000 * 128,
                  avail + 64 * short_words, avail + 64 * short_words + 1, 640000 * 128, 3 * long_words + 5):
            for words in (long_words, short_words):
                assert _segments(next0, n, words) == lib.nerf_amd_mt19937_segments(next0, n, words), (next0, n, words)
            S, table, lv, words = segment_plan(next0, n, levels, long_words, n_short, short_words)
            if n <= avail + short_words:
                assert (S, table) == (1, None)
            elif n <= avail + 64 * short_words:
                assert table == "short" and lv == -63 and words == short_words and 2 <= S <= 64
                assert avail + (S - 1) * short_words < n <= avail + S * short_words
            else:
                assert table == "long" and lv == levels and words == long_words and S >= 2
                assert avail + (S - 1) * long_words < n <= avail + S * long_words
    assert segment_plan(0, 4096 * 64, levels, long_words, n_short, short_words)[0] == 7
    assert segment_plan(0, 16000 * 128, levels, long_words, n_short, short_words)[0] == 52
    assert segment_plan(0, 640000 * 128, levels, long_words, n_short, short_words)[:2] == (129, "long")
