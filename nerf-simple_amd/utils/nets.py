"""Drop-in for the reference's utils/nets.py: the ``Nerf`` module.

Same constructor, same 24 state-dict keys and shapes (fp32 master weights,
``load_state_dict(..., strict=True)`` compatible with reference checkpoints,
reference utils/nets.py:9-32, test.py:28), but ``forward`` runs the fused HIP
kernel (encoding + 12 dense layers in one launch, csrc/mlp_bf16_16.hip /
csrc/mlp_f32.hip) on an MFMA-fragment-ordered copy of the weights.  That packed
copy is a derived cache, rebuilt whenever a parameter changes.
"""
import warnings

import torch
import torch.nn as nn

from .. import _lib

# fp16 operands: the 16-bit mode that meets BASELINE's 0.05 dB PSNR criterion on every weight set
# tried (bf16's 8-bit weight mantissa does not on high-gain weights: DESIGN.md section 2); 5 % slower
# than bf16 (clock), values must stay below 65504.
DEFAULT_PRECISION = "fp16"


class _Packed:
    """One packed weight image of a module: the device buffer, the parameter versions it was derived from,
    and -- for the 16-bit images -- the state of the range guard (guarded_launch)."""
    __slots__ = ("stamp", "buf", "probed", "demoted", "host", "event")     # guard state: 16-bit images only

    def __init__(self, stamp, buf):
        self.stamp, self.buf = stamp, buf
        self.reset_guard()

    def reset_guard(self):
        self.probed = False          # the status word has been read once (blocking) after a launch with these weights
        self.demoted = False         # flagged: requests for this precision go to the next one (fp16 -> bf16 -> fp32)
        self.host = None             # pinned int32[2]: lazily copied status words of later launches
        self.event = None            # recorded behind that copy


def _status_flags(words):
    """The two status words of a packed 16-bit image (include/nerf_amd.h) as STATUS_* bits."""
    return (_lib.STATUS_NONFINITE if int(words[0]) else 0) | (_lib.STATUS_WEIGHT_RANGE if int(words[1]) else 0)


def packed_status(packed, code):
    """STATUS_* bits of a packed image on the device (synchronises); 0 for images without a status block."""
    off = int(_lib.lib().nerf_amd_packed_status_offset(code))
    return 0 if off < 0 else _status_flags(packed[off:off + 8].view(torch.int32).cpu())


_NEXT_PRECISION = {_lib.FP16: _lib.BF16, _lib.BF16: _lib.F32}


def guarded_launch(nets, code, launch):
    """Run ``launch(code, [packed image of each net])`` -- which enqueues kernels through the C ABI and returns
    their outputs -- under the range guard of the 16-bit kernels.

    The reference network is fp32 with no range limit (utils/nets.py:16-32); fp16 MFMA operands overflow beyond
    65504, and the 16-bit kernels' integer ReLU does not even keep the resulting NaN (it zeroes one whose sign bit is
    set): an overflow ends as NaN pixels or as finite garbage.  Those kernels therefore set a sticky flag in the
    status block behind the packed image whenever a point shows a non-finite accumulator or output, and the packer
    flags weights that do not fit (include/nerf_amd.h nerf_amd_packed_status_offset).  Policy, per weight set:
      * the FIRST launch of a 16-bit precision is followed by one blocking read of the flags; if one is set, a
        UserWarning is issued, the module is demoted for these weights -- fp16 to bf16 operands (fp32's exponent range),
        bf16 to the fp32 kernel, whose ReLU keeps NaN like torch's, so that non-finite inputs or weights show in the
        outputs exactly as in the reference -- and the call is repeated: the caller never sees the bad pixels;
      * later launches (other rays may still overflow) copy the flags to pinned memory asynchronously and the NEXT
        call looks at them without waiting: demotion then takes effect from that call on, with the warning naming the
        earlier render."""
    off = None
    while code != _lib.F32:
        ents = [n._packed_entry(code) for n in nets]
        if any(e.demoted for e in ents):
            code = _NEXT_PRECISION[code]
            continue
        dev = ents[0].buf.device
        capturing = torch.cuda.is_current_stream_capturing()
        if off is None:
            off = int(_lib.lib().nerf_amd_packed_status_offset(code))      # the same for both 16-bit images
        late = 0
        if not capturing:
            for n, e in zip(nets, ents):
                if e.event is not None and e.event.query():
                    flags, e.event = _status_flags(e.host), None
                    if flags:
                        n._demote(e, code, flags, "an earlier render")
                        late |= flags
        if late:
            code = _NEXT_PRECISION[code]
            continue
        out = launch(code, [e.buf for e in ents])
        if capturing:
            return out
        redo = False
        for n, e in zip(nets, ents):
            word = e.buf[off:off + 8].view(torch.int32)     # [non-finite value seen, weight out of range]
            if not e.probed:
                flags = _status_flags(word.cpu())           # blocking, once per weight set and precision
                e.probed = True
                if flags:
                    n._demote(e, code, flags, "this render (repeated)")
                    redo = True
            elif e.event is None:
                if e.host is None:
                    e.host = torch.zeros(2, dtype=torch.int32).pin_memory()
                e.host.copy_(word, non_blocking=True)
                e.event = torch.cuda.Event()
                e.event.record(torch.cuda.current_stream(dev))
        if not redo:
            return out
        code = _NEXT_PRECISION[code]
    return launch(code, [n.packed_weights(code) for n in nets])


class Nerf(nn.Module):
    """8x256 ReLU MLP with a skip-concat of the encoded position after layer 5,
    a sigma head, and a view-direction colour head (reference utils/nets.py:8-43).

    forward(v): v [P,6] = [x,y,z,d1,d2,d3] -> [P,4] = [r,g,b,sigma], raw (no
    sigmoid on rgb; softplus on sigma is applied by the compositor).

    Sizes other than the default (10, 4, 256) are accepted, as in the reference, and run layer by layer in fp32
    (utils/generic_mlp.py: forward and backward on nerf_amd_linear_f32); everything below about `precision`, packed
    weight images and the fused training kernels concerns the default shape.

    precision: 'fp16' (fp16 MFMA operands, fp32 accumulate; default: 11-bit mantissa,
               hidden activations must stay below 65504), 'bf16' (bf16 operands: same
               cycles, 5 % faster clock, 8-bit mantissa) or 'fp32' (exact-f32 MFMA).  Keyword-only
               superset of the reference signature.  It selects the INFERENCE kernel;
               with gradients enabled 'bf16' and 'fp16' modules both run the bf16
               training kernels (training.py); 'fp32' trains exactly, layer by layer (utils/generic_mlp.py).
    """

    def __init__(self, Lp=10, Ld=4, H=256, *, precision=None):
        super().__init__()
        import numbers
        if not all(isinstance(x, numbers.Integral) for x in (Lp, Ld, H)) or Lp < 1 or Ld < 1 or H < 2:
            # (the reference's encoder concatenates an empty list at L = 0, utils/xyz.py:6-14; H // 2 = 0 has no colour head)
            raise RuntimeError(f"Nerf(Lp={Lp}, Ld={Ld}, H={H}): sizes must be integers with Lp, Ld >= 1 and H >= 2")
        # Nerf() = (10, 4, 256) -- the one shape the reference ever constructs (train.py:41, test.py:27) -- runs on the
        # fused kernels; any other size runs layer by layer in fp32 on the strided GEMM kernel (utils/generic_mlp.py):
        # same results as the reference module, `precision` has no effect there
        Lp, Ld, H = int(Lp), int(Ld), int(H)
        self.Lp, self.Ld, self.H = Lp, Ld, H
        self.precision = precision or DEFAULT_PRECISION
        _lib.precision_code(self.precision)
        cx, cd = 3 + 6 * Lp, 3 + 6 * Ld

        def relu_stack(dims):
            mods = []
            for a, b in zip(dims[:-1], dims[1:]):
                mods += [nn.Linear(a, b), nn.ReLU()]
            return nn.Sequential(*mods)

        # module names fix the state-dict keys (checkpoint contract)
        self.layers_0 = relu_stack([cx, H, H, H, H, H])
        self.skip_conn_layer = relu_stack([H + cx, H])
        self.layers_1 = relu_stack([H, H, H])
        self.sigma_fc = nn.Sequential(nn.Linear(H, 1))
        self.layers_2 = nn.Linear(H, H)
        self.color_fc = nn.Sequential(nn.Linear(H + cd, H // 2), nn.ReLU(), nn.Linear(H // 2, 3))
        self._packed = {}       # (device, precision code) -> _Packed

    # the packed images are a derived cache (device buffers, pinned words, events): copies and pickles of the module
    # start without them and re-pack on first use
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_packed"] = {}
        state.pop("_watch", None)                # training.py's status watch: pinned words and events
        state.pop("_watch_calls", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._packed = {}

    # ---- packed-weight cache ------------------------------------------------
    def _fused_ok(self):
        return (self.Lp, self.Ld, self.H) == (10, 4, 256)

    def _param_list(self):
        return [p for _, p in self.named_parameters()]

    def packed_weights(self, precision=None):
        """Device buffer with the weight image the fused kernels stream; packs on
        first use and again after any in-place parameter update / reassignment."""
        return self._packed_entry(precision).buf

    def _packed_entry(self, precision=None):
        code = _lib.precision_code(self.precision if precision is None else precision)
        if not self._fused_ok():
            raise RuntimeError(f"Nerf(Lp={self.Lp}, Ld={self.Ld}, H={self.H}) has no packed weight image (NERF_AMD_EUNSUP): "
                               "the fused kernels implement (10, 4, 256); other sizes run through utils/generic_mlp.py")
        params = self._param_list()
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("Nerf must be moved to the GPU (.cuda()) before use; there is no CPU path")
        key = (dev, code)
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._packed.get(key)
        if hit is not None and hit.stamp == stamp:
            return hit
        lib = _lib.lib()
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in params])
        if flat.numel() != lib.nerf_amd_param_count():
            raise RuntimeError("unexpected parameter count")
        packed = torch.empty(lib.nerf_amd_packed_bytes(code), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(packed), code,
                                                 _lib.stream_ptr(dev)), "nerf_amd_pack_weights")
        self._packed[key] = _Packed(stamp, packed)
        return self._packed[key]

    def _demote(self, entry, code, flags, where):
        entry.demoted = True
        why = []
        if flags & _lib.STATUS_WEIGHT_RANGE:
            why.append("a weight beyond 65504" if code == _lib.FP16 else "a weight that is not finite")
        if flags & _lib.STATUS_NONFINITE:
            why.append(f"a non-finite value inside the network in {where}")
        if code == _lib.FP16:
            msg = ("Nerf: fp16 MFMA operands left their range (" + " and ".join(why) + "); these weights now render with "
                   "bf16 operands (precision='bf16': fp32's exponent range, 8-bit mantissa) until they change")
        else:
            msg = ("Nerf: " + " and ".join(why) + " with bf16 operands (non-finite inputs or weights?); these weights now "
                   "render with the fp32 kernel, which propagates NaN / inf like the reference, until they change")
        warnings.warn(msg, UserWarning, stacklevel=4)

    def repack_from_flat(self, flat):
        """Re-derive every packed image already in use from a flat fp32 parameter
        vector (state_dict order) whose views ARE this module's parameters
        (optim.FusedAdam): no per-tensor concatenation, cache stays valid."""
        if not self._fused_ok():
            return                                   # sizes on the layer-by-layer path read the parameters themselves
        lib = _lib.lib()
        dev = flat.device
        stamp = tuple((p.data_ptr(), p._version) for p in self._param_list())
        codes = {code for (d, code) in self._packed if d == dev} or {_lib.precision_code(self.precision)}
        with torch.cuda.device(dev):
            if _lib.BF16 in codes and _lib.BF16_BWD in codes:
                # the training pair in one launch
                a, b = self._packed[(dev, _lib.BF16)].buf, self._packed[(dev, _lib.BF16_BWD)].buf
                _lib.check(lib.nerf_amd_pack_weights_train(_lib.ptr(flat), _lib.ptr(a), _lib.ptr(b), _lib.stream_ptr(dev)),
                           "nerf_amd_pack_weights_train")
                self._packed[(dev, _lib.BF16)] = _Packed(stamp, a)
                self._packed[(dev, _lib.BF16_BWD)] = _Packed(stamp, b)
                codes = codes - {_lib.BF16, _lib.BF16_BWD}
            for code in codes:
                hit = self._packed.get((dev, code))
                packed = hit.buf if hit is not None else torch.empty(
                    lib.nerf_amd_packed_bytes(code), dtype=torch.uint8, device=dev)
                _lib.check(lib.nerf_amd_pack_weights(_lib.ptr(flat), _lib.ptr(packed), code,
                                                     _lib.stream_ptr(dev)), "nerf_amd_pack_weights")
                self._packed[(dev, code)] = _Packed(stamp, packed)       # new weights: the range guard starts over

    def drop_packed(self, dev, keep=()):
        """Forget the packed images of device ``dev`` except the precision codes in ``keep``: for callers that
        update the parameters behind autograd's back (training.GraphedTrainStep re-packs only the images its
        graphs use); the next use of a dropped image packs it again."""
        for key in [k for k in self._packed if k[0] == dev and k[1] not in keep]:
            del self._packed[key]

    # ---- forward ---------------------------------------------------------------
    def forward(self, v, *, precision=None):
        _lib.require_cuda_f32(v, "v")
        if v.dim() != 2 or v.shape[1] != 6:
            raise RuntimeError("Nerf.forward expects a [P, 6] tensor")
        from .xyz import range_check_values
        range_check_values(v)                        # the reference's range warning (utils/xyz.py:8-9), raised lazily
        if not self._fused_ok():
            from . import generic_mlp
            if torch.is_grad_enabled():
                return generic_mlp.forward(self, v)
            with torch.no_grad():
                return generic_mlp.forward(self, v)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ..training import nerf_forward_autograd
            return nerf_forward_autograd(self, v, self.precision if precision is None else precision)
        return self.forward_inference(v, precision=precision)

    def forward_inference(self, v, *, precision=None):
        if not self._fused_ok():
            with torch.no_grad():
                return self.forward(v)
        code = _lib.precision_code(self.precision if precision is None else precision)
        v = v.detach().contiguous()
        P = v.shape[0]

        def launch(code, packed):
            out = torch.empty((P, 4), dtype=torch.float32, device=v.device)
            with torch.cuda.device(v.device):
                _lib.check(_lib.lib().nerf_amd_mlp_forward(_lib.ptr(v), _lib.ptr(packed[0]), _lib.ptr(out),
                                                           P, code, _lib.stream_ptr(v.device)),
                           "nerf_amd_mlp_forward")
            return out

        return guarded_launch([self], code, launch)


    def fp16_headroom(self, v):
        """Largest hidden activation of the network on the query points v [P,6], as a fraction of the fp16
        range (65504): the default precision='fp16' render needs this well below 1 (a trained NeRF sits
        around 1e-4 .. 1e-3); use precision='bf16' otherwise.  Diagnostic: runs the bf16 training forward
        once and scans the activations it saves (layers 0..9, post-ReLU / linear)."""
        _lib.require_cuda_f32(v, "v")
        if not self._fused_ok():
            return 0.0                               # fp32 throughout: no 16-bit range to leave
        lib = _lib.lib()
        v = v.detach().contiguous()
        P = v.shape[0]
        if P == 0:
            return 0.0
        nbytes = int(lib.nerf_amd_train_activation_bytes(P))
        bf16_bytes = 10 * ((P + 255) // 256) * 256 * 512            # the point-blocked bf16 region (nerf_amd.h)
        acts = torch.zeros(nbytes, dtype=torch.uint8, device=v.device)
        out = torch.empty((P, 4), dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(lib.nerf_amd_mlp_forward_train_points(_lib.ptr(v), _lib.ptr(self.packed_weights(_lib.BF16)),
                                                             _lib.ptr(out), _lib.ptr(acts), P, _lib.stream_ptr(v.device)),
                       "nerf_amd_mlp_forward_train_points")
        return float(acts[:bf16_bytes].view(torch.bfloat16).float().abs().max()) / 65504.0


class CoarseNet(nn.Module):
    """Placeholder, as in the reference (utils/nets.py:45-46: hierarchical
    sampling is not implemented there)."""


class FineNet(nn.Module):
    """Placeholder, as in the reference (utils/nets.py:48-49)."""
