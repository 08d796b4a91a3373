"""The reference's jitter draw, generated on the GPU with the reference's own numbers.

``render_nerf`` in the reference draws ``u = torch.rand(B, N)`` on torch's CPU default
generator and moves it to the GPU (utils/rendering.py:28-30).  Keeping that contract -- same
values, same advance of the caller's generator -- used to cost more than the render itself
(MT19937 on one host core + a 4 B/sample PCIe copy: 247 ms against a 65 ms render at
800x800x128).  ``reference_rand`` continues the generator's stream on the device instead
(csrc/host_rng.hip): it uploads the 624 state words, lets one workgroup produce the draws
straight into HBM, and writes the words and counters back into the generator -- 5 KB each way.

The generator state is torch's ``CPUGeneratorImplState`` byte image (ATen/CPUGeneratorImpl.cpp):
    0  uint64 seed | 8 int32 left | 12 int32 seeded | 16 uint64 next | 24 uint64 state[624]
    | 5016 .. normal-distribution cache (untouched)
with at::mt19937's bookkeeping: a draw first decrements ``left`` and regenerates the block when
it reaches 0 (then left = 624, next = 0), then returns tempered state[next++].  The layout is
checked against torch itself before first use (``layout_ok``); if the check fails -- another
torch version with another layout -- the draw falls back to the reference's own
``torch.rand(B, N).to(device)``, which is the same numbers by definition.
Set NERF_AMD_HOST_RNG=1 to force that path.
"""
import os
import struct

import numpy as np
import torch

from .. import _lib

_N = 624
_STATE_BYTES = 5056
_layout_ok = None


def _parse(state):
    b = state.numpy().tobytes()
    _seed, left, seeded, nxt = struct.unpack_from("<QiiQ", b, 0)
    words = np.frombuffer(b, dtype="<u8", count=_N, offset=24)
    return left, seeded, nxt, words


def _advance(left, nxt, n):
    """(left, next, new blocks) after n draws (at::mt19937 bookkeeping)."""
    avail = left - 1                       # unread words of the current block
    if n <= avail:
        return left - n, nxt + n, 0
    r = n - avail
    blocks = (r + _N - 1) // _N
    used = r - (blocks - 1) * _N           # words read from the last new block, 1..624
    return _N + 1 - used, used, blocks


def _patched(state, left, nxt, words32):
    b = bytearray(state.numpy().tobytes())
    struct.pack_into("<i", b, 8, int(left))
    struct.pack_into("<Q", b, 16, int(nxt))
    if words32 is not None:
        b[24:24 + 8 * _N] = words32.astype("<u8").tobytes()
    return torch.frombuffer(b, dtype=torch.uint8).clone()


def layout_ok():
    """Pin the state layout and the bookkeeping against torch itself (CPU only, microseconds):
    a scratch generator is advanced by torch.rand, and the counters predicted from its previous
    state by ``_advance`` must match the ones torch reports."""
    global _layout_ok
    if _layout_ok is None:
        try:
            g = torch.Generator()
            g.manual_seed(20240229)
            ok = g.get_state().numel() == _STATE_BYTES
            for n in (3, 700, 624 - 79, 1300):
                if not ok:
                    break
                left, seeded, nxt, words = _parse(g.get_state())
                ok = seeded == 1 and 0 <= nxt <= _N and 1 <= left <= _N and int(words.max()) < 2 ** 32
                torch.rand(n, generator=g)
                l2, _, n2, w2 = _parse(g.get_state())
                l1, n1, blocks = _advance(left, nxt, n)
                ok = ok and (l1, n1) == (l2, n2) and (blocks > 0 or np.array_equal(words, w2))
            _layout_ok = bool(ok)
        except Exception:
            _layout_ok = False
    return _layout_ok


_jump = {}


def _jump_polys(device):
    """(polys [levels, 624] on ``device``, levels, seg_words, short_polys [63, 624], short_seg_words) from
    utils/mt19937_jump.npz, or None."""
    key = str(device)
    if key not in _jump:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt19937_jump.npz")
        try:
            z = np.load(path)
            polys = torch.from_numpy(z["polys"].astype(np.uint32).view(np.int32)).to(device)
            short = torch.from_numpy(z["short_polys"].astype(np.uint32).view(np.int32)).to(device)
            _jump[key] = (polys, int(polys.shape[0]), int(z["seg_words"]), short, int(z["short_seg_words"]))
        except Exception:
            _jump[key] = None
    return _jump[key]


def _segments(next0, n, seg_words):
    """Segments of a draw of n numbers (nerf_amd_mt19937_segments): the first takes the block's unread words and
    seg_words new ones, every further one seg_words."""
    avail = _N - int(next0)
    return 1 + -(-(n - avail - seg_words) // seg_words) if n > avail + seg_words else 1


def segment_plan(next0, n, levels, seg_words, n_short, short_words):
    """How a draw of n numbers is cut: (S, table, levels argument, segment words) with table 'short' (every start state
    from state 0 in ONE jump launch: training batches, the reference's test batch), 'long' (doubling tree: images) or
    None (one workgroup, S = 1)."""
    S_short = _segments(next0, n, short_words)
    if 1 < S_short <= 1 + n_short:
        return S_short, "short", -n_short, short_words
    S = _segments(next0, n, seg_words)
    if 1 < S <= (1 << levels):
        return S, "long", levels, seg_words
    return 1, None, 0, 0


def _launch_uniform(words_dev, next0, out, n, state_out, device):
    """nerf_amd_mt19937_uniform, or its multi-workgroup form when the draw spans several segments."""
    lib = _lib.lib()
    st = _lib.stream_ptr(device)
    jp = _jump_polys(device)
    if jp is not None:
        long_polys, levels, seg_words, short, short_words = jp
        S, table, levels, seg_words = segment_plan(next0, n, levels, seg_words, int(short.shape[0]), short_words)
        if table is not None:
            polys = short if table == "short" else long_polys
            ws = torch.empty((S, _N), dtype=torch.int32, device=device)
            _lib.check(lib.nerf_amd_mt19937_uniform_par(_lib.ptr(words_dev), int(next0), _lib.ptr(out), int(n),
                                                        _lib.ptr(state_out), _lib.ptr(polys), levels, seg_words,
                                                        _lib.ptr(ws), st), "nerf_amd_mt19937_uniform_par")
            ws.record_stream(torch.cuda.current_stream(device))
            return
    _lib.check(lib.nerf_amd_mt19937_uniform(_lib.ptr(words_dev), int(next0), _lib.ptr(out), int(n), _lib.ptr(state_out), st),
               "nerf_amd_mt19937_uniform")


_side_streams = {}


def _side_stream(device):
    """One high-priority side stream per device, reused: a fresh torch.cuda.Stream() comes from a
    round-robin pool and now and then shares the default stream's hardware queue, which
    serialises the generator behind the renders (seen in a kernel trace: same queue id, no
    overlap); the high-priority pool maps to other queues."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device, priority=-1)
    return _side_streams[key]


class GeneratorSession:
    """torch's CPU default generator continued on the device over CONSECUTIVE consumers -- the reference's training
    iteration draws ``torch.randperm(n)`` (RayGenerator.select, utils/dataload.py:151) and then ``torch.rand(B, N)``
    (render_nerf, utils/rendering.py:28) from the one stream.  The 624 state words go up once, every consumer leaves the
    words it ends on in device memory for the next, the counters are kept here (at::mt19937's bookkeeping, ``_advance``),
    and ``finish()`` -- to be called once everything that follows has been enqueued -- waits for the generator kernels
    alone and writes words and counters back into torch's generator: values and generator state are torch's, bit for bit."""

    def __init__(self, device):
        self.device = device
        self.state = torch.get_rng_state()
        self.left, _seeded, self.nxt, words = _parse(self.state)
        # (a pinned staging ring for this 2.5 KB upload was tried: the host then runs ahead of the GPU and the graphed
        # reference-stream iteration got SLOWER, 1.38 -> 2.01 ms; not understood, not kept)
        self.words_dev = torch.from_numpy(words.astype(np.uint32).view(np.int32)).to(device)
        self.changed = False                     # the words on the device are no longer the generator's
        self.event = None

    def _first_unread(self):
        return _N + 1 - int(self.left)           # 624 = block exhausted, as after seeding

    def _moved(self, new_words, left, nxt, blocks):
        if blocks > 0:
            self.words_dev, self.changed = new_words, True
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(self.device))
        self.left, self.nxt = left, nxt

    def rand(self, B, N, out=None):
        """``torch.rand(B, N).to(device)``: same values, same advance of the stream (``out``: a contiguous [B,N] buffer)."""
        n = int(B) * int(N)
        u = torch.empty((B, N), dtype=torch.float32, device=self.device) if out is None else out
        if n == 0:
            return u
        left, nxt, blocks = _advance(self.left, self.nxt, n)
        state_out = torch.empty(_N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _launch_uniform(self.words_dev, self._first_unread(), u, n, state_out, self.device)
            self._moved(state_out, left, nxt, blocks)
        return u

    def randperm_draws(self, n, B):
        """The 32-bit outputs ``torch.randperm(n)`` takes its first min(B, n - 1) swap positions from (uint32 in an int32
        tensor), with the stream moved past all n - 1 draws of the call (randperm_cpu, n < 2^32 / 20: one draw per
        element but the last).  The draws nobody looks at are jumped over: x^(624 q) mod phi, one polynomial per table
        size and block phase, made on first use (nerf_amd_mt19937_jump_poly) -- see csrc/select.hip."""
        n, B = int(n), int(B)
        total = max(n - 1, 0)
        first = min(B, total)
        lib = _lib.lib()
        draws = torch.empty(max(first, 1), dtype=torch.int32, device=self.device)
        if total == 0:
            return draws[:0]
        left, nxt, blocks = _advance(self.left, self.nxt, total)
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            if blocks <= _SEQUENTIAL_BLOCKS:
                # a short permutation: draw all of it (one workgroup, 0.6 us per block) and keep the head
                everything = torch.empty(total, dtype=torch.int32, device=self.device)
                state_out = torch.empty(_N, dtype=torch.int32, device=self.device)
                _lib.check(lib.nerf_amd_mt19937_raw(_lib.ptr(self.words_dev), self._first_unread(), _lib.ptr(everything), total,
                                                    _lib.ptr(state_out), st), "nerf_amd_mt19937_raw")
                draws = everything[:first]
            else:
                _lib.check(lib.nerf_amd_mt19937_raw(_lib.ptr(self.words_dev), self._first_unread(), _lib.ptr(draws), first, None, st),
                           "nerf_amd_mt19937_raw")
                state_out = torch.empty(_N, dtype=torch.int32, device=self.device)
                _lib.check(lib.nerf_amd_mt19937_advance(_lib.ptr(self.words_dev), _lib.ptr(advance_poly(blocks - 1, self.device)),
                                                        _lib.ptr(state_out), st), "nerf_amd_mt19937_advance")
            self._moved(state_out, left, nxt, blocks)
        return draws[:first]

    def randperm_then_rand(self, n, B_sel, B, N, out=None):
        """``torch.randperm(n)`` followed by ``torch.rand(B, N)`` -- the reference's rg.select and render_nerf's jitter draw
        (train.py:47-51) -- as (first min(B_sel, n - 1) raw draws of the shuffle, u [B,N]).  For a large table the jump over
        the shuffle's unused draws and the jump to the jitter's segment starts are ONE launch with summed distances
        (nerf_amd_mt19937_uniform_after) instead of two dependent ones; otherwise the two calls above."""
        n, B_sel, nj = int(n), int(B_sel), int(B) * int(N)
        total = max(n - 1, 0)
        left1, nxt1, blocks1 = _advance(self.left, self.nxt, total)
        jp = _jump_polys(self.device)
        if blocks1 <= _SEQUENTIAL_BLOCKS or nj == 0 or jp is None:
            return self.randperm_draws(n, B_sel), self.rand(B, N, out=out)
        short_words = jp[4]
        next1 = _N + 1 - int(left1)
        S = _segments(next1, nj, short_words)
        if S > 256:
            return self.randperm_draws(n, B_sel), self.rand(B, N, out=out)
        lib = _lib.lib()
        first = min(B_sel, total)
        draws = torch.empty(max(first, 1), dtype=torch.int32, device=self.device)
        u = torch.empty((B, N), dtype=torch.float32, device=self.device) if out is None else out
        state_out = torch.empty(_N, dtype=torch.int32, device=self.device)
        seg_states = torch.empty((S, _N), dtype=torch.int32, device=self.device)
        polys = advance_poly_table(blocks1 - 1, short_words // _N, S, self.device)
        left2, nxt2, _blocks2 = _advance(left1, nxt1, nj)
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            _lib.check(lib.nerf_amd_mt19937_raw(_lib.ptr(self.words_dev), self._first_unread(), _lib.ptr(draws), first, None, st),
                       "nerf_amd_mt19937_raw")
            _lib.check(lib.nerf_amd_mt19937_uniform_after(_lib.ptr(self.words_dev), _lib.ptr(polys), S, next1, _lib.ptr(u), nj,
                                                          _lib.ptr(state_out), short_words, _lib.ptr(seg_states), st),
                       "nerf_amd_mt19937_uniform_after")
            seg_states.record_stream(torch.cuda.current_stream(self.device))
            self._moved(state_out, left2, nxt2, 1)          # the shuffle alone has regenerated: the words are new
        return draws[:first], u

    def finish(self):
        """Make torch's generator current (a wait for the generator kernels only, not for what was enqueued behind them)."""
        if self.state is None:
            return
        words = None
        if self.changed:
            side = _side_stream(self.device)
            with torch.cuda.stream(side):
                side.wait_event(self.event)
                host = self.words_dev.to("cpu")
            side.synchronize()
            words = host.numpy().view(np.uint32)
        torch.set_rng_state(_patched(self.state, self.left, self.nxt, words))
        self.state = None


_SEQUENTIAL_BLOCKS = 48                      # up to ~30,000 draws are cheaper drawn than jumped over (85 us)
_phi = None
_advance_polys = {}


def advance_poly(q, device):
    """x^(624 q) mod phi on ``device`` (624 words), cached: the jump of nerf_amd_mt19937_advance."""
    global _phi
    key = (int(q), str(device))
    if key not in _advance_polys:
        host_key = (int(q), "host")
        if host_key not in _advance_polys:
            if _phi is None:
                path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt19937_jump.npz")
                _phi = np.ascontiguousarray(np.load(path)["phi"].astype(np.uint32))
            out = np.zeros(_N, dtype=np.uint32)
            _lib.check(_lib.lib().nerf_amd_mt19937_jump_poly(int(q), _phi.ctypes.data, out.ctypes.data), "nerf_amd_mt19937_jump_poly")
            _advance_polys[host_key] = out
        _advance_polys[key] = torch.from_numpy(_advance_polys[host_key].view(np.int32)).to(device)
    return _advance_polys[key]


_poly_tables = {}


def advance_poly_table(q0, step, count, device):
    """[count, 624] on ``device``: x^(624 (q0 + b * step)) mod phi, b < count (nerf_amd_mt19937_uniform_after)."""
    key = (int(q0), int(step), int(count), str(device))
    if key not in _poly_tables:
        if len(_poly_tables) > 64:
            _poly_tables.clear()
        _poly_tables[key] = torch.stack([advance_poly(q0 + b * step, device) for b in range(count)]).contiguous()
    return _poly_tables[key]


class _Done:
    """A finished draw (host fallback)."""

    def finish(self):
        pass


def reference_rand(B, N, device):
    """``torch.rand(B, N).to(device)`` of the reference -- same values, same effect on torch's CPU
    default generator -- without generating or copying B*N floats on the host.
    Returns (u [B, N] on ``device``, pending): call ``pending.finish()`` once the launches that
    follow have been enqueued (it waits for the generator kernel and restores the generator)."""
    n = int(B) * int(N)
    if host_fallback() or n == 0:
        return torch.rand(B, N).to(device), _Done()
    session = GeneratorSession(device)
    return session.rand(B, N), session


def host_fallback():
    """The draws come from torch on the host (NERF_AMD_HOST_RNG=1, or a torch build whose generator layout is not the
    one pinned by ``layout_ok``): the reference's own calls, by definition the same numbers."""
    return os.environ.get("NERF_AMD_HOST_RNG") == "1" or not layout_ok()


class ReferenceJitter:
    """The jitter of CONSECUTIVE render_nerf calls -- the batches of an image driver.  The
    reference draws ``torch.rand(B_k, N)`` inside each call, i.e. consecutive pieces of one stream,
    so the whole image's draws are produced by ONE launch sequence before the first batch (with
    jump-ahead that is ~3 ms for 8.2e7 draws, far cheaper than overlapping per-batch pieces with the
    renders: the jump kernels need 80 KiB of LDS and would wait for the persistent render grid).
    ``batch(k)`` returns piece k; ``finish()`` restores torch's CPU generator to where the
    reference's draws would have left it."""

    def __init__(self, batch_rays, N, device):
        self.device, self.N = device, int(N)
        self.sizes = [int(b) for b in batch_rays]
        self.fallback = host_fallback() or not self.sizes
        self.pending = _Done()
        if self.fallback:
            return
        self.offsets, row = [], 0
        for b in self.sizes:
            self.offsets.append(row)
            row += b
        self.u, self.pending = reference_rand(row, self.N, device)

    def batch(self, k):
        if self.fallback:
            return torch.rand(self.sizes[k], self.N).to(self.device)
        return self.u[self.offsets[k]:self.offsets[k] + self.sizes[k]]

    def finish(self):
        self.pending.finish()
