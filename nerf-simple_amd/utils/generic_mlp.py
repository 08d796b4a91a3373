"""Nerf(Lp, Ld, H) of sizes the fused kernels are not built for (reference utils/nets.py:8-43 takes any; it only ever
constructs Nerf() = (10, 4, 256), which the fused kernels implement).  The same network layer by layer in fp32: the HIP
positional encoder, then every nn.Linear -- forward and backward -- as one strided GEMM kernel
(csrc/linear_generic.hip, nerf_amd_linear_f32).  Activations live in HBM between layers; this is the tidy path for
unusual shapes, not the fast one.

    forward  (utils/nets.py:34-43):  x, d = positional_encoder(v)
        h = layers_0(x)                       5 x (Linear + ReLU)
        h = skip_conn_layer(cat(h, x))        Linear + ReLU; the concatenation is two GEMMs into one output
        h = layers_1(h)                       2 x (Linear + ReLU)
        sigma = sigma_fc(h);  f = layers_2(h) (no ReLU)
        rgb = color_fc(cat(f, d))             Linear + ReLU + Linear
        out = cat(rgb, sigma)                 both heads write their columns of out[P, 4]
    backward: autograd of exactly that, hand-written: dX = (dY * relu') W, dW = (dY * relu')^T X, db = (dY * relu')^T 1,
    the ReLU derivative taken from the saved activation inside the GEMM's operand load.
"""
import torch

from .. import _lib
from .xyz import positional_encoder

RELU, ACC = 1, 2

# state-dict order of the reference module (utils/nets.py:16-32): 12 Linear layers
LAYER_NAMES = ("layers_0.0", "layers_0.2", "layers_0.4", "layers_0.6", "layers_0.8", "skip_conn_layer.0",
               "layers_1.0", "layers_1.2", "sigma_fc.0", "layers_2", "color_fc.0", "color_fc.2")


def _gemm(A, sa_i, sa_k, B, sb_k, sb_j, C, ldc, M, N, K, *, mask=None, bias=None, flags=0, a_off=0, b_off=0, c_off=0):
    """C[c_off + i*ldc + j] (+)= sum_k A[a_off + i*sa_i + k*sa_k] * B[b_off + k*sb_k + j*sb_j]; offsets in elements."""
    dev = C.device
    pa = A.data_ptr() + 4 * a_off
    pm = 0 if mask is None else mask.data_ptr() + 4 * a_off
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nerf_amd_linear_f32(pa, sa_i, sa_k, pm or None, B.data_ptr() + 4 * b_off, sb_k, sb_j,
                                                  _lib.ptr(bias), C.data_ptr() + 4 * c_off, ldc, M, N, K, flags,
                                                  _lib.stream_ptr(dev)), "nerf_amd_linear_f32")


def _linear(x, w, b, out, *, relu, w_col0=0, accumulate=False, out_col0=0):
    """out[:, out_col0 : out_col0 + w.shape[0]] (+)= x @ w[:, w_col0 : w_col0 + x.shape[1]].T (+ b) (ReLU)."""
    P, K = x.shape
    n_out, ldw = w.shape
    _gemm(x, K, 1, w, 1, ldw, out, out.shape[1], P, n_out, K, bias=b, b_off=w_col0, c_off=out_col0,
          flags=(RELU if relu else 0) | (ACC if accumulate else 0))


class _GenericMlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, Lp, Ld, *params):
        w = [p.detach() for p in params[0::2]]
        b = [p.detach() for p in params[1::2]]
        v = _lib.require_cuda_f32(v, "v").contiguous()
        if any(not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) for p in w + b):
            raise RuntimeError("Nerf parameters must be contiguous fp32 tensors on the GPU")
        P, H = v.shape[0], w[0].shape[0]
        new = lambda n: torch.empty((P, n), dtype=torch.float32, device=v.device)    # noqa: E731
        posx, posd = positional_encoder(v.detach(), Lp, Ld)
        keep = any(ctx.needs_input_grad[3:])
        acts = []                                  # post-ReLU outputs of layers 0..7 (what the backward needs)
        h = posx
        for L in range(5):
            y = new(H)
            _linear(h, w[L], b[L], y, relu=True)
            if keep:
                acts.append(y)
            h = y
        y = new(H)                                 # skip: cat(h, posx) @ W^T = h @ W[:, :H]^T + posx @ W[:, H:]^T
        _linear(h, w[5], b[5], y, relu=False)
        _linear(posx, w[5], None, y, relu=True, w_col0=H, accumulate=True)
        if keep:
            acts.append(y)
        else:
            del posx
        h = y
        for L in (6, 7):
            y = new(H)
            _linear(h, w[L], b[L], y, relu=True)
            if keep:
                acts.append(y)
            h = y
        out = new(4)
        _linear(h, w[8], b[8], out, relu=False, out_col0=3)                          # sigma -> out[:, 3]
        f = new(H)
        _linear(h, w[9], b[9], f, relu=False)
        c1 = new(w[10].shape[0])
        _linear(f, w[10], b[10], c1, relu=False)
        _linear(posd, w[10], None, c1, relu=True, w_col0=H, accumulate=True)
        _linear(c1, w[11], b[11], out, relu=False)                                   # rgb -> out[:, 0:3]
        if keep:
            ctx.save_for_backward(posx, posd, f, c1, *acts, *w)
        ctx.shapes = (P, H)
        return out

    @staticmethod
    def backward(ctx, g_out):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("Nerf.forward: gradients with respect to the input points are not provided")
        saved = ctx.saved_tensors
        posx, posd, f, c1 = saved[:4]
        acts, w = saved[4:12], saved[12:]
        P, H = ctx.shapes
        dev = g_out.device
        g = _lib.require_cuda_f32(g_out, "grad").contiguous()
        one = torch.ones(1, dtype=torch.float32, device=dev)
        # all 24 gradients as consecutive views of ONE flat vector in state_dict order, like the fused backward's: the
        # data-parallel exchange is one all-reduce of it and FusedAdam reads it in place (parallel.flat_grad_view)
        flat = torch.zeros(sum(x.numel() + x.shape[0] for x in w), dtype=torch.float32, device=dev)
        gw, gb, off = [], [], 0
        for x in w:
            gw.append(flat[off:off + x.numel()].view(x.shape))
            off += x.numel()
            gb.append(flat[off:off + x.shape[0]])
            off += x.shape[0]
        new = lambda n: torch.empty((P, n), dtype=torch.float32, device=dev)         # noqa: E731

        def wgrad(L, dy, ld_dy, n_out, x, *, dy_off=0, mask=None, w_col0=0):
            """gw[L][:, w_col0 : w_col0 + x.shape[1]] += (dy * relu')^T x ;  gb[L] += (dy * relu')^T 1 (once per layer)."""
            K_in = x.shape[1]
            _gemm(dy, 1, ld_dy, x, K_in, 1, gw[L], gw[L].shape[1], n_out, K_in, P, mask=mask, a_off=dy_off,
                  c_off=w_col0, flags=ACC)
            if w_col0 == 0:
                _gemm(dy, 1, ld_dy, one, 0, 0, gb[L], 1, n_out, 1, P, mask=mask, a_off=dy_off, flags=ACC)

        def xgrad(dy, ld_dy, n_out, wL, dx, *, dy_off=0, mask=None, w_col0=0, accumulate=False):
            """dx (+)= (dy * relu') @ wL[:, w_col0 : w_col0 + dx.shape[1]]"""
            _gemm(dy, ld_dy, 1, wL, wL.shape[1], 1, dx, dx.shape[1], P, dx.shape[1], n_out, mask=mask, a_off=dy_off,
                  b_off=w_col0, flags=ACC if accumulate else 0)

        Hc = c1.shape[1]
        # colour head: rgb = c1 @ W11^T + b11 ; c1 = relu(cat(f, posd) @ W10^T + b10)
        wgrad(11, g, 4, 3, c1)
        d_c1 = new(Hc)
        xgrad(g, 4, 3, w[11], d_c1)
        wgrad(10, d_c1, Hc, Hc, f, mask=c1)
        wgrad(10, d_c1, Hc, Hc, posd, mask=c1, w_col0=H)
        d_f = new(H)
        xgrad(d_c1, Hc, Hc, w[10], d_f, mask=c1)
        # f = h7 @ W9^T + b9 (no ReLU) ; sigma = h7 @ W8^T + b8
        h7 = acts[7]
        wgrad(9, d_f, H, H, h7)
        wgrad(8, g, 4, 1, h7, dy_off=3)
        d_h = new(H)
        xgrad(d_f, H, H, w[9], d_h)
        xgrad(g, 4, 1, w[8], d_h, dy_off=3, accumulate=True)
        # layers_1 (7, 6), the skip layer (5), layers_0 (4 .. 0): each d_h is the gradient of acts[L] before its ReLU mask
        for L in (7, 6):
            x = acts[L - 1]
            wgrad(L, d_h, H, H, x, mask=acts[L])
            d_prev = new(H)
            xgrad(d_h, H, H, w[L], d_prev, mask=acts[L])
            d_h = d_prev
        wgrad(5, d_h, H, H, acts[4], mask=acts[5])
        wgrad(5, d_h, H, H, posx, mask=acts[5], w_col0=H)
        d_prev = new(H)
        xgrad(d_h, H, H, w[5], d_prev, mask=acts[5])
        d_h = d_prev
        for L in (4, 3, 2, 1):
            wgrad(L, d_h, H, H, acts[L - 1], mask=acts[L])
            d_prev = new(H)
            xgrad(d_h, H, H, w[L], d_prev, mask=acts[L])
            d_h = d_prev
        wgrad(0, d_h, H, H, posx, mask=acts[0])
        grads = []
        for L in range(12):
            grads += [gw[L], gb[L]]
        return (None, None, None, *grads)


INFERENCE_CHUNK = 1 << 20      # points per pass without gradients: bounds the activations in HBM (~ 6 x H x 4 B per point live)


def forward(net, v):
    """Nerf.forward(v) for a module of any (Lp, Ld, H): [P, 6] -> [P, 4] fp32, with gradients for the parameters."""
    params = []
    mods = dict(net.named_modules())
    for name in LAYER_NAMES:
        m = mods[name]
        params += [m.weight, m.bias]
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if needs_grad or v.shape[0] <= INFERENCE_CHUNK:
        return _GenericMlp.apply(v, net.Lp, net.Ld, *params)
    # inference on many points (an image's worth of samples): chunk by chunk, nothing kept between chunks
    out = torch.empty((v.shape[0], 4), dtype=torch.float32, device=v.device)
    for s in range(0, v.shape[0], INFERENCE_CHUNK):
        out[s:s + INFERENCE_CHUNK] = _GenericMlp.apply(v[s:s + INFERENCE_CHUNK], net.Lp, net.Ld, *params)
    return out
