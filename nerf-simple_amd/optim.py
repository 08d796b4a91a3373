"""Fused optimizer for the training loop.

``FusedAdam(net)`` is the reference's ``torch.optim.Adam(net.parameters(), lr=5e-4)``
(train.py:43) with the same update rule and defaults, restructured for the GPU:

  * the 24 parameter tensors become views of ONE flat fp32 buffer (state_dict
    order), and so do the moments; ``step()`` is one HIP kernel over 595,844
    elements instead of a multi-tensor sweep;
  * ``step()`` ends by re-deriving the packed MFMA weight images the fused kernels
    stream (forward bf16 image, backward image) straight from the flat buffer;
  * ``param_groups[0]['lr']`` is honoured every step, so the reference's
    ``for p in optimizer.param_groups: p['lr'] *= decay`` loop (train.py:56-57) works
    unchanged.
"""
import torch

from . import _lib


class FusedAdam:
    def __init__(self, net, lr=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        self.params = [p for _, p in net.named_parameters()]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam needs the module on the GPU")
        with torch.no_grad():
            self.flat = torch.cat([p.detach().reshape(-1).float() for p in self.params]).contiguous()
            off = 0
            for p in self.params:                      # parameters become views of the flat buffer
                n = p.numel()
                p.data = self.flat[off:off + n].view(p.shape)
                off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps}]
        self.step_count = 0
        self._grad = torch.empty_like(self.flat)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def _flat_grad(self):
        # the fused backward hands out views of ONE flat vector: use it in place
        from .parallel import flat_grad_view
        flat = flat_grad_view(self.params)
        if flat is not None and flat.numel() == self.flat.numel():
            return flat
        torch.cat([p.grad.reshape(-1).float() for p in self.params], out=self._grad)
        return self._grad

    @torch.no_grad()
    def step(self):
        g = self._flat_grad()
        pg = self.param_groups[0]
        self.step_count += 1
        lib = _lib.lib()
        dev = self.flat.device
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_adam_step(
                _lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                self.flat.numel(), float(pg["lr"]), float(pg["betas"][0]), float(pg["betas"][1]),
                float(pg["eps"]), self.step_count, _lib.stream_ptr(dev)), "nerf_amd_adam_step")
        # the kernel wrote through the flat buffer (the parameters' _version did not move):
        # re-derive the packed images now and stamp the cache with the current versions
        self.net.repack_from_flat(self.flat)
