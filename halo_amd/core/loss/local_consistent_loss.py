"""LocalConsistentLoss on HIP kernels -- host mirror of core/loss/local_consistent_loss.py:5-17
(LocalDiscrepancy + DetectSPBoundary of core/loss/boundary.py fused; no one-hot / conv tensors).

loss = mean over {boundary pixels with a valid label} of  sum_c |p - mean3x3(p)|   ('l1')
                                                      or  sum_c p*log(p/(mean3x3(p)+1e-6)+1e-6)   ('kl')
with p = softmax(x, dim=1), a replicate-padded 3x3 box mean, and the boundary = 8-neighbour Laplacian of
the label map != 0 (zero padding).
"""
import torch
import torch.nn as nn

from ... import _lib


class _LocalConsistentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label, kl):
        dev = _lib.require_device(x, label)
        xc = x.detach().float().contiguous()
        lab = label.to(torch.int64).contiguous()
        B, O, h, w = xc.shape
        need_grad = x.requires_grad
        p = torch.empty_like(xc)
        sums = torch.empty(2, dtype=torch.float64, device=dev)
        ca = torch.empty_like(xc) if need_grad else None
        cb = torch.empty_like(xc) if need_grad else None
        mask = torch.empty((B, h, w), dtype=torch.uint8, device=dev) if need_grad else None      # ca / cb hold values where mask = 1 only
        L = _lib.lib()
        nws = L.halo_loss_workspace_bytes(B * h * w)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(L.halo_local_consistent_fwd(_lib.ptr(xc), _lib.ptr(lab), B, O, h, w, 1 if kl else 0, _lib.ptr(p), _lib.ptr(sums),
                                               _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(mask), _lib.ptr(ws), nws, _lib.stream_ptr(dev)),
                   "halo_local_consistent_fwd")
        if need_grad:
            ctx.save_for_backward(p, ca, cb, mask, sums)
        ctx.in_dtype = x.dtype
        return (sums[0] / sums[1]).to(torch.float32)          # 0/0 = nan for an empty selection, like tensor[mask].mean()

    @staticmethod
    def backward(ctx, g):
        p, ca, cb, mask, sums = ctx.saved_tensors
        B, O, h, w = p.shape
        gx = torch.empty_like(p)
        g32 = g.detach().float().reshape(1).contiguous()
        _lib.check(_lib.lib().halo_local_consistent_bwd(_lib.ptr(p), _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(mask), B, O, h, w, _lib.ptr(sums),
                                                        _lib.ptr(g32), _lib.ptr(gx), _lib.stream_ptr(p.device)),
                   "halo_local_consistent_bwd")
        return gx.to(ctx.in_dtype), None, None


class LocalConsistentLoss(nn.Module):
    def __init__(self, in_channels, l_type='l1'):
        super(LocalConsistentLoss, self).__init__()
        if l_type not in ("l1", "kl"):
            raise NotImplementedError("not implemented local soft loss: {}".format(l_type))
        self.in_channels = in_channels
        self.l_type = l_type

    def forward(self, x, label):
        return _LocalConsistentFn.apply(x, label, self.l_type == "kl")
