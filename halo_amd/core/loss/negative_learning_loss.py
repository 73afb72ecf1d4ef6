"""NegativeLearningLoss on HIP kernels -- host mirror of core/loss/negative_learning_loss.py:6-16.

loss = sum(-mask * log(1 - predict + 1e-6)) / sum(mask),  mask = (predict < threshold).detach()
One fused reduction kernel forward, one element-wise kernel backward (halo_amd/csrc/halo_loss.hip).
"""
import torch
import torch.nn as nn

from ... import _lib


class _NegativeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predict, threshold):
        dev = _lib.require_device(predict)
        p = predict.detach().float().contiguous()
        sums = torch.empty(2, dtype=torch.float64, device=dev)
        L = _lib.lib()
        nws = L.halo_loss_workspace_bytes(p.numel())
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(L.halo_negative_learning_fwd(_lib.ptr(p), p.numel(), float(threshold), _lib.ptr(sums), _lib.ptr(ws), nws,
                                                _lib.stream_ptr(dev)), "halo_negative_learning_fwd")
        ctx.save_for_backward(p, sums)
        ctx.threshold, ctx.in_dtype = threshold, predict.dtype
        return (sums[0] / sums[1]).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        p, sums = ctx.saved_tensors
        gp = torch.empty_like(p)
        g32 = g.detach().float().reshape(1).contiguous()
        _lib.check(_lib.lib().halo_negative_learning_bwd(_lib.ptr(p), p.numel(), float(ctx.threshold), _lib.ptr(sums),
                                                         _lib.ptr(g32), _lib.ptr(gp), _lib.stream_ptr(p.device)),
                   "halo_negative_learning_bwd")
        return gp.to(ctx.in_dtype), None


class NegativeLearningLoss(nn.Module):
    def __init__(self, threshold=0.05):
        super(NegativeLearningLoss, self).__init__()
        self.threshold = threshold

    def forward(self, predict):
        return _NegativeFn.apply(predict, self.threshold)
