"""HyperMapper / HyperMLR on HIP kernels -- host mirror of core/utils/hyperbolic.py.

Same class names, method names, argument meaning, parameter names (`P_MLR`, `A_MLR`, float64,
kaiming-uniform(a=sqrt(5)) -- checkpoint keys `classifier.conv_seg.P_MLR/A_MLR` load unchanged)
and dtypes as the reference.  The arithmetic (geoopt's stereographic math in the reference,
hyperbolic.py:8) runs in halo_amd/csrc/halo_hyperbolic.hip.

Autograd: `HyperMapper.expmap` and `HyperMLR.forward` -- the two ops of the head
tail the training step differentiates (core/models/classifier.py:553-554) -- are
torch.autograd.Functions with HIP backward kernels; logmap and the two distances (no caller
differentiates them) keep their HIP forward and get their backward from torch autograd over a
device-side statement of the same formulas; bilinear_align_corners is inference-only and raises
if a gradient is requested instead of silently detaching.
"""
import math
import os

import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

from ... import _lib

PROJ_EPS = 1e-3


def _no_grad_only(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "this halo_amd op (bilinear_align_corners) is inference-only: the training path resizes with F.interpolate "
            "(core/models/classifier.py); wrap the call in torch.no_grad()")


# ---- geoopt's formulas in device-side torch arithmetic, for the BACKWARD of the three ops no caller of the reference
# differentiates (logmap, poincare_distance, poincare_distance_origin: hyperbolic.py:51-83).  Their forward values always come
# from the HIP kernels; only when a gradient is requested does autograd run through this restatement (geoopt >= 0.3
# stereographic/math.py: sabs + 1e-15, artanh clamp 1 - 1e-7, norm clamp_min 1e-15, project eps 1e-5 / 4e-3).
def _t_artan_k(x, c):
    ks = math.sqrt(abs(-c) + 1e-15)
    z = (x * ks).clamp(-1 + 1e-7, 1 - 1e-7)
    return (torch.log1p(z) - torch.log1p(-z)) * (0.5 / ks)


def _t_project(x, c):
    eps = 4e-3 if x.dtype == torch.float32 else 1e-5
    maxnorm = (1 - eps) / math.sqrt(abs(-c) + 1e-15)
    norm = x.norm(dim=-1, keepdim=True, p=2).clamp_min(1e-15)
    return torch.where(norm > maxnorm, x / norm * maxnorm, x)


def _t_logmap(x, c):
    y = x.double()
    yn = y.norm(dim=-1, p=2, keepdim=True).clamp_min(1e-15)
    return _t_project((y / yn) * _t_artan_k(yn, c), c)


def _t_dist0(x, c, dim):
    return 2.0 * _t_artan_k(x.norm(dim=dim, p=2), c)


def _t_dist(x, y, c):
    k = -c
    a = -x
    x2 = a.pow(2).sum(dim=-1, keepdim=True)
    y2 = y.pow(2).sum(dim=-1, keepdim=True)
    xy = (a * y).sum(dim=-1, keepdim=True)
    num = (1 - 2 * k * xy - k * y2) * a + (1 + k * x2) * y
    den = (1 - 2 * k * xy + k ** 2 * x2 * y2).clamp_min(1e-15)
    return 2.0 * _t_artan_k((num / den).norm(dim=-1, p=2), c)


class _HipForwardTorchBackward(torch.autograd.Function):
    """forward(*tensors) on a HIP kernel; backward through `restate(*tensors)`, the same formula in torch ops on the device."""

    @staticmethod
    def forward(ctx, fwd, restate, *tensors):
        ctx.restate = restate
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            return fwd(*tensors)

    @staticmethod
    def backward(ctx, gout):
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(ctx.needs_input_grad[2 + i]) for i, t in enumerate(ctx.saved_tensors)]
            out = ctx.restate(*ins)
            want = [t for t in ins if t.requires_grad]
            grads = iter(torch.autograd.grad(out, want, gout.to(out.dtype), allow_unused=True))
        return (None, None) + tuple(next(grads) if t.requires_grad else None for t in ins)


def _differentiable(fwd, restate, *tensors):
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return _HipForwardTorchBackward.apply(fwd, restate, *tensors)
    return fwd(*tensors)


def _split(shape, dim):
    dim = dim % len(shape)
    outer = 1
    for s in shape[:dim]:
        outer *= s
    inner = 1
    for s in shape[dim + 1:]:
        inner *= s
    return dim, outer, shape[dim], inner


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _expmap_forward(x, c, dim):
    dev = _lib.require_device(x)
    x = x.contiguous()
    if x.dtype not in (torch.float32, torch.float64):
        x = x.double()
    y = torch.empty(x.shape, dtype=torch.float64, device=dev)
    if x.numel():
        _, outer, C, inner = _split(x.shape, dim)
        _lib.check(_lib.lib().halo_expmap0_project(_lib.ptr(x), _lib.dtype_code(x), _lib.ptr(y), outer, C, inner,
                                                   float(c), _lib.stream_ptr(dev)), "halo_expmap0_project")
    return x, y


class _ExpmapFn(torch.autograd.Function):
    """y = project(expmap0(x.double()))  (hyperbolic.py:37-38) with the HIP Jacobian-transpose product."""

    @staticmethod
    def forward(ctx, x, c, dim):
        xc, y = _expmap_forward(x.detach(), c, dim)
        ctx.save_for_backward(xc)
        ctx.c, ctx.dim, ctx.in_dtype = c, dim, x.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        (xc,) = ctx.saved_tensors
        dev = xc.device
        gy = gy.double().contiguous()
        gx = torch.empty_like(xc)
        if xc.numel():
            _, outer, C, inner = _split(xc.shape, ctx.dim)
            _lib.check(_lib.lib().halo_expmap0_project_bwd(_lib.ptr(xc), _lib.dtype_code(xc), _lib.ptr(gy), _lib.ptr(gx),
                                                           outer, C, inner, float(ctx.c), _lib.stream_ptr(dev)),
                       "halo_expmap0_project_bwd")
        return gx.to(ctx.in_dtype), None, None


def _mlr_forward(x, P, A, c, out_dtype):
    dev = _lib.require_device(x, P, A)
    B, Cc, H, W = x.shape
    O = P.shape[0]
    out = torch.empty((B, O, H, W), dtype=out_dtype, device=dev)
    if out.numel():
        L = _lib.lib()
        nws = L.halo_hypermlr_workspace_bytes(O, Cc)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(L.halo_hypermlr_logits(_lib.ptr(x), _lib.ptr(P), _lib.ptr(A), _lib.ptr(out), _lib.dtype_code(out),
                                          B, Cc, O, H * W, float(c), _lib.ptr(ws), nws, _lib.stream_ptr(dev)),
                   "halo_hypermlr_logits")
    return out


def head_tail_fused(feat, P, A, c, out_dtype=torch.float32):
    """embed = expmap(feat, dim=1) (float64) and out = HyperMLR(embed) in ONE kernel (halo_head_tail: the heads' own 64 channels,
    <= 32 classes, an even pixel count) -- or None when the shape is not served; the caller then makes the two calls, whose results
    the fused kernel reproduces bit for bit.  Inference only (no autograd)."""
    dev = _lib.require_device(feat, P, A)
    if feat.dim() != 4 or feat.dtype != torch.float32 or feat.shape[1] != 64 or feat.numel() == 0:
        return None
    feat = feat.contiguous()
    B, Cc, H, W = feat.shape
    O = P.shape[0]
    embed = torch.empty((B, Cc, H, W), dtype=torch.float64, device=dev)
    out = torch.empty((B, O, H, W), dtype=out_dtype, device=dev)
    L = _lib.lib()
    nws = L.halo_hypermlr_workspace_bytes(O, Cc)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    rc = L.halo_head_tail(_lib.ptr(feat), _lib.ptr(P.detach().contiguous()), _lib.ptr(A.detach().contiguous()), _lib.ptr(embed), _lib.ptr(out),
                          _lib.dtype_code(out), B, Cc, O, H * W, float(c), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
    if rc == 1:
        return None
    _lib.check(rc, "halo_head_tail")
    return out, embed


def _pixel_contraction(D, X, chunk=512):
    """sum over batch and pixels of D[b,:,n] X[b,:,n]^T  ->  (rows(D), rows(X)).  The output is tiny (2O x C)
    and the contraction very long (B*H*W): as ONE GEMM the BLAS library runs it on a handful of workgroups
    (10.9 ms for 2x51200 pixels); split over pixel chunks it is a batched GEMM plus a fixed-order sum."""
    B, J, N = D.shape
    Cc = X.shape[1]
    pad = (-N) % chunk
    if pad:
        D = torch.nn.functional.pad(D, (0, pad))
        X = torch.nn.functional.pad(X, (0, pad))
    S = (N + pad) // chunk
    Dk = D.reshape(B, J, S, chunk).permute(0, 2, 1, 3).reshape(B * S, J, chunk)
    Xk = X.reshape(B, Cc, S, chunk).permute(0, 2, 3, 1).reshape(B * S, chunk, Cc)
    return torch.bmm(Dk, Xk).sum(dim=0)


class _HyperMLRFn(torch.autograd.Function):
    """HyperMLR._hyper_logits (hyperbolic.py:120-184), float64, with gradients for x, P_MLR and A_MLR.

    backward, at the heads' shapes (<= 20 classes, 64 | C <= 256): ONE native call (halo_hypermlr_backward: the reverse sweep
    through the Moebius / projection / asinh algebra per pixel and class, d x = W^T D + 2 x dxx, d W = D x^T and the
    parameter algebra: prep + three kernels, fixed summation order).  Any other shape (and HALO_MLR_BWD_TERMS=1, the cross-check):
    one HIP kernel for the reverse sweep + the two dense contractions as library GEMMs (torch.einsum -> rocBLAS),
        W = [-P ; A/||A||],  D = [dpx ; dxa]."""

    @staticmethod
    def forward(ctx, x, P, A, c, out_dtype=torch.float64):
        # out_dtype=float32: the head's `.float()` (classifier.py:554) fused into the forward kernel's store; the backward then
        # receives a float32 gradient and the native call reads it as it is (no cast kernels either way)
        xd = x.detach().double().contiguous()
        Pd, Ad = P.detach().contiguous(), A.detach().contiguous()
        ctx.save_for_backward(xd, Pd, Ad)
        ctx.c = c
        return _mlr_forward(xd, Pd, Ad, c, out_dtype)

    @staticmethod
    def backward(ctx, gout):
        x, P, A = ctx.saved_tensors
        dev = x.device
        B, Cc, H, W = x.shape
        O, hw = P.shape[0], H * W
        L = _lib.lib()
        nfused = 0 if os.environ.get("HALO_MLR_BWD_TERMS") else L.halo_hypermlr_backward_workspace_bytes(B, Cc, O, hw)
        gout = (gout if (nfused and gout.dtype == torch.float32) else gout.double()).contiguous()
        if nfused:
            # the heads' shapes (<= 20 classes, 64 | C <= 256): the whole backward on the device in one call
            gx = torch.empty((B, Cc, H, W), dtype=torch.float64, device=dev)
            gP, gA = torch.empty_like(P), torch.empty_like(A)
            ws = torch.empty(nfused, dtype=torch.uint8, device=dev)
            _lib.check(L.halo_hypermlr_backward(_lib.ptr(x), _lib.ptr(P), _lib.ptr(A), _lib.ptr(gout), _lib.dtype_code(gout), B, Cc, O, hw, float(ctx.c),
                                                _lib.ptr(gx), _lib.ptr(gP), _lib.ptr(gA), _lib.ptr(ws), nfused, _lib.stream_ptr(dev)),
                       "halo_hypermlr_backward")
            return gx, gP, gA, None, None
        terms = torch.empty((5, B, O, hw), dtype=torch.float64, device=dev)       # dpx, dxa, dpp, dpa, dan
        dxx = torch.empty((B, hw), dtype=torch.float64, device=dev)
        nws = L.halo_hypermlr_workspace_bytes(O, Cc)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(L.halo_hypermlr_bwd_terms(_lib.ptr(x), _lib.ptr(P), _lib.ptr(A), _lib.ptr(gout), B, Cc, O, hw,
                                             float(ctx.c), _lib.ptr(terms[0]), _lib.ptr(terms[1]), _lib.ptr(dxx),
                                             _lib.ptr(terms[2]), _lib.ptr(terms[3]), _lib.ptr(terms[4]), _lib.ptr(ws), nws,
                                             _lib.stream_ptr(dev)), "halo_hypermlr_bwd_terms")
        dpx, dxa = terms[0], terms[1]
        dpp, dpa, dan = (terms[k].sum(dim=(0, 2)) for k in (2, 3, 4))            # (O,)
        a_norm = A.norm(dim=1)                                                    # hyperbolic.py:172
        dn = a_norm.clamp_min(1e-12)                                              # F.normalize eps, :173
        An = A / dn[:, None]
        xf = x.reshape(B, Cc, hw)
        gx = torch.einsum("oc,bon->bcn", -P, dpx) + torch.einsum("oc,bon->bcn", An, dxa) + 2.0 * xf * dxx[:, None, :]
        gW = _pixel_contraction(torch.cat([dpx, dxa], dim=1), xf)                 # (2O, C) = sum_{b,n} D[b,:,n] x[b,:,n]^T
        g_negP = gW[:O]                                                           # d L / d (-P) through px
        g_An = gW[O:] + dpa[:, None] * (-P)                                       # through xa and pa = <-P, An>
        gP = -g_negP + dpp[:, None] * (2.0 * P) - dpa[:, None] * An               # pp = ||P||^2
        gA = (g_An - (g_An * An).sum(dim=1, keepdim=True) * An) / dn[:, None] + dan[:, None] * A / a_norm[:, None]
        return gx.reshape(B, Cc, H, W), gP, gA, None, None


class HyperMapper(object):
    """Maps between Euclidean and hyperbolic space and computes distances (hyperbolic.py:16-97)."""

    def __init__(self, c=1.) -> None:
        self.c = c
        self.K = torch.tensor(-self.c, dtype=float)

    def expmap(self, x, dim=-1):
        """project(expmap0(x.double())) -> float64 (hyperbolic.py:28-39); differentiable."""
        if _needs_grad(x):
            _lib.require_device(x)
            return _ExpmapFn.apply(x, float(self.c), dim)
        return _expmap_forward(x, self.c, dim)[1]

    def expmap2(self, inputs, dim=-1):
        """Alternative expmap with +1e-15 and eps 1e-3 (hyperbolic.py:41-49; no caller in-tree).
        Elementwise torch ops on the device; only the projection's norm is a reduction."""
        dev = _lib.require_device(inputs)
        sqrt_c = torch.sqrt(torch.abs(self.K)).to(dev)
        inputs = inputs + 1e-15
        norm = torch.norm(inputs, dim=dim)
        gamma = torch.tanh(sqrt_c * norm) / (sqrt_c * norm)
        scaled = gamma.unsqueeze(dim) * inputs
        maxnorm = (1 - PROJ_EPS) / ((self.K.abs() + 1e-15) ** 0.5).to(dev)
        n = scaled.norm(dim=dim, keepdim=True, p=2).clamp_min(1e-15)
        return torch.where(n > maxnorm, scaled / n * maxnorm, scaled)

    def logmap(self, x):
        """project(logmap0(x.double())) over the last dim (hyperbolic.py:51-60).  Differentiable (no caller of the reference
        asks: the backward runs through device-side torch autograd of the same formula)."""
        return _differentiable(self._logmap_fwd, lambda t: _t_logmap(t, self.c), x)

    def _logmap_fwd(self, x):
        dev = _lib.require_device(x)
        x = x.double().contiguous()
        y = torch.empty_like(x)
        if x.numel() == 0:
            return y
        _, outer, C, inner = _split(x.shape, -1)
        _lib.check(_lib.lib().halo_logmap0_project(_lib.ptr(x), _lib.ptr(y), outer, C, inner, float(self.c),
                                                   _lib.stream_ptr(dev)), "halo_logmap0_project")
        return y

    def poincare_distance(self, x, y):
        """geoopt dist over the last dim (hyperbolic.py:62-72; no live caller in-tree).  Differentiable like logmap."""
        return _differentiable(self._pdist_fwd, lambda a, b: _t_dist(*torch.broadcast_tensors(a, b), self.c), x, y)

    def _pdist_fwd(self, x, y):
        dev = _lib.require_device(x, y)
        in_dtype = torch.promote_types(x.dtype, y.dtype)
        x, y = torch.broadcast_tensors(x, y)
        x = x.double().contiguous()
        y = y.double().contiguous()
        out = torch.empty(x.shape[:-1], dtype=torch.float64, device=dev)
        if out.numel():
            _lib.check(_lib.lib().halo_pdist(_lib.ptr(x), _lib.ptr(y), _lib.ptr(out), out.numel(), x.shape[-1],
                                             float(self.c), _lib.stream_ptr(dev)), "halo_pdist")
        return out if in_dtype == torch.float64 else out.to(in_dtype)

    def poincare_distance_origin(self, x, dim=-1):
        """geoopt dist0: 2/sqrt(c) artanh(sqrt(c)||x||), dtype preserved (hyperbolic.py:74-83).  Differentiable like logmap."""
        return _differentiable(lambda t: self._dist0_fwd(t, dim), lambda t: _t_dist0(t, self.c, dim), x)

    def _dist0_fwd(self, x, dim=-1):
        dev = _lib.require_device(x)
        x = x.contiguous()
        d, outer, C, inner = _split(x.shape, dim)
        out = torch.empty(x.shape[:d] + x.shape[d + 1:], dtype=x.dtype, device=dev)
        if out.numel():
            _lib.check(_lib.lib().halo_dist0(_lib.ptr(x), _lib.dtype_code(x), _lib.ptr(out), outer, C, inner,
                                             float(self.c), _lib.stream_ptr(dev)), "halo_dist0")
        return out

    def cosine_distance(self, x, y):
        """2 - 2 cos (hyperbolic.py:85-97; no caller in-tree)."""
        _lib.require_device(x, y)
        x = torch.nn.functional.normalize(x, dim=-1, p=2)
        y = torch.nn.functional.normalize(y, dim=-1, p=2)
        return 2 - 2 * (x * y).sum(dim=-1)


class HyperMLR(nn.Module):
    """Multinomial logistic regression in hyperbolic space (hyperbolic.py:100-188)."""

    def __init__(self, out_channels, num_classes, c=1.):
        super().__init__()
        self.c = c
        self.K = torch.tensor(c, dtype=float)
        self.num_classes = num_classes
        self.P_MLR = Parameter(torch.empty((num_classes, out_channels), dtype=torch.double))
        self.A_MLR = Parameter(torch.empty((num_classes, out_channels), dtype=torch.double))
        kaiming_uniform_(self.P_MLR, a=math.sqrt(5))
        kaiming_uniform_(self.A_MLR, a=math.sqrt(5))

    def _hyper_logits(self, inputs, out_dtype=torch.float64):
        """inputs (B,C,H,W) float64 -> (B,O,H,W).  out_dtype=float32 fuses the head's `.float()`
        (core/models/classifier.py:373,554)."""
        _lib.require_device(inputs, self.P_MLR, self.A_MLR)
        if _needs_grad(inputs, self.P_MLR, self.A_MLR):          # training: HIP backward; float32 logits straight from the kernel
            if out_dtype in (torch.float64, torch.float32):
                return _HyperMLRFn.apply(inputs, self.P_MLR, self.A_MLR, float(self.c), out_dtype)
            return _HyperMLRFn.apply(inputs, self.P_MLR, self.A_MLR, float(self.c)).to(out_dtype)
        return _mlr_forward(inputs.double().contiguous(), self.P_MLR.detach().contiguous(),
                            self.A_MLR.detach().contiguous(), self.c, out_dtype)

    def forward(self, x):
        return self._hyper_logits(x)


class HyperMetrics(object):
    """Compute metrics for embeddings in euclidean and hyperbolic space (hyperbolic.py:191-228; no caller in the
    reference tree -- kept so that `from core.utils.hyperbolic import HyperMetrics` keeps working after install()).

    Args:
        c (float, optional): Hyperbolic curvature. Defaults to 1.0

    The exponential maps and the Poincare distance run on the HIP kernels behind HyperMapper; the remaining
    element-wise glue (mse, norms, acos) is device-side torch arithmetic on the (N, d) inputs."""

    def __init__(self, c=1.) -> None:
        self.c = c
        self.mapper = HyperMapper(c=self.c)

    def compute(self, x, y):
        """x, y (N, d) on a ROCm device -> dict of the six metrics of hyperbolic.py:202-228: mse and cosine distance of the
        Euclidean vectors; after the exponential map, the two Euclidean radii, the angle between the points' directions in degrees
        and their Poincare distance."""
        _lib.require_device(x, y)
        on_ball = [self.mapper.expmap(v) for v in (x, y)]                          # HIP: halo_expmap0_project
        radii = [torch.linalg.norm(p, dim=-1) for p in on_ball]
        directions = [p / r.unsqueeze(-1) for p, r in zip(on_ball, radii)]
        cosine = (directions[0] * directions[1]).sum(dim=-1)
        return {
            "mse": torch.nn.functional.mse_loss(x, y),
            "cosine_dist": self.mapper.cosine_distance(x, y),
            "radius_x": radii[0],
            "radius_y": radii[1],
            "ang_e": torch.acos(cosine) * 180 / math.pi,
            "poincare_dist": self.mapper.poincare_distance(*on_ball),              # HIP: halo_pdist
        }


def bilinear_align_corners(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) for float32/float64 NCHW
    (core/active/build.py:123-125,133-135; classifier.py:375-377,556-557)."""
    _no_grad_only(x)
    dev = _lib.require_device(x)
    x = x.contiguous()
    H, W = int(size[0]), int(size[1])
    h, w = x.shape[-2:]
    out = torch.empty(x.shape[:-2] + (H, W), dtype=x.dtype, device=dev)
    if out.numel():
        planes = x.numel() // (h * w)
        _lib.check(_lib.lib().halo_bilinear_upsample(_lib.ptr(x), _lib.ptr(out), _lib.dtype_code(x), planes, h, w, H, W,
                                                     _lib.stream_ptr(dev)), "halo_bilinear_upsample")
    return out
