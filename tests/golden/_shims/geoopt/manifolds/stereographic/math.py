"""Stand-in for ``geoopt.manifolds.stereographic.math`` (fixture generation ONLY).

geoopt is an un-vendored, un-pinned third-party dependency of the reference
(requirements.txt:13 ``geoopt``; imported at core/utils/hyperbolic.py:8) and is
not installed in this image.  This module restates, in plain torch, the five
functions the reference's hot path calls -- ``expmap0``, ``project``,
``logmap0``, ``dist0``, ``dist`` -- from geoopt's published formulas
(geoopt >= 0.3, ``stereographic/math.py``):

* ``sabs(x) = |x| + 1e-15``
* ``tanh(x) = x.clamp(-15, 15).tanh()``
* ``artanh(x) = 0.5 * (log(1 + z) - log(1 - z))``, ``z = x.clamp(-1+1e-7, 1-1e-7)``,
  evaluated in the INPUT dtype
* ``tan_k / artan_k`` for k < 0: ``tanh(x*sqrt(sabs(k))) / sqrt(sabs(k))`` and
  ``artanh(x*sqrt(sabs(k))) / sqrt(sabs(k))``
* ``project``: eps = 4e-3 (float32) / 1e-5 (float64)
* ``mobius_add`` denominator ``clamp_min(1e-15)``

It exists so that ``tests/golden/make_fixtures.py`` can import and RUN the
reference's own Python (core/utils/hyperbolic.py, core/active/*.py) in this
container.  Because this file, not geoopt itself, supplied the arithmetic, the
geoopt layer of the fixtures is "parity unpinned" (see DESIGN.md); the
closed-form known-answer tests in tests/test_oracle_kat.py pin that layer
independently with mpmath.

``artanh`` and float32: rounds 1-3 of this stand-in took the two logarithms in float64 and cast back --
the form of geoopt's OLDER ``poincare/math.py`` ``Artanh`` autograd function.  ``stereographic/math.py`` (the module the
reference imports) reads ``x = x.clamp(-1 + 1e-7, 1 - 1e-7); return (torch.log(1 + x).sub(torch.log(1 - x))).mul(0.5)``:
plain ``log`` (not ``log1p``) in the input dtype.  Round 4 restates it that way here, in the oracle
(oracle/halo_oracle.c:dist0_from_ssq_f32) and in the device recipe (halo_devmath.hpp:dist0_from_ssq(float)).  For float64
inputs (the reference's HYPER=True path) nothing changes.  For float32 inputs the rounding of ``1 + z`` / ``1 - z`` to
float32 is part of the convention: it moves a radius by up to ~1.2e-7 ABSOLUTE against the float64-log form (a large
relative change only for radii << 1e-3).  tests/test_oracle_kat.py brackets both forms (dist0 and logmap0).

Only the k < 0 (Poincare ball) branch is needed: HyperMapper always passes
``k = tensor(-c)`` with c > 0 (hyperbolic.py:26).
"""
import torch


def sabs(x, eps: float = 1e-15):
    return x.abs().add(eps)


def tanh(x):
    return x.clamp(-15, 15).tanh()


def artanh(x):
    x = x.clamp(-1 + 1e-7, 1 - 1e-7)
    return (torch.log(1 + x).sub(torch.log(1 - x))).mul(0.5)


def _k_sqrt(k):
    assert bool(torch.all(k < 0)), "stand-in implements the Poincare-ball branch only"
    return sabs(k).sqrt()


def tan_k(x, k):
    ks = _k_sqrt(k)
    return ks.reciprocal() * tanh(x * ks)


def artan_k(x, k):
    ks = _k_sqrt(k)
    return ks.reciprocal() * artanh(x * ks)


def expmap0(u, *, k, dim=-1):
    u_norm = u.norm(dim=dim, p=2, keepdim=True).clamp_min(1e-15)
    return tan_k(u_norm, k) * (u / u_norm)


def project(x, *, k, dim=-1, eps=-1.0):
    if eps < 0:
        eps = 4e-3 if x.dtype == torch.float32 else 1e-5
    maxnorm = (1 - eps) / (sabs(k) ** 0.5)
    maxnorm = torch.where(k.lt(0), maxnorm, k.new_full((), 1e15))
    norm = x.norm(dim=dim, keepdim=True, p=2).clamp_min(1e-15)
    cond = norm > maxnorm
    projected = x / norm * maxnorm
    return torch.where(cond, projected, x)


def logmap0(y, *, k, dim=-1):
    y_norm = y.norm(dim=dim, p=2, keepdim=True).clamp_min(1e-15)
    return (y / y_norm) * artan_k(y_norm, k)


def dist0(x, *, k, keepdim=False, dim=-1):
    return 2.0 * artan_k(x.norm(dim=dim, p=2, keepdim=keepdim), k)


def mobius_add(x, y, *, k, dim=-1):
    x2 = x.pow(2).sum(dim=dim, keepdim=True)
    y2 = y.pow(2).sum(dim=dim, keepdim=True)
    xy = (x * y).sum(dim=dim, keepdim=True)
    num = (1 - 2 * k * xy - k * y2) * x + (1 + k * x2) * y
    denom = 1 - 2 * k * xy + k ** 2 * x2 * y2
    return num / denom.clamp_min(1e-15)


def dist(x, y, *, k, keepdim=False, dim=-1):
    return 2.0 * artan_k(
        mobius_add(-x, y, k=k, dim=dim).norm(dim=dim, p=2, keepdim=keepdim), k
    )
