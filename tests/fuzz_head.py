"""Randomised differential test of the head-side entry points on one MI355X against the CPU oracle: expmap / logmap / dist0 / pdist
(bit for bit where the two share one written sequence of operations, a stated tolerance where the device uses the matrix cores or
the library's tanh), HyperMLR logits (float64 and float32 output, class counts on both sides of the MFMA path's limit of 32),
bilinear resize (float32 / float64, up- and down-sampling, degenerate sizes).  `python tests/fuzz_head.py [n_cases] [seed]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import halo_amd  # noqa: F401
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners
from oracle import halo_oracle as ho

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(SEED)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and (np.array_equal(a, b, equal_nan=True))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    m = np.isfinite(a) & np.isfinite(b)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return np.inf
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(1e-300, np.maximum(np.abs(a[m]), np.abs(b[m])))) if m.any() else 0.0)


worst = {}
worst_mlr_case = None
t0 = time.time()
for i in range(N):
    c = float(rng.choice([1.0, 1.0, 0.5, 2.0, 0.1]))
    m = HyperMapper(c=c)
    B, C = int(rng.integers(1, 3)), int(rng.choice([1, 2, 7, 16, 33, 64, 130]))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
    scale = float(rng.choice([1e-3, 0.05, 0.4, 3.0, 60.0]))
    z = (rng.standard_normal((B, C, h, w)) * scale).astype(np.float32)
    if rng.random() < 0.3:
        z[0, :, 0, 0] = 0.0
    desc = dict(i=i, c=c, B=B, C=C, h=h, w=w, scale=scale)
    # expmap over dim 1 (the head's call) and over the last dim
    e_dev = m.expmap(t(z), dim=1).cpu().numpy()
    e_or = ho.expmap(z, c, dim=1)
    r = rel(e_dev, e_or)
    worst["expmap dim=1"] = max(worst.get("expmap dim=1", 0.0), r)
    if r > 1e-14:
        print("MISMATCH expmap", r, desc); sys.exit(1)
    zl = z.reshape(B * C * h, w) if w > 1 else z.reshape(-1, C)
    r = rel(m.expmap(t(zl)).cpu().numpy(), ho.expmap(zl, c, dim=-1))
    if r > 1e-14:
        print("MISMATCH expmap lastdim", r, desc); sys.exit(1)
    # dist0: float64 and float32, dim 1 -- one written fma chain: bit for bit
    for arr in (e_or, e_or.astype(np.float32)):
        d_dev = m.poincare_distance_origin(t(arr), dim=1).cpu().numpy()
        if not bits(d_dev, ho.dist0(arr, c, dim=1)):
            print("MISMATCH dist0", arr.dtype, desc); sys.exit(1)
    # logmap / pdist over the last dim (float64)
    pts = np.ascontiguousarray(np.moveaxis(e_or, 1, -1).reshape(-1, C))
    r = rel(m.logmap(t(pts)).cpu().numpy(), ho.logmap(pts, c))
    worst["logmap"] = max(worst.get("logmap", 0.0), r)
    if r > 1e-12:
        print("MISMATCH logmap", r, desc); sys.exit(1)
    q = ho.expmap((rng.standard_normal(pts.shape) * 0.5).astype(np.float32), c, dim=-1)
    dd, do = m.poincare_distance(t(pts), t(q)).cpu().numpy(), ho.dist(pts, q, c)
    ok = np.isfinite(do)
    near = np.max(np.linalg.norm(pts, axis=1)) * np.sqrt(c) > 1 - 1e-4 or np.max(np.linalg.norm(q, axis=1)) * np.sqrt(c) > 1 - 1e-4
    r = rel(dd[ok], do[ok])
    worst["pdist"] = max(worst.get("pdist", 0.0), 0.0 if near else r)
    if r > (1e-5 if near else 1e-9):          # next to the boundary artanh amplifies rounding by 1 / (1 - z^2)
        print("MISMATCH pdist", r, desc); sys.exit(1)
    # HyperMLR logits
    O = int(rng.choice([2, 16, 19, 19, 32, 33, 40]))
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    mlr = HyperMLR(C, O, c=c).to(dev)
    P, A = mlr.P_MLR.detach().cpu().numpy(), mlr.A_MLR.detach().cpu().numpy()
    want = ho.hypermlr(e_or, P, A, c)
    with torch.no_grad():
        got64 = mlr._hyper_logits(t(e_or)).cpu().numpy()
        got32 = mlr._hyper_logits(t(e_or), out_dtype=torch.float32).cpu().numpy()
    r = float(np.max(np.abs(got64 - want) / np.maximum(1.0, np.abs(want))))
    if r > worst.get("hypermlr f64", 0.0):
        worst_mlr_case = dict(desc, O=O, r=r)
    worst["hypermlr f64"] = max(worst.get("hypermlr f64", 0.0), r)
    if not np.array_equal(np.isnan(got64), np.isnan(want)):
        print("MISMATCH hypermlr (NaN pattern)", dict(desc, O=O)); sys.exit(1)
    if r > 1e-9:
        # Is it the conditioning of the case (the Moebius denominator D = 1 + 2K<x,-p> + K^2 |x|^2 |p|^2 near zero amplifies ANY rounding:
        # the matrix-core contraction's summation order against the oracle's sequential chain is enough) or the one-quotient epilogue?
        # The device's REFERENCE-ORDER epilogue (HALO_MLR_EPI_REF=1) on the same contraction decides: equally far -> conditioning, tallied.
        os.environ["HALO_MLR_EPI_REF"] = "1"
        with torch.no_grad():
            got_ref = mlr._hyper_logits(t(e_or)).cpu().numpy()
        del os.environ["HALO_MLR_EPI_REF"]
        r_ref = float(np.max(np.abs(got_ref - want) / np.maximum(1.0, np.abs(want))))
        xx = (e_or ** 2).sum(axis=1)                                                        # (B, h, w)
        px = np.einsum("bchw,oc->bohw", e_or, -P)
        D = 1.0 + 2.0 * c * px + (c * c) * xx[:, None] * (P ** 2).sum(axis=1)[None, :, None, None]
        print("ill-conditioned hypermlr case: one-quotient %.2e, reference-order %.2e from the oracle; min |D| %.2e" % (r, r_ref, float(np.abs(D).min())), dict(desc, O=O))
        if r > 1e-5 or r_ref < 0.1 * r:
            print("MISMATCH hypermlr", r, dict(desc, O=O)); sys.exit(1)
        worst["hypermlr ill-conditioned cases"] = worst.get("hypermlr ill-conditioned cases", 0) + 1
    if got32.dtype != np.float32 or np.max(np.abs(got32 - want.astype(np.float32)) / np.maximum(1.0, np.abs(want))) > 2e-6:
        print("MISMATCH hypermlr f32 output", dict(desc, O=O)); sys.exit(1)
    # bilinear resize: one written tap order, bit for bit
    H2, W2 = int(rng.integers(1, 90)), int(rng.integers(1, 150))
    for arr in (z, e_or):
        up = bilinear_align_corners(t(arr), (H2, W2)).cpu().numpy()
        if not bits(up, ho.bilinear(arr, (H2, W2))):
            print("MISMATCH bilinear", arr.dtype, dict(desc, H2=H2, W2=W2)); sys.exit(1)
    if i % 25 == 0:
        print("case %d ok %s" % (i, dict(desc, O=O, H2=H2, W2=W2)), flush=True)
# the fused native HyperMLR backward (halo_hypermlr_backward: <= 20 classes, 64 | C <= 256) against the term-map path it replaces
from halo_amd.core.utils.hyperbolic import _HyperMLRFn
for i in range(max(1, N // 10)):
    c = float(rng.choice([1.0, 1.0, 0.5, 2.0, 0.1]))
    B, C, O = int(rng.integers(1, 4)), int(rng.choice([64, 64, 128, 192, 256])), int(rng.integers(1, 21))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
    scale = float(rng.choice([1e-3, 0.05, 0.2, 1.0, 30.0]))
    z = (rng.standard_normal((B, C, h, w)) * scale).astype(np.float32)
    if rng.random() < 0.3:
        z[0, :, 0, 0] = 0.0
    desc = dict(i=i, c=c, B=B, C=C, O=O, h=h, w=w, scale=scale)
    x0 = HyperMapper(c).expmap(t(z), dim=1).double()
    bound = 1.0 / np.sqrt(C)
    P0, A0, Wt = t(rng.uniform(-bound, bound, (O, C))), t(rng.uniform(-bound, bound, (O, C))), t(rng.standard_normal((B, O, h, w)))
    res = []
    for env in (None, "1"):
        if env:
            os.environ["HALO_MLR_BWD_TERMS"] = env
        else:
            os.environ.pop("HALO_MLR_BWD_TERMS", None)
        xg, Pg, Ag = x0.clone().requires_grad_(True), P0.clone().requires_grad_(True), A0.clone().requires_grad_(True)
        (_HyperMLRFn.apply(xg, Pg, Ag, c) * Wt).sum().backward()
        res.append([g.grad.cpu().numpy() for g in (xg, Pg, Ag)])
    os.environ.pop("HALO_MLR_BWD_TERMS", None)
    for name, a_, b_ in zip(("gx", "gP", "gA"), res[0], res[1]):
        r = float(np.abs(a_ - b_).max() / (np.abs(b_).max() + 1e-300))
        worst["mlr backward " + name] = max(worst.get("mlr backward " + name, 0.0), r)
        if not np.isfinite(a_).all() or r > 1e-10:
            print("MISMATCH fused backward", name, r, desc); sys.exit(1)
print("worst hypermlr case:", worst_mlr_case)
print("fuzz_head: %d cases, seed %d, %.0f s: dist0 / bilinear bit for bit; worst relative differences %s" %
      (N, SEED, time.time() - t0, {k: float("%.2g" % v) for k, v in worst.items()}))
