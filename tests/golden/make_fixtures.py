#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference's own Python.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_fixtures.py            # writes tests/golden/*.npz

What runs is the reference's code, imported from /root/reference:
  core/utils/hyperbolic.py   HyperMapper.expmap/logmap/poincare_distance[_origin], HyperMLR
  core/active/floating_region.py  FloatingRegionScore.forward (every branch)
  core/active/build.py       select_pixels_to_label, RegionSelection
with three stand-ins for things this image lacks (tests/golden/_shims):
  * yacs.config.CfgNode        -> attribute dict
  * geoopt.manifolds.stereographic.math -> torch restatement of geoopt's formulas
    (=> the geoopt layer is "parity unpinned"; see the shim's docstring)
  * torch.Tensor.cuda / nn.Module.cuda -> identity (floating_region.py:85-87,181-198
    hard-code .cuda(); there is no GPU in the build container)

Only DATA is written: inputs and the reference's outputs.  No reference source
is copied into the repository.
"""
import math
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HALO_REFERENCE", "/root/reference")
OUT_DIR = os.environ.get("HALO_FIXTURE_OUT", HERE)      # tests/test_fixtures_reproduce.py regenerates into a temp dir


def import_reference():
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # core/models/__init__ pulls torchvision/mmcv (absent); the hot path does not need it.
    import core.configs  # noqa: F401
    import core.utils.hyperbolic as hyp
    import core.active.floating_region as fr
    import core.active.build as ab
    return core.configs.cfg, hyp, fr, ab


def bilinear_up(x, size):
    """The reference's resize (build.py:123-125,133-135)."""
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def make_inputs(hyp, H, W, C, O, seed, sigma=0.1, scale=4, curvature=1.0):
    """Low-res latent -> reference head tail -> reference x4 upsample."""
    g = torch.Generator().manual_seed(seed)
    h, w = H // scale, W // scale
    z = torch.randn(1, C, h, w, generator=g, dtype=torch.float32) * sigma
    # a few far-out vectors so tanh clamp (+-15) and project() saturation are exercised
    z[0, :, 0, 0] *= 400.0
    z[0, :, h // 2, w // 3] *= 60.0
    z[0, :, h - 1, w - 1] = 0.0  # exact origin: norm clamp_min(1e-15) path
    mapper = hyp.HyperMapper(c=curvature)
    mlr = hyp.HyperMLR(C, O, c=curvature)
    with torch.no_grad():
        torch.manual_seed(seed + 7)
        torch.nn.init.kaiming_uniform_(mlr.P_MLR, a=math.sqrt(5))
        torch.nn.init.kaiming_uniform_(mlr.A_MLR, a=math.sqrt(5))
        embed_lr = mapper.expmap(z, dim=1)                    # classifier.py:553
        logit_lr64 = mlr(embed_lr.double())                   # classifier.py:554
        logit_lr = logit_lr64.float()
        logit = bilinear_up(logit_lr, (H, W))                 # build.py:123-125
        embed = bilinear_up(embed_lr, (H, W))                 # build.py:133-135
        radius_lr = mapper.poincare_distance_origin(embed_lr, dim=1)
    gt = torch.randint(0, O, (H, W), generator=g, dtype=torch.int64)
    ign = torch.rand(H, W, generator=g) < 0.05
    gt[ign] = 255
    prior_active = torch.zeros(H, W, dtype=torch.bool)
    # a previously labelled block plus scattered pixels (round > 1 state)
    prior_active[H // 4: H // 4 + 7, W // 2: W // 2 + 9] = True
    prior_active |= torch.rand(H, W, generator=g) < 0.01
    return dict(z=z, P_MLR=mlr.P_MLR.detach().clone(), A_MLR=mlr.A_MLR.detach().clone(),
                embed_lr=embed_lr, logit_lr64=logit_lr64, logit_lr=logit_lr,
                radius_lr=radius_lr, logit=logit, embed=embed, gt=gt,
                prior_active=prior_active)


def _better(a, b):
    """torch.max ordering: NaN beats everything, otherwise '>'."""
    if math.isnan(a):
        return not math.isnan(b)
    if math.isnan(b):
        return False
    return a > b


def run_selection(ab, score, n, active_radius, mask_radius, active, selected, active_mask, gt):
    """Drive the reference's select_pixels_to_label one region at a time (n calls
    with active_regions=1 are the same loop as one call with active_regions=n,
    build.py:37-62) so that the ORDER of picks and the runner-up gap can be
    recorded; the state that is stored is the reference's own."""
    picks = []
    min_rel_gap = float("inf")
    for _ in range(n):
        before = score.clone()
        if not torch.isnan(before).any() and float(before.max()) == -float("inf"):
            break
        ab.select_pixels_to_label(score, 1, active_radius, mask_radius,
                                  active, selected, active_mask, gt)
        changed = torch.isinf(score) & (score < 0) & ~(torch.isinf(before) & (before < 0))
        idx = torch.nonzero(changed)
        best = None
        for hh, ww in idx.tolist():
            v = float(before[hh, ww])
            if best is None:
                best = (v, hh, ww)
                continue
            bv, bh, bw = best
            if _better(v, bv) or (not _better(bv, v) and (ww, hh) < (bw, bh)):
                best = (v, hh, ww)
        assert best is not None
        v, hh, ww = best
        picks.append((hh, ww, v))
        if not math.isnan(v):
            rest = before.clone()
            rest[hh, ww] = -float("inf")
            ru = float(rest.max())
            if math.isfinite(ru) and v != 0.0:
                min_rel_gap = min(min_rel_gap, (v - ru) / abs(v))
    arr = np.array(picks, dtype=np.float64).reshape(-1, 3)
    return arr, min_rel_gap


COMBOS = [
    # (tag, unc_type, pur_type, normalize, mask_radius, K)
    ("halo", "entropy", "radius", True, 5, 100),           # configs/gtav/source_target.yaml:20-27
    ("ripu", "entropy", "ripu", False, 3, 100),            # configs/gtav/ripu.yaml:23-29
    ("hyper", "entropy", "hyper", True, 5, 100),           # defaults.py:68 (PURITY default)
    ("hyperK10", "entropy", "hyper", True, 5, 10),
    ("none_radius", "none", "radius", True, 5, 100),       # 0/0 -> NaN map
    ("pixent_euc", "pixel_entropy", "euc_norm", True, 5, 100),
    ("oracle", "oracle_acc", "oracle_ripu", False, 3, 100),
    ("ent_none", "entropy", "none", False, 5, 100),
    ("vestigial", "hyperbolic", "ripu", True, 5, 100),     # visualize.py:27-30: zeros branch
]


def gen_case(cfg, hyp, fr, ab, name, H, W, C, O, seed, n_regions, combos, f32_embed=False):
    cfg.MODEL.NUM_CLASSES = O
    inp = make_inputs(hyp, H, W, C, O, seed)
    out = {k: v.numpy() for k, v in inp.items()}
    out["meta_HWCO"] = np.array([H, W, C, O], dtype=np.int64)
    out["meta_n_regions"] = np.array([n_regions], dtype=np.int64)
    embed = inp["embed"].float() if f32_embed else inp["embed"]
    if f32_embed:
        out["embed"] = embed.numpy()
    gaps = {}
    for tag, unc, pur, norm, mrad, K in combos:
        frs = fr.FloatingRegionScore(in_channels=O, size=3, purity_type=pur, K=K)
        with torch.no_grad():
            score, imp, uncm = frs(inp["logit"].clone(), decoder_out=embed.clone(),
                                   unc_type=unc, pur_type=pur, normalize=norm,
                                   ground_truth=inp["gt"].clone())
        out[f"{tag}__score"] = score.numpy().copy()
        out[f"{tag}__impurity"] = imp.numpy().copy()
        out[f"{tag}__uncertainty"] = uncm.numpy().copy()
        # round 1 (build.py:145-160), prior picks masked first
        active = inp["prior_active"].clone()
        selected = torch.zeros(H, W, dtype=torch.bool)
        active_mask = torch.full((H, W), 255, dtype=torch.int64)
        s = score.clone()
        s[active] = -float("inf")
        picks1, g1 = run_selection(ab, s, n_regions, 1, mrad, active, selected, active_mask, inp["gt"])
        out[f"{tag}__r1_picks"] = picks1
        out[f"{tag}__r1_score"] = s.numpy().copy()
        out[f"{tag}__r1_active"] = active.numpy().copy()
        out[f"{tag}__r1_selected"] = selected.numpy().copy()
        out[f"{tag}__r1_active_mask"] = active_mask.numpy().copy()
        # round 2 carries active/selected/active_mask over (cityscapes.py:245-251)
        s = score.clone()
        s[active] = -float("inf")
        picks2, g2 = run_selection(ab, s, n_regions, 1, mrad, active, selected, active_mask, inp["gt"])
        out[f"{tag}__r2_picks"] = picks2
        out[f"{tag}__r2_active"] = active.numpy().copy()
        out[f"{tag}__r2_selected"] = selected.numpy().copy()
        out[f"{tag}__r2_active_mask"] = active_mask.numpy().copy()
        out[f"{tag}__params"] = np.array([mrad, K, int(norm)], dtype=np.int64)
        gaps[tag] = min(g1, g2)
        out[f"{tag}__min_rel_gap"] = np.array([gaps[tag]], dtype=np.float64)
        print(f"  {name}/{tag}: score {score.dtype} picks r1={len(picks1)} r2={len(picks2)} "
              f"min_rel_gap={gaps[tag]:.3e}")
    np.savez_compressed(os.path.join(OUT_DIR, f"{name}.npz"), **out)


def gen_hypermapper(hyp):
    """Last-dim API of HyperMapper incl. the in-tree-dead methods (hyperbolic.py:41-97)."""
    g = torch.Generator().manual_seed(99)
    out = {}
    for c in (1.0, 0.5):
        m = hyp.HyperMapper(c=c)
        x = torch.randn(37, 12, generator=g, dtype=torch.float32) * 0.4
        x[3] *= 100.0
        x[5] = 0.0
        xh = m.expmap(x)
        yh = m.expmap(torch.randn(37, 12, generator=g, dtype=torch.float32) * 0.7)
        tag = f"c{c}"
        out[f"{tag}__x"] = x.numpy()
        out[f"{tag}__expmap"] = xh.numpy()
        out[f"{tag}__y_h"] = yh.numpy()
        out[f"{tag}__logmap"] = m.logmap(xh).numpy()
        out[f"{tag}__dist0"] = m.poincare_distance_origin(xh).numpy()
        out[f"{tag}__dist"] = m.poincare_distance(xh, yh).numpy()
        out[f"{tag}__expmap2"] = m.expmap2(x.double()).numpy()
        out[f"{tag}__cosine"] = m.cosine_distance(x[6:], yh[6:].float()).numpy()
        # float32 points: geoopt's dist0 then stays in float32 (artanh's two logs in the input dtype, shim docstring); rows
        # scaled to radii from 1e-4 to the clamp at 1 - 1e-7 -- own generator, so the arrays above keep their bits
        g3 = torch.Generator().manual_seed(101 + int(c * 10))
        xf = torch.randn(41, 12, generator=g3, dtype=torch.float32)
        xf = xf / xf.norm(dim=1, keepdim=True)
        radii = torch.cat([torch.logspace(-4, -0.01, 30), torch.tensor([0.99, 0.999, 0.9999, 0.99999, 0.999999, 0.9999999, 1.0, 1.5,
                                                                         0.0, 0.3, 0.6])]).float() / math.sqrt(c)
        xf = (xf * radii[:, None]).contiguous()
        out[f"{tag}__x_f32"] = xf.numpy()
        out[f"{tag}__dist0_f32"] = m.poincare_distance_origin(xf).numpy()
        assert out[f"{tag}__dist0_f32"].dtype == np.float32
    # HyperMetrics.compute (hyperbolic.py:191-228; no caller in-tree) -- its own generator, so the arrays above keep their bits
    g2 = torch.Generator().manual_seed(100)
    for c in (1.0, 0.5):
        hm = hyp.HyperMetrics(c=c)
        a = torch.randn(23, 10, generator=g2, dtype=torch.float32) * 0.5
        b = torch.randn(23, 10, generator=g2, dtype=torch.float32) * 0.5
        met = hm.compute(a, b)
        out[f"hm_c{c}__x"], out[f"hm_c{c}__y"] = a.numpy(), b.numpy()
        for key, val in met.items():
            out[f"hm_c{c}__{key}"] = val.numpy()
    np.savez_compressed(os.path.join(OUT_DIR, "hypermapper.npz"), **out)


def gen_helpers(cfg, hyp, fr):
    """FloatingRegionScore's helper methods (floating_region.py:70-127) called directly."""
    H, W, C, O = 48, 80, 8, 19
    cfg.MODEL.NUM_CLASSES = O
    inp = make_inputs(hyp, H, W, C, O, seed=77)
    out = {"meta_HWCO": np.array([H, W, C, O], dtype=np.int64)}
    with torch.no_grad():
        logit = inp["logit"][0]
        p = torch.softmax(logit, dim=0)
        out["logit"] = logit.numpy(); out["p"] = p.numpy(); out["gt"] = inp["gt"].numpy()
        out["embed"] = inp["embed"].numpy()
        f_hyp = fr.FloatingRegionScore(in_channels=O, size=3, purity_type="hyper", K=100)
        f_rip = fr.FloatingRegionScore(in_channels=O, size=5, purity_type="ripu")
        out["pixel_entropy"] = f_hyp.compute_pixel_entropy(p).numpy()
        out["ru_entropy_k3"] = f_hyp.compute_region_uncertainty("entropy", logit, p).numpy()
        out["ru_entropy_k5"] = f_rip.compute_region_uncertainty("entropy", logit, p).numpy()
        out["ru_oracle_acc"] = f_hyp.compute_region_uncertainty("oracle_acc", logit, p, ground_truth=inp["gt"]).numpy()
        out["ru_none"] = f_hyp.compute_region_uncertainty("none", logit, p).numpy()
        out["ru_hyperbolic"] = f_hyp.compute_region_uncertainty("hyperbolic", logit, p).numpy()
        q = f_hyp.quantize_uncert_map(inp["embed"])
        out["quantized"] = q.numpy()
        imp, cnt = f_hyp.compute_region_impurity(q, 100)
        out["imp_hyper"] = imp.numpy(); out["cnt_hyper"] = cnt.numpy()
        am = torch.argmax(p, dim=0)
        imp, cnt = f_rip.compute_region_impurity(am, O)
        out["argmax"] = am.numpy(); out["imp_ripu_k5"] = imp.numpy(); out["cnt_ripu_k5"] = cnt.numpy()
    np.savez_compressed(os.path.join(OUT_DIR, "helpers.npz"), **out)


def gen_grads(hyp):
    """Autograd of the reference head tail (classifier.py:553-554): d loss / d {feat, P_MLR, A_MLR} for
    loss = <out, Wt> + <embed, Ve>, with far-out (projected / tanh-clamped) and exact-origin pixels."""
    out = {}
    # (c64_o19: the head's own 64 channels x 19 classes -- the shape the fused native backward serves; every case draws from its own
    #  generator, so adding one leaves the other arrays' bits alone)
    for tag, (C, O, h, w, c) in {"c8_o19": (8, 19, 9, 13, 1.0), "c16_o16_k07": (16, 16, 6, 10, 0.7), "c64_o19": (64, 19, 10, 14, 1.0)}.items():
        g = torch.Generator().manual_seed(5 + C)
        z = torch.randn(2, C, h, w, generator=g, dtype=torch.float32) * 0.3
        z[0, :, 0, 0] *= 300.0            # tanh clamp + project
        z[1, :, 2, 3] *= 20.0             # project only
        z[0, :, 1, 1] = 0.0               # exact origin
        z.requires_grad_(True)
        mapper = hyp.HyperMapper(c=c)
        mlr = hyp.HyperMLR(C, O, c=c)
        with torch.no_grad():
            torch.manual_seed(3)
            torch.nn.init.kaiming_uniform_(mlr.P_MLR, a=math.sqrt(5))
            torch.nn.init.kaiming_uniform_(mlr.A_MLR, a=math.sqrt(5))
        embed = mapper.expmap(z, dim=1)
        embed.retain_grad()
        logits = mlr(embed.double()).float()
        Wt = torch.randn(logits.shape, generator=g, dtype=torch.float32)
        Ve = torch.randn(embed.shape, generator=g, dtype=torch.float64) * 0.1
        loss = (logits * Wt).sum() + (embed * Ve).sum()
        loss.backward()
        out.update({f"{tag}__z": z.detach().numpy(), f"{tag}__P": mlr.P_MLR.detach().numpy(), f"{tag}__A": mlr.A_MLR.detach().numpy(),
                    f"{tag}__Wt": Wt.numpy(), f"{tag}__Ve": Ve.numpy(), f"{tag}__c": np.array([c]),
                    f"{tag}__embed": embed.detach().numpy(), f"{tag}__logits": logits.detach().numpy(),
                    f"{tag}__g_z": z.grad.numpy(), f"{tag}__g_embed": embed.grad.numpy(),
                    f"{tag}__g_P": mlr.P_MLR.grad.numpy(), f"{tag}__g_A": mlr.A_MLR.grad.numpy()})
        print(f"  grads {tag}: |g_z| max {float(z.grad.abs().max()):.3e}, nan {bool(torch.isnan(z.grad).any())}")
    # the three geoopt-backed ops nothing in the tree differentiates (hyperbolic.py:51-83) are differentiable all the same:
    # d <op(x), W> / d x under the reference's autograd -- own generator, so the arrays above keep their bits
    g3 = torch.Generator().manual_seed(77)
    for c in (1.0, 0.7):
        m = hyp.HyperMapper(c=c)
        xh = m.expmap(torch.randn(21, 9, generator=g3, dtype=torch.float32) * 0.5).detach()
        yh = m.expmap(torch.randn(21, 9, generator=g3, dtype=torch.float32) * 0.5).detach()
        xh[4] = m.expmap(torch.randn(9, generator=g3, dtype=torch.float32) * 40.0).detach()     # on the projection limit
        W1 = torch.randn(21, 9, generator=g3, dtype=torch.float64)
        W2 = torch.randn(21, generator=g3, dtype=torch.float64)
        tag = f"ops_c{c}"
        out[f"{tag}__x"], out[f"{tag}__y"], out[f"{tag}__W1"], out[f"{tag}__W2"] = xh.numpy(), yh.numpy(), W1.numpy(), W2.numpy()
        a = xh.clone().requires_grad_(True)
        (m.logmap(a) * W1).sum().backward()
        out[f"{tag}__g_logmap"] = a.grad.numpy()
        a = xh.clone().requires_grad_(True); b = yh.clone().requires_grad_(True)
        (m.poincare_distance(a, b) * W2).sum().backward()
        out[f"{tag}__g_dist_x"], out[f"{tag}__g_dist_y"] = a.grad.numpy(), b.grad.numpy()
        a = xh.clone().requires_grad_(True)
        (m.poincare_distance_origin(a) * W2).sum().backward()
        out[f"{tag}__g_dist0"] = a.grad.numpy()
    np.savez_compressed(os.path.join(OUT_DIR, "grads.npz"), **out)


def gen_losses():
    """core/loss: NegativeLearningLoss and LocalConsistentLoss (l1 and kl) forward values and gradients
    w.r.t. their inputs, from the reference's own modules under autograd."""
    from core.loss.local_consistent_loss import LocalConsistentLoss
    from core.loss.negative_learning_loss import NegativeLearningLoss
    g = torch.Generator().manual_seed(123)
    out = {}
    B, O, h, w = 2, 19, 24, 40
    low = torch.randn(B, O, h // 4, w // 4, generator=g) * 2.0
    x = F.interpolate(low, size=(h, w), mode="bilinear", align_corners=True) + 0.3 * torch.randn(B, O, h, w, generator=g)
    x = x.clone().requires_grad_(True)
    label = torch.randint(0, O, (B, h // 4, w // 4), generator=g)
    label = label.repeat_interleave(4, dim=1).repeat_interleave(4, dim=2)         # blocky label map: real boundaries
    label[torch.rand(B, h, w, generator=g) < 0.05] = 255
    out["x"] = x.detach().numpy(); out["label"] = label.numpy()
    for lt in ("l1", "kl"):
        crit = LocalConsistentLoss(O, lt)
        loss = crit(x, label)
        (gx,) = torch.autograd.grad(loss, x)
        out[f"lcl_{lt}__loss"] = np.array([loss.item()], dtype=np.float64)
        out[f"lcl_{lt}__gx"] = gx.numpy()
        print(f"  LocalConsistentLoss {lt}: loss {loss.item():.6f} |gx| max {float(gx.abs().max()):.3e}")
    pr = torch.softmax(x.detach(), dim=1).clone().requires_grad_(True)
    nl = NegativeLearningLoss(threshold=0.05)
    loss = nl(pr)
    (gp,) = torch.autograd.grad(loss, pr)
    out["neg__p"] = pr.detach().numpy(); out["neg__loss"] = np.array([loss.item()], dtype=np.float64); out["neg__gp"] = gp.numpy()
    print(f"  NegativeLearningLoss: loss {loss.item():.6f}")
    # degenerate: no boundary pixel at all (constant labels) -> mean over an empty selection
    lab0 = torch.zeros(1, 8, 8, dtype=torch.long)
    x0 = torch.randn(1, O, 8, 8, generator=g).requires_grad_(True)
    l0 = LocalConsistentLoss(O, "l1")(x0, lab0)
    (g0,) = torch.autograd.grad(l0, x0, allow_unused=True)
    out["empty__x"] = x0.detach().numpy(); out["empty__loss_isnan"] = np.array([bool(torch.isnan(l0))])
    out["empty__gx"] = (g0 if g0 is not None else torch.zeros_like(x0)).numpy()
    np.savez_compressed(os.path.join(OUT_DIR, "losses.npz"), **out)


class _FakeExtractor(torch.nn.Module):
    def forward(self, x):
        return x


class _FakeClassifier(torch.nn.Module):
    """Returns pre-computed low-res (logits, embed) per call -- stands for backbone+head."""

    def __init__(self, outs):
        super().__init__()
        self.outs = outs
        self.i = 0

    def forward(self, feat, size=None):
        o = self.outs[self.i % len(self.outs)]
        self.i += 1
        return o


def gen_region_selection(cfg, hyp, fr, ab):
    """Two rounds of the reference's RegionSelection driver (build.py:71-186) over a
    3-image pool, through its real PNG / torch.save persistence."""
    from PIL import Image
    H, W, C, O = 48, 96, 8, 19
    cfg.MODEL.NUM_CLASSES = O
    cfg.MODEL.HYPER = True
    cfg.ACTIVE.UNCERTAINTY = "entropy"
    cfg.ACTIVE.PURITY = "radius"
    cfg.ACTIVE.NORMALIZE = True
    cfg.ACTIVE.RADIUS_K = 1
    cfg.ACTIVE.MASK_RADIUS_K = 5
    cfg.ACTIVE.BUDGET = 0.05
    cfg.ACTIVE.SELECT_ITER = [0, 1, 2, 3, 4]
    cfg.ACTIVE.K = 100
    cfg.ACTIVE.VIZ_MASK = False
    out = {"meta_HWCO": np.array([H, W, C, O], dtype=np.int64)}
    tmp = tempfile.mkdtemp(prefix="halo_fix_")
    imgs = []
    for i in range(3):
        inp = make_inputs(hyp, H, W, C, O, seed=500 + i)
        imgs.append(inp)
        out[f"img{i}__logit_lr"] = inp["logit_lr"].numpy()
        out[f"img{i}__embed_lr"] = inp["embed_lr"].numpy()
        out[f"img{i}__gt"] = inp["gt"].numpy()
        Image.fromarray(np.full((H, W), 255, dtype=np.uint8)).save(os.path.join(tmp, f"m{i}.png"))
        torch.save({"active": torch.tensor([0], dtype=torch.bool),
                    "selected": torch.tensor([0], dtype=torch.bool)},
                   os.path.join(tmp, f"i{i}.pth"))

    def loader():
        for i, inp in enumerate(imgs):
            ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
            a, s = ind["active"], ind["selected"]
            mask = torch.from_numpy(np.array(Image.open(os.path.join(tmp, f"m{i}.png")),
                                             dtype=np.uint8)).long()
            if a.size() == (1,):  # cityscapes.py:249-251
                a = torch.zeros(H, W, dtype=torch.bool)
                s = torch.zeros(H, W, dtype=torch.bool)
            yield {"img": torch.zeros(1, 3, H // 2, W // 2),
                   "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
                   "origin_mask": mask[None], "origin_label": inp["gt"][None],
                   "size": torch.tensor([[H, W]]), "active": a[None], "selected": s[None],
                   "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")],
                   "name": [f"img{i}"]}

    for rnd in (1, 2):
        clf = _FakeClassifier([(inp["logit_lr"], inp["embed_lr"]) for inp in imgs])
        ab.RegionSelection(cfg, _FakeExtractor(), clf, list(loader()), rnd)
        for i in range(3):
            ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
            out[f"r{rnd}_img{i}__active"] = ind["active"].numpy()
            out[f"r{rnd}_img{i}__selected"] = ind["selected"].numpy()
            out[f"r{rnd}_img{i}__mask_png"] = np.array(
                Image.open(os.path.join(tmp, f"m{i}.png")), dtype=np.uint8)
        print(f"  region_selection round {rnd}: selected px img0 = "
              f"{int(out[f'r{rnd}_img0__selected'].sum())}")
    np.savez_compressed(os.path.join(OUT_DIR, "region_selection.npz"), **out)


def gen_padding(cfg, hyp, fr, ab):
    """FloatingRegionScore(padding_mode=...) for the three non-default modes nn.Conv2d accepts (floating_region.py:49,63; no
    caller in the reference's tree passes one, so these are the reference's CLASS run directly): forward maps for several
    branches and window sizes, the helper methods, and the first-round picks."""
    H, W, C, O = 24, 40, 8, 19
    cfg.MODEL.NUM_CLASSES = O
    inp = make_inputs(hyp, H, W, C, O, 77)
    out = {k: inp[k].numpy() for k in ("logit", "embed", "gt", "prior_active")}
    out["meta_HWCO"] = np.array([H, W, C, O], dtype=np.int64)
    combos = [("halo", "entropy", "radius", True, 3, 100), ("ripu5", "entropy", "ripu", False, 5, 100),
              ("hyperK10", "entropy", "hyper", True, 3, 10), ("oracle", "oracle_acc", "oracle_ripu", False, 3, 100)]
    for mode in ("reflect", "replicate", "circular"):
        for tag, unc, pur, norm, size, K in combos:
            frs = fr.FloatingRegionScore(in_channels=O, padding_mode=mode, size=size, purity_type=pur, K=K)
            with torch.no_grad():
                score, imp, uncm = frs(inp["logit"].clone(), decoder_out=inp["embed"].clone(), unc_type=unc, pur_type=pur,
                                       normalize=norm, ground_truth=inp["gt"].clone())
            key = f"{mode}__{tag}"
            out[key + "__score"], out[key + "__impurity"], out[key + "__uncertainty"] = score.numpy().copy(), imp.numpy().copy(), uncm.numpy().copy()
            out[key + "__params"] = np.array([size, K, int(norm)], dtype=np.int64)
            active = inp["prior_active"].clone()
            selected = torch.zeros(H, W, dtype=torch.bool)
            active_mask = torch.full((H, W), 255, dtype=torch.int64)
            s = score.clone()
            s[active] = -float("inf")
            picks, gap = run_selection(ab, s, 12, 1, 3, active, selected, active_mask, inp["gt"])
            out[key + "__picks"], out[key + "__active_mask"] = picks, active_mask.numpy().copy()
            out[key + "__min_rel_gap"] = np.array([gap], dtype=np.float64)
            print(f"  padding/{key}: picks {len(picks)} min_rel_gap {gap:.3e}")
        # helper methods with the mode (size 5 windows): box-summed entropy of given probabilities, window histogram impurity
        frs = fr.FloatingRegionScore(in_channels=O, padding_mode=mode, size=5, purity_type="ripu")
        with torch.no_grad():
            p = torch.softmax(inp["logit"][0], dim=0)
            out[f"{mode}__region_unc_k5"] = frs.compute_region_uncertainty("entropy", inp["logit"][0], p).numpy().copy()
            imp, cnt = frs.compute_region_impurity(p.argmax(dim=0), O)
            out[f"{mode}__imp_k5"], out[f"{mode}__cnt_k5"] = imp.numpy().copy(), cnt.numpy().copy()
    np.savez_compressed(os.path.join(OUT_DIR, "padding.npz"), **out)


# ---- reference-held vectors at sizes where ATen's CPU kernels take their production paths ----
sys.path.insert(0, os.path.dirname(HERE))          # tests/fullsize_inputs.py: numpy + oracle C only, travels to the GPU box
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # the repository root: oracle/
SAMPLE_STRIDE = 1031                                # the maps are sampled at every 1031st pixel (prime): 2035 values per map

# tag -> (seed, C, branch, modifiers, float32 embedding); 1024 x 2048, 19 classes, 2331 regions (core/active/build.py:148-150)
FULLSIZE = {
    "halo_c256_s1234": (1234, 256, "halo", (), False),                               # the bench's image 0 shape (BASELINE configs[1])
    "halo_c64_late_sat_peak_s3": (3, 64, "halo", ("late_round", "saturated", "peaked"), False),
    "ripu_s2": (2, 64, "ripu", (), False),
    "hyper_s1": (1, 64, "hyper", (), False),                                       # the reference's DEFAULT purity (defaults.py:69)
}


def run_reference_fullsize(fr, ab, inp, branch, O=19, n=None):
    """The reference's FloatingRegionScore.forward + select_pixels_to_label on numpy inputs (tests/fullsize_inputs.build).
    The pick table is read off the reference's own score map by the oracle's selector (the reference keeps no table,
    build.py:37-62) and CHECKED against the masks the reference's select_pixels_to_label left behind."""
    import fullsize_inputs as fi
    import oracle.halo_oracle as ho
    unc, pur, norm, mrad, K = fi.BRANCHES[branch]
    logit, embed, gt = torch.from_numpy(inp["logit"]), torch.from_numpy(inp["embed"]), torch.from_numpy(inp["gt"])
    H, W = gt.shape
    n = fi.n_regions(H, W) if n is None else n
    frs = fr.FloatingRegionScore(in_channels=O, size=3, purity_type=pur, K=K)
    with torch.no_grad():
        score, imp, uncm = frs(logit.clone(), decoder_out=embed, unc_type=unc, pur_type=pur, normalize=norm, ground_truth=gt.clone())
    active = torch.from_numpy(inp["prior"].copy())
    selected = torch.zeros(H, W, dtype=torch.bool)
    amask = torch.full((H, W), 255, dtype=torch.int64)
    score = score.clone()
    score[active] = -float("inf")                                                   # build.py:146
    s0 = score.numpy().copy()
    ab.select_pixels_to_label(score, n, 1, mrad, active, selected, amask, gt)        # build.py:151-160
    a2, s2, m2 = inp["prior"].copy(), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
    _, _, _, _, picks = ho.select_pixels_to_label(s0.copy(), n, 1, mrad, a2, s2, m2, inp["gt"], return_picks=True)
    assert np.array_equal(a2, active.numpy()) and np.array_equal(s2, selected.numpy()) and np.array_equal(m2, amask.numpy()), \
        "the replayed table does not reproduce the reference's masks"
    return dict(score=s0, impurity=imp.numpy(), uncertainty=uncm.numpy(), active=a2, selected=s2, active_mask=m2, picks=picks)


def mask_digest(res):
    import hashlib
    hsh = hashlib.sha256()
    for k in ("active", "selected", "active_mask"):
        hsh.update(np.ascontiguousarray(res[k]).tobytes())
    return np.frombuffer(hsh.digest(), dtype=np.uint8).copy()


def gen_fullsize(cfg, fr, ab):
    """Pick tables + sampled maps of the reference at 1024 x 2048 (what VERDICT r5 calls reference-held golden at BASELINE size)."""
    import fullsize_inputs as fi
    cfg.MODEL.NUM_CLASSES = 19
    out = {}
    for tag, (seed, C, branch, mods, f32) in FULLSIZE.items():
        inp = fi.build(seed, C=C, mods=mods, f32_embed=f32)
        res = run_reference_fullsize(fr, ab, inp, branch)
        out[f"{tag}__digest"] = np.frombuffer(fi.digest(inp).encode(), dtype=np.uint8).copy()
        out[f"{tag}__picks"] = res["picks"]
        out[f"{tag}__mask_digest"] = mask_digest(res)
        out[f"{tag}__n_selected"] = np.array([int(res["selected"].sum()), int(res["active"].sum())], dtype=np.int64)
        for k in ("score", "impurity", "uncertainty"):
            out[f"{tag}__{k}_sample"] = res[k].ravel()[::SAMPLE_STRIDE].copy()
        print(f"  fullsize/{tag}: picks {len(res['picks'])}, selected px {int(res['selected'].sum())}, score {res['score'].dtype}")
    np.savez_compressed(os.path.join(OUT_DIR, "fullsize_picks.npz"), **out)


def files_digest(mask_png, active, selected):
    import hashlib
    hsh = hashlib.sha256()
    for a in (mask_png, active, selected):
        hsh.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(hsh.digest(), dtype=np.uint8).copy()


DRIVER_SEEDS = (41,)


def gen_fullsize_driver(cfg, hyp, fr, ab):
    """The reference's OWN RegionSelection (core/active/build.py:71-186: its two F.interpolate calls, scorer, selector, PNG and
    torch.save) over the real pipeline's geometry at full label size, two rounds: digests of the files it leaves behind."""
    import fullsize_inputs as fi
    from PIL import Image
    H, W, C, O = 1024, 2048, 64, 19
    cfg.MODEL.NUM_CLASSES, cfg.MODEL.HYPER = O, True
    cfg.ACTIVE.UNCERTAINTY, cfg.ACTIVE.PURITY, cfg.ACTIVE.NORMALIZE = "entropy", "radius", True
    cfg.ACTIVE.RADIUS_K, cfg.ACTIVE.MASK_RADIUS_K, cfg.ACTIVE.BUDGET, cfg.ACTIVE.SELECT_ITER = 1, 5, 0.05, [0, 1, 2, 3, 4]
    cfg.ACTIVE.K, cfg.ACTIVE.VIZ_MASK = 100, False
    out = {}
    tmp = tempfile.mkdtemp(prefix="halo_fix_drv_")
    for seed in DRIVER_SEEDS:
        inp = fi.build_driver_inputs(seed, C=C, O=O, H=H, W=W)
        out[f"s{seed}__digest"] = np.frombuffer(fi.digest(inp).encode(), dtype=np.uint8).copy()
        pm, pi = os.path.join(tmp, f"m{seed}.png"), os.path.join(tmp, f"i{seed}.pth")
        Image.fromarray(np.full((H, W), 255, dtype=np.uint8)).save(pm)
        torch.save({"active": torch.tensor([0], dtype=torch.bool), "selected": torch.tensor([0], dtype=torch.bool)}, pi)
        gt = torch.from_numpy(inp["gt"])
        for rnd in (1, 2):
            ind = torch.load(pi)
            a, s_ = ind["active"], ind["selected"]
            if a.size() == (1,):                                                # cityscapes.py:249-251
                a, s_ = torch.zeros(H, W, dtype=torch.bool), torch.zeros(H, W, dtype=torch.bool)
            mask = torch.from_numpy(np.array(Image.open(pm), dtype=np.uint8)).long()
            item = {"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [pm], "origin_mask": mask[None], "origin_label": gt[None],
                    "size": torch.tensor([[H, W]]), "active": a[None], "selected": s_[None], "path_to_indicator": [pi], "name": [f"s{seed}"]}
            clf = _FakeClassifier([(torch.from_numpy(inp["logit_lr"]), torch.from_numpy(inp["embed_lr"]))])
            ab.RegionSelection(cfg, _FakeExtractor(), clf, [item], rnd)
            ind = torch.load(pi)
            png = np.array(Image.open(pm), dtype=np.uint8)
            out[f"s{seed}__r{rnd}_files_digest"] = files_digest(png, ind["active"].numpy(), ind["selected"].numpy())
            out[f"s{seed}__r{rnd}_counts"] = np.array([int(ind["selected"].sum()), int(ind["active"].sum()), int((png != 255).sum())], dtype=np.int64)
            print(f"  fullsize driver seed {seed} round {rnd}: selected px {int(ind['selected'].sum())}, labelled px {int((png != 255).sum())}")
    np.savez_compressed(os.path.join(OUT_DIR, "fullsize_driver.npz"), **out)


def gen_mid(cfg, hyp, fr, ab):
    """A case ABOVE 20480 pixels (112 x 192), where the reference's 3 x 3 box convolution runs in oneDNN like at every production
    size (aten/src/ATen/native/Convolution.cpp:use_mkldnn; the cases A-D take the im2col + MKL sgemm path, whose summation order is
    MKL's): low-res inputs (the x4 resize is regenerated by the consumer and checked through `logit_digest`) and the three maps."""
    import hashlib
    H, W, C, O = 112, 192, 8, 19
    cfg.MODEL.NUM_CLASSES = O
    inp = make_inputs(hyp, H, W, C, O, 55)
    out = {k: inp[k].numpy() for k in ("logit_lr", "embed_lr", "gt", "prior_active")}
    out["meta_HWCO"] = np.array([H, W, C, O], dtype=np.int64)
    hsh = hashlib.sha256()
    hsh.update(inp["logit"].numpy().tobytes())
    hsh.update(inp["embed"].numpy().tobytes())
    out["resized_digest"] = np.frombuffer(hsh.digest(), dtype=np.uint8).copy()
    for tag, unc, pur, norm, mrad, K in [c for c in COMBOS if c[0] in ("halo", "ripu", "hyper")]:
        frs = fr.FloatingRegionScore(in_channels=O, size=3, purity_type=pur, K=K)
        with torch.no_grad():
            score, imp, uncm = frs(inp["logit"].clone(), decoder_out=inp["embed"].clone(), unc_type=unc, pur_type=pur, normalize=norm,
                                   ground_truth=inp["gt"].clone())
        out[f"{tag}__score"], out[f"{tag}__impurity"], out[f"{tag}__uncertainty"] = score.numpy().copy(), imp.numpy().copy(), uncm.numpy().copy()
        active = inp["prior_active"].clone()
        selected = torch.zeros(H, W, dtype=torch.bool)
        active_mask = torch.full((H, W), 255, dtype=torch.int64)
        s = score.clone()
        s[active] = -float("inf")
        picks, gap = run_selection(ab, s, 40, 1, mrad, active, selected, active_mask, inp["gt"])
        out[f"{tag}__picks"] = picks
        out[f"{tag}__params"] = np.array([mrad, K, int(norm)], dtype=np.int64)
        print(f"  mid/{tag}: picks {len(picks)} min_rel_gap {gap:.3e}")
    np.savez_compressed(os.path.join(OUT_DIR, "mid_112x192_c8_o19.npz"), **out)


def main():
    torch.set_num_threads(4)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    cfg, hyp, fr, ab = import_reference()
    cfg.MODEL.CURVATURE = 1.0
    print("torch", torch.__version__)
    if only == "helpers":            # add-on vectors without regenerating the rest
        gen_helpers(cfg, hyp, fr)
        return
    if only == "grads":
        gen_grads(hyp)
        return
    if only == "losses":
        gen_losses()
        return
    if only == "padding":
        gen_padding(cfg, hyp, fr, ab)
        return
    if only == "fullsize":
        gen_fullsize(cfg, fr, ab)
        return
    if only == "fullsize_driver":
        gen_fullsize_driver(cfg, hyp, fr, ab)
        return
    if only == "mid":
        gen_mid(cfg, hyp, fr, ab)
        return
    print("case A 32x64 C8 O19 (selection runs to exhaustion)")
    gen_case(cfg, hyp, fr, ab, "case_a_32x64_c8_o19", 32, 64, 8, 19, 11, 200, COMBOS)
    print("case B 64x128 C16 O19")
    gen_case(cfg, hyp, fr, ab, "case_b_64x128_c16_o19", 64, 128, 16, 19, 22, 60, COMBOS)
    print("case C 48x96 C8 O16 (SYNTHIA class count; log(19) stays hard-coded)")
    gen_case(cfg, hyp, fr, ab, "case_c_48x96_c8_o16", 48, 96, 8, 16, 33, 40,
             [c for c in COMBOS if c[0] in ("halo", "ripu", "hyper", "oracle")])
    print("case D 40x72 C8 O19 float32 embed (MODEL.HYPER=False style input)")
    gen_case(cfg, hyp, fr, ab, "case_d_40x72_c8_o19_f32", 40, 72, 8, 19, 44, 30,
             [c for c in COMBOS if c[0] in ("halo", "hyper", "pixent_euc")], f32_embed=True)
    print("hypermapper last-dim API")
    gen_hypermapper(hyp)
    print("helper methods")
    gen_helpers(cfg, hyp, fr)
    print("head-tail gradients")
    gen_grads(hyp)
    print("training losses")
    gen_losses()
    print("RegionSelection driver, 2 rounds")
    gen_region_selection(cfg, hyp, fr, ab)
    print("padding modes of the two box windows")
    gen_padding(cfg, hyp, fr, ab)
    print("mid 112x192 C8 O19 (above ATen's 20480-pixel switch to oneDNN)")
    gen_mid(cfg, hyp, fr, ab)
    print("reference pick tables at 1024 x 2048")
    gen_fullsize(cfg, fr, ab)
    print("the reference's RegionSelection driver at 1024 x 2048, real head geometry")
    gen_fullsize_driver(cfg, hyp, fr, ab)


if __name__ == "__main__":
    main()
