"""halo_amd.install(): the reference tree's own import lines resolve to the HIP path (SURVEY 8b:
"import paths + names").  Runs against a FAKE `core` tree shaped like the reference's (same package layout,
same import statements as core/train_learners.py:12-20, core/utils/visualize.py:6-8, core/models/classifier.py:4-5,
core/active/__init__.py:1), in a subprocess so the test session's sys.modules stay clean."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

TREE = {
    "core/__init__.py": "",
    "core/configs/__init__.py": """
        class _N(dict):
            __getattr__ = dict.__getitem__
        cfg = _N(MODEL=_N(NUM_CLASSES=16, HYPER=True, CURVATURE=0.7, REDUCED_CHANNELS=8),
                 ACTIVE=_N(UNCERTAINTY='entropy', PURITY='radius', SELECT_ITER=[0, 1], BUDGET=0.02, RADIUS_K=1,
                           NORMALIZE=True, MASK_RADIUS_K=5, K=100, VIZ_MASK=False))
        """,
    "core/active/__init__.py": "from .build import *\n",
    "core/active/build.py": "raise ImportError('the reference build.py must not be loaded after install()')\n",
    "core/active/floating_region.py": "raise ImportError('the reference floating_region.py must not be loaded')\n",
    "core/utils/hyperbolic.py": "raise ImportError('needs geoopt')\n",
    "core/loss/__init__.py": "",
    "core/loss/local_consistent_loss.py": "raise ImportError('reference loss must not be loaded')\n",
    "core/loss/negative_learning_loss.py": "raise ImportError('reference loss must not be loaded')\n",
    "core/models/__init__.py": "",
    "core/models/classifier.py": """
        import torch.nn as nn
        from ..utils.hyperbolic import HyperMapper, HyperMLR
        from core.configs import cfg

        class ASPP_Classifier_V2_Hyper(nn.Module):
            def __init__(self, in_channels, dilation_series, padding_series, num_classes, reduced_channels):
                super().__init__()
                self.conv2d_list = nn.ModuleList(nn.Conv2d(in_channels, reduced_channels, 3, 1, p, d) for d, p in
                                                 zip(dilation_series, padding_series))
                self.mapper = HyperMapper(c=cfg.MODEL.CURVATURE)
                self.conv_seg = HyperMLR(reduced_channels, num_classes, c=cfg.MODEL.CURVATURE)

            def forward(self, x, size=None):
                raise RuntimeError('reference forward: should have been patched')

        class DepthwiseSeparableASPP_Hyper(nn.Module):
            def forward(self, x, size=None):
                raise RuntimeError('reference forward: should have been patched')
        """,
    "core/utils/visualize.py": """
        from core.active.floating_region import FloatingRegionScore
        from core.configs import cfg
        """,
    "core/train_learners.py": """
        from core.active.build import RegionSelection
        from core.configs import cfg
        from core.loss.local_consistent_loss import LocalConsistentLoss
        from core.loss.negative_learning_loss import NegativeLearningLoss
        from core.utils.visualize import FloatingRegionScore
        from core.active import select_pixels_to_label
        """,
}

SCRIPT = """
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tree!r})
import halo_amd
names = halo_amd.install()
assert 'core.active.build' in names and halo_amd.install() == names            # idempotent
import core.train_learners as tl                                               # the learner's own import lines
import halo_amd.core.active.build as hb, halo_amd.core.active.floating_region as hf
import halo_amd.core.loss as hl, halo_amd.core.utils.hyperbolic as hh
assert tl.RegionSelection is hb.RegionSelection and tl.select_pixels_to_label is hb.select_pixels_to_label
assert tl.FloatingRegionScore is hf.FloatingRegionScore
assert tl.LocalConsistentLoss is hl.LocalConsistentLoss and tl.NegativeLearningLoss is hl.NegativeLearningLoss
import core.utils.hyperbolic, core.active.floating_region
assert core.utils.hyperbolic is hh and core.active.floating_region is hf
from core.utils.hyperbolic import HyperMapper, HyperMLR, HyperMetrics          # every public name of the reference module
assert HyperMetrics(c=0.5).mapper.c == 0.5
# ADVICE r2: an aliased module keeps ITS OWN spec (relative imports inside it, importlib.reload)
import importlib
assert hb.__spec__.name == 'halo_amd.core.active.build' and hb.__package__ == 'halo_amd.core.active'
assert importlib.reload(hb) is hb and sys.modules['core.active.build'] is hb
# ONE cfg object: the reference's (curvature 0.7 reaches the scorer and the heads)
import core.configs, halo_amd.core.configs as hc
assert hc.cfg is core.configs.cfg and hf.cfg is core.configs.cfg and hb.cfg is core.configs.cfg
assert hf.FloatingRegionScore(in_channels=16, size=3).mapper.c == 0.7
# the reference's head classes, built by the reference's constructor, now run the HIP tail
import core.models.classifier as rc, halo_amd.core.models.classifier as hcls
assert rc.ASPP_Classifier_V2_Hyper.forward is hcls.v2_hyper_forward
assert rc.DepthwiseSeparableASPP_Hyper.forward is hcls.v3plus_hyper_forward
head = rc.ASPP_Classifier_V2_Hyper(4, [6, 12], [6, 12], 16, 8)
assert isinstance(head.conv_seg, hh.HyperMLR) and head.mapper.c == 0.7
assert sorted(head.state_dict()) == ['conv2d_list.0.bias', 'conv2d_list.0.weight', 'conv2d_list.1.bias',
                                     'conv2d_list.1.weight', 'conv_seg.A_MLR', 'conv_seg.P_MLR']
import torch
try:
    head({{'out': torch.zeros(1, 4, 5, 5)}}, size=(9, 9))
    raise SystemExit('a CPU tensor must not be served')
except halo_amd._lib.HaloHipError:
    pass                                                                       # reached the HIP tail: no CPU fallback
own = hcls.ASPP_Classifier_V2_Hyper(4, [6, 12], [6, 12], 16, 8)                 # halo_amd's own drop-in class
assert sorted(own.state_dict()) == sorted(head.state_dict()) and own.mapper.c == 0.7
halo_amd.uninstall()
assert 'core.active.build' not in sys.modules or sys.modules['core.active.build'] is not hb
assert rc.ASPP_Classifier_V2_Hyper.forward is not hcls.v2_hyper_forward
print('install ok')
"""


def _write_tree(base):
    for rel, body in TREE.items():
        path = os.path.join(base, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(body))


def test_install_serves_the_reference_import_paths(tmp_path):
    _write_tree(str(tmp_path))
    r = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT, tree=str(tmp_path))], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "install ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_install_after_the_tree_was_partly_imported(tmp_path):
    """core.configs and core.models.classifier imported BEFORE install(): attributes of already-imported parents
    are rebound and the heads are patched in place."""
    _write_tree(str(tmp_path))
    # the fake classifier needs core.utils.hyperbolic at import time: give the pre-install phase a stub
    with open(os.path.join(str(tmp_path), "core", "utils", "hyperbolic.py"), "w") as f:
        f.write("class HyperMapper:\n    def __init__(self, c=1.0):\n        self.c = c\n"
                "import torch.nn as nn\nclass HyperMLR(nn.Module):\n    def __init__(self, ch, n, c=1.0):\n"
                "        super().__init__()\n        self.c, self.K, self.num_classes = c, c, n\n"
                "        import torch\n        self.P_MLR = nn.Parameter(torch.zeros(n, ch, dtype=torch.double))\n"
                "        self.A_MLR = nn.Parameter(torch.ones(n, ch, dtype=torch.double))\n")
    script = """
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tree!r})
import core.configs, core.utils.hyperbolic as ref_h, core.models.classifier as rc
head = rc.ASPP_Classifier_V2_Hyper(4, [6], [6], 16, 8)                          # built with the reference's own classes
import halo_amd
halo_amd.install()
import halo_amd.core.utils.hyperbolic as hh, halo_amd.core.models.classifier as hcls, halo_amd.core.configs as hc
import core.utils
assert core.utils.hyperbolic is hh and sys.modules['core.utils.hyperbolic'] is hh
assert hc.cfg is core.configs.cfg
assert rc.ASPP_Classifier_V2_Hyper.forward is hcls.v2_hyper_forward
mapper, seg = hcls._tail_modules(head)                                          # reference-built head: params shared, not copied
assert isinstance(mapper, hh.HyperMapper) and mapper.c == 0.7 and isinstance(seg, hh.HyperMLR)
assert seg.P_MLR is head.conv_seg.P_MLR and seg.A_MLR is head.conv_seg.A_MLR
assert sorted(head.state_dict()) == ['conv2d_list.0.bias', 'conv2d_list.0.weight', 'conv_seg.A_MLR', 'conv_seg.P_MLR']
print('late install ok')
""".format(root=ROOT, tree=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "late install ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


REAL = """
import sys, types
sys.path.insert(0, {root!r}); sys.path.insert(0, {shims!r}); sys.path.insert(0, '/root/reference')
import torch
import halo_amd
halo_amd.install()
import core.configs
# core/models/__init__ pulls torchvision / mmcv (absent here): import the classifier module alone
m = types.ModuleType('core.models'); m.__path__ = ['/root/reference/core/models']; sys.modules['core.models'] = m
import core.models.classifier as rc, core.active, core.utils.visualize as viz
import halo_amd.core.active.build as hb, halo_amd.core.active.floating_region as hf, halo_amd.core.models.classifier as hc
import halo_amd.core.configs as cfgmod
assert core.active.RegionSelection is hb.RegionSelection and viz.FloatingRegionScore is hf.FloatingRegionScore
assert cfgmod.cfg is core.configs.cfg
# the patched forwards run the reference-built heads' own conv bodies and hand the SAME tensor to the tail
# that the reference's forward maps with expmap (checked through the geoopt stand-in, CPU)
import geoopt.manifolds.stereographic.math as gmath
torch.manual_seed(0)
v3 = rc.DepthwiseSeparableASPP_Hyper(inplanes=32, dilation_series=[6, 12, 18], padding_series=[6, 12, 18], num_classes=19,
                                     norm_layer=torch.nn.BatchNorm2d, reduced_channels=8, hfr=True).eval()
v2 = rc.ASPP_Classifier_V2_Hyper(32, [6, 12], [6, 12], 19, 8).eval()
x = {{'out': torch.randn(2, 32, 6, 10), 'low': torch.randn(2, 256, 12, 20)}}
seen = {{}}
def fake_tail(feat, mapper, seg, size=None, resize_embed=False):
    seen['feat'], seen['resize_embed'], seen['size'] = feat, resize_embed, size
    return None, None
hc.hyper_head_tail = fake_tail
K = torch.tensor(-core.configs.cfg.MODEL.CURVATURE, dtype=torch.float64)
for head, resize in ((v3, False), (v2, True)):
    with torch.no_grad():
        head(x, size=(24, 40))
        got = gmath.project(gmath.expmap0(seen['feat'].double(), k=K, dim=1), k=K, dim=1)
        # the reference's own forward needs its own (geoopt-backed) hyperbolic classes: rebuild them on the shared parameters
        halo_amd.uninstall()
        for k in [k for k in sys.modules if k.startswith('core.utils.hyperbolic')]:
            del sys.modules[k]
        import core.utils.hyperbolic as ref_h
        head.mapper = ref_h.HyperMapper(c=core.configs.cfg.MODEL.CURVATURE)
        mlr = ref_h.HyperMLR(8, 19, c=core.configs.cfg.MODEL.CURVATURE)
        mlr.P_MLR, mlr.A_MLR = head.conv_seg.P_MLR, head.conv_seg.A_MLR
        head.conv_seg = mlr
        torch.Tensor.cuda = lambda self, *a, **k: self
        out, embed = type(head).forward(head, x, size=None)
        halo_amd.install()
    assert seen['resize_embed'] is resize and seen['size'] == (24, 40)
    assert torch.equal(got, embed), type(head).__name__
print('real tree ok')
"""


def test_install_against_the_real_reference_tree():
    """Build container only (skipped where /root/reference does not exist)."""
    import pytest
    if not os.path.isdir("/root/reference/core"):
        pytest.skip("reference tree not present")
    r = subprocess.run([sys.executable, "-c", REAL.format(root=ROOT, shims=os.path.join(ROOT, "tests", "golden", "_shims"))],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "real tree ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
