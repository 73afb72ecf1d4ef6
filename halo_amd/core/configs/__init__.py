"""`cfg` stand-in for the keys the acquisition path reads.

The reference keeps one global yacs CfgNode (core/configs/__init__.py:1, defaults.py) that the
hot path reads at construction time (floating_region.py:39,68; build.py:75-81;
classifier.py:361-362).  yacs is not required here: `cfg` is an attribute dict holding the same
keys with the reference's defaults.  Inside the reference tree call `use(core.configs.cfg)` once
 so both packages see one object.
"""


class CfgNode(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        self[name] = value


def _defaults():
    c = CfgNode()
    c.MODEL = CfgNode(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0, REDUCED_CHANNELS=64)   # defaults.py:9-15
    c.ACTIVE = CfgNode(UNCERTAINTY="entropy", PURITY="hyper", SELECT_ITER=[0, 15000, 30000, 40000, 50000],
                       BUDGET=0.05, RADIUS_K=1, NORMALIZE=True, MASK_RADIUS_K=5, K=100,
                       VIZ_MASK=False)                                                    # defaults.py:64-79
    c.SEED = -1
    return c


cfg = _defaults()


def use(other):
    """Make `cfg` read through to the reference's own config object."""
    global cfg
    cfg = other
    from ..active import floating_region, build
    from ..utils import hyperbolic  # noqa: F401
    floating_region.cfg = other
    build.cfg = other
    import sys
    heads = sys.modules.get(__name__.rsplit(".", 1)[0] + ".models.classifier")
    if heads is not None:
        heads.cfg = other
    return other
