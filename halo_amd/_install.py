"""`halo_amd.install()` -- make the reference tree import the HIP path under its OWN module names.

The reference imports its acquisition path as `core.active.build`, `core.active.floating_region`,
`core.utils.hyperbolic`, `core.loss.local_consistent_loss`, `core.loss.negative_learning_loss`
(core/train_learners.py:12,16-17; core/utils/visualize.py:6; core/models/classifier.py:4;
core/active/build.py:12; core/active/floating_region.py:9).  After `install()` those names resolve
to halo_amd's modules, so the learner, the heads and the visualiser run unchanged:

    import halo_amd; halo_amd.install()          # first lines of train.py / test.py
    from core.train_learners import ...          # unchanged

What it does
  1. serves halo_amd.core.{active.build, active.floating_region, utils.hyperbolic,
     loss.local_consistent_loss, loss.negative_learning_loss} under the `core.*` names through a
     sys.meta_path finder (parents `core`, `core.active`, ... still come from the reference tree);
     reference modules of those names that were imported earlier are replaced in sys.modules and on
     their parent packages;
  2. when `core.configs` is importable, makes halo_amd read the reference's ONE cfg object
     (halo_amd.core.configs.use) -- otherwise that happens at the first import of core.configs;
  3. patches the `forward` of core.models.classifier's two hyperbolic head classes with the
     HIP-tail forwards (halo_amd.core.models.classifier.patch_reference_heads), now if the module
     is already imported, else right after its import.
`uninstall()` undoes 1 and 3.
"""
import importlib
import importlib.abc
import importlib.util
import sys

ALIASES = {
    "core.active.build": "halo_amd.core.active.build",
    "core.active.floating_region": "halo_amd.core.active.floating_region",
    "core.utils.hyperbolic": "halo_amd.core.utils.hyperbolic",
    "core.loss.local_consistent_loss": "halo_amd.core.loss.local_consistent_loss",
    "core.loss.negative_learning_loss": "halo_amd.core.loss.negative_learning_loss",
}
_POST_IMPORT = ("core.models.classifier", "core.configs")
_saved = {}
_finder = None


def _after_import(name, module):
    if name == "core.models.classifier":
        from .core.models.classifier import patch_reference_heads
        patch_reference_heads(module)
    elif name == "core.configs" and hasattr(module, "cfg"):
        from .core import configs
        configs.use(module.cfg)


class _PostImportFinder(importlib.abc.MetaPathFinder):
    """Lets the normal finders locate core.models.classifier / core.configs, then runs _after_import once
    the module body has executed."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname in ALIASES:              # served from halo_amd; the import system then also binds it on its parent
            return importlib.util.spec_from_loader(fullname, _Alias(ALIASES[fullname]))
        if fullname not in _POST_IMPORT:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None and hasattr(spec.loader, "exec_module"):
                spec.loader = _Wrap(spec.loader, fullname)
                return spec
        return None


class _Alias(importlib.abc.Loader):
    """Loader that hands out an existing halo_amd module under a `core.*` name."""

    def __init__(self, real):
        self.real = real
        self._spec = None

    def create_module(self, spec):
        mod = importlib.import_module(self.real)
        self._spec = getattr(mod, "__spec__", None)
        return mod

    def exec_module(self, module):
        # importlib's module_from_spec has just overwritten the module's __spec__ with the alias' spec: put the real one
        # back, so that __spec__.parent == __package__ (relative imports executed later inside the module, importlib.reload)
        if self._spec is not None:
            module.__spec__ = self._spec


class _Wrap(importlib.abc.Loader):
    def __init__(self, inner, name):
        self.inner, self.name = inner, name

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        _after_import(self.name, module)

    def __getattr__(self, item):                 # get_source / is_package / get_filename ... for tracebacks
        return getattr(self.inner, item)


def install():
    """Idempotent.  Returns the list of `core.*` module names now served by halo_amd."""
    global _finder
    if _finder is None:                          # fresh imports of the aliased names are answered by the finder
        _finder = _PostImportFinder()
        sys.meta_path.insert(0, _finder)
    for alias, real in ALIASES.items():
        mod = importlib.import_module(real)
        if alias in sys.modules and sys.modules[alias] is not mod:     # the reference's module was imported earlier
            _saved.setdefault(alias, sys.modules[alias])
            sys.modules[alias] = mod
            parent, _, leaf = alias.rpartition(".")
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
    # `from core.active import RegionSelection` (core/active/__init__.py star-imports build)
    act = sys.modules.get("core.active")
    if act is not None:
        build = importlib.import_module(ALIASES["core.active.build"])
        for k in ("RegionSelection", "select_pixels_to_label", "to_np_array"):
            if hasattr(act, k):
                setattr(act, k, getattr(build, k))
    for name in _POST_IMPORT:
        if name in sys.modules:
            _after_import(name, sys.modules[name])
    if "core.configs" not in sys.modules:
        try:                                      # bind the reference's cfg now when its tree is importable
            spec = importlib.util.find_spec("core.configs")
        except (ImportError, ValueError):
            spec = None
        if spec is not None:
            try:
                importlib.import_module("core.configs")
            except ImportError:
                pass                              # e.g. yacs missing: the stand-in cfg stays until use() is called
    return sorted(ALIASES)


def uninstall():
    global _finder
    for alias, real in ALIASES.items():
        if alias in sys.modules and sys.modules[alias] is sys.modules.get(real):
            old = _saved.get(alias)
            if old is None:
                del sys.modules[alias]
            else:
                sys.modules[alias] = old
            parent, _, leaf = alias.rpartition(".")
            if parent in sys.modules and getattr(sys.modules[parent], leaf, None) is sys.modules.get(real):
                if old is None:
                    delattr(sys.modules[parent], leaf)
                else:
                    setattr(sys.modules[parent], leaf, old)
    _saved.clear()
    if _finder is not None and _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    _finder = None
    mod = sys.modules.get("core.models.classifier")
    if mod is not None:
        for name in ("ASPP_Classifier_V2_Hyper", "DepthwiseSeparableASPP_Hyper"):
            cls = getattr(mod, name, None)
            ref = getattr(cls, "_reference_forward", None) if cls is not None else None
            if ref is not None:
                cls.forward = ref
                del cls._reference_forward
