"""Minimal stand-in for ``yacs.config.CfgNode`` (fixture generation ONLY).

The reference builds its global ``cfg`` with yacs (core/configs/defaults.py:3),
which is not installed in this image.  Attribute-style nested dict is all the
hot path reads (floating_region.py:39,68; build.py:23,75-81).
"""


class CfgNode(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        self[name] = value

    def set_new_allowed(self, flag):  # misc.py:155
        pass

    def freeze(self):
        pass

    def defrost(self):
        pass
