"""Drop-in alias so that the reference's imports (``from src.ggn import compute_ggn_vp``,
``from src.lla import ...``, ``from src.sample import ...``, ``from src.stochtrace import ...``)
resolve to the MI355X-native package in ``laplace-inducing-points_amd/`` (the same module objects as
``lip_amd.*``)."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
import lip_amd as _pkg  # noqa: E402

__path__ = list(_pkg.__path__)
for _m in ("utils", "netspec", "toymodels", "scalemodels", "ggn", "lla", "sample", "stochtrace", "krylov", "train_alpha", "evaluate", "train_inducing", "checkpoint"):
    _mod = _il.import_module("lip_amd." + _m)
    _sys.modules["src." + _m] = _mod
    globals()[_m] = _mod
