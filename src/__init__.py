"""Drop-in alias so that the reference's imports (``from src.ggn import compute_ggn_vp``,
``from src.lla import ...``, ``from src.sample import ...``, ``from src.stochtrace import ...``)
resolve to the MI355X-native package in ``laplace-inducing-points_amd/``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "laplace-inducing-points_amd")]
