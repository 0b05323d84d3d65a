"""MI355X-native linearised-Laplace / inducing-point posterior engine.

Same call surface as the hot path of nrholm1/Laplace-Inducing-Points (``src/ggn.py``,
``src/lla.py``, ``src/sample.py``, ``src/stochtrace.py``); the arithmetic runs in
hand-written HIP kernels for gfx950 behind a C-ABI library (``include/lip.h``).
The directory name carries a hyphen, so the package is importable as ``lip_amd``
and — for drop-in use with the reference's ``from src.ggn import ...`` — as ``src``.
"""
__version__ = "0.1.0"
