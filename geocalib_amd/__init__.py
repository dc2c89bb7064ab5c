"""geocalib_amd -- MI355X-native (gfx950) implementation of GeoCalib's LM calibration path.

Public surface (mirrors the reference's geocalib package for this path):
    LMOptimizer                         geocalib/lm_optimizer.py:141   (+ optimizer_step, update_lambda, early_stop)
    Camera models / camera_models       geocalib/camera.py
    Gravity                             geocalib/gravity.py
    GeoCalib (extractor with calibrate) geocalib/extractor.py:15  (CNN supplied by the caller)
"""
from .camera import BaseCamera, Pinhole, Radial, SimpleDivisional, SimpleRadial, camera_models  # noqa: F401
from .gravity import Gravity  # noqa: F401
from .lm_optimizer import LMOptimizer, get_trivial_estimation  # noqa: F401
from .extractor import GeoCalib  # noqa: F401,E402

__version__ = "0.1.0"
