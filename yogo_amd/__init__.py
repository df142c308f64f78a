"""yogo_amd -- MI355X-native (gfx950, hand-written HIP) implementation of the YOGO hot path.

Drop-in surface for the path (reference: czbiohub-sf/yogo): ``yogo_amd.model.YOGO``, ``yogo_amd.model_defns``
(``MODELS`` / ``register_model`` / ``get_model_func``), ``yogo_amd.yogo_loss.YOGOLoss``,
``yogo_amd.utils.format_preds``.  The compute lives in ``yogo_amd/lib/libyogo_hip.so`` (C ABI: include/yogo_hip.h).
"""
from yogo_amd import model, model_defns, utils, yogo_loss  # noqa: F401

__version__ = "0.1.0"
