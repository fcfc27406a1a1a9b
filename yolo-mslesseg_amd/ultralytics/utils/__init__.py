"""`ultralytics.utils.LOGGER` — the reference only calls `.setLevel(...)` on it [REF scripts/train.py:98]."""
from mslesseg_amd.yolo import LOGGER  # noqa: F401
