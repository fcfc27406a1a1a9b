"""`convert_segment_masks_to_yolo_seg` (B5b) [REF yolo_mslesseg/scripts/extraer_dataset.py:215-227]."""
from mslesseg_amd.labels import convert_segment_masks_to_yolo_seg  # noqa: F401
