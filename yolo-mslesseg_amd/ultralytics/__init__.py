"""Drop-in `ultralytics` import surface for srozenblum/YOLO-MSLesSeg, backed by mslesseg_amd (MI355X HIP kernels).

Put `yolo-mslesseg_amd/` on PYTHONPATH *instead of* installing ultralytics; the reference's own imports then
resolve here unchanged:
    from ultralytics import YOLO                                             [REF yolo_mslesseg/utils/utils.py:64]
    from ultralytics.utils import LOGGER                                     [REF yolo_mslesseg/scripts/train.py:78]
    from ultralytics.data.converter import convert_segment_masks_to_yolo_seg [REF yolo_mslesseg/scripts/extraer_dataset.py:82-83]
"""
from mslesseg_amd.yolo import YOLO  # noqa: F401

__version__ = "8.3.70+mslesseg.amd"
__all__ = ["YOLO"]
