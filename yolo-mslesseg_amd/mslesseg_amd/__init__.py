"""mslesseg_amd — MI355X-native YOLO11-seg predict/train path behind the ultralytics `YOLO()` call surface
used by srozenblum/YOLO-MSLesSeg."""
__all__ = ["YOLO"]


def __getattr__(name):
    if name == "YOLO":
        from .yolo import YOLO

        return YOLO
    raise AttributeError(name)
