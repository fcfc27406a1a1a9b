"""ctypes binding of libmslesseg_hip.so (include/mslesseg_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parents[1] / "lib" / "libmslesseg_hip.so"

MSL_BF16, MSL_F32, MSL_F32S = 0, 1, 2  # MSL_F32S: fp32 tensors, split-precision conv products on the f16 matrix cores (include/mslesseg_hip.h)
PRED_STRIDE = 40
ABI_VERSION = 2  # include/mslesseg_hip.h MSL_ABI_VERSION
LANE_MAIN_FREE = 0x10000  # lane word (msl_run_program_lanes): on the caller's stream, as one of the open region's chains (MSL_LANE_MAIN_FREE)

OP_CONV, OP_STEM, OP_DWCONV, OP_SPPF_POOL, OP_UPSAMPLE2X, OP_ATTENTION = 1, 2, 3, 4, 5, 6
OP_HEAD_DECODE, OP_NMS, OP_MASK_LOWRES, OP_MASK_UPSAMPLE, OP_MASK_MERGE, OP_LETTERBOX = 7, 8, 9, 10, 11, 12
OP_VOL_INSERT, OP_VOL_CONSENSUS, OP_VOL_DICE = 13, 14, 15
OP_BN_STATS, OP_BN_FINALIZE, OP_BN_ACT, OP_BN_ACT_BWD_REDUCE, OP_BN_ACT_BWD_APPLY = 16, 17, 18, 19, 20
OP_COLSUM, OP_F64_DRAIN, OP_ADD_VIEW, OP_UPSAMPLE2X_BWD, OP_SPPF_POOL_BWD = 21, 22, 23, 24, 25
OP_CONV_WGRAD, OP_DW_WGRAD, OP_STEM_WGRAD, OP_CAST_PAD, OP_GATHER_CAST, OP_ADAMW, OP_EMA = 26, 27, 28, 29, 30, 31, 32
OP_SEG_LOSS, OP_ATTENTION_BWD, OP_SLICE_EXTRACT, OP_SGD, OP_AUGMENT, OP_RASTER_MASKS, OP_MASK_IOU = 33, 34, 35, 36, 37, 38, 39

EXPORTS = (
    "msl_abi_version", "msl_last_error", "msl_launch", "msl_run_program", "msl_run_program_lanes", "msl_graph_create", "msl_graph_create_lanes", "msl_graph_launch",
    "msl_graph_destroy", "msl_event_create", "msl_event_record", "msl_event_elapsed_ms", "msl_event_destroy",
    "msl_seg_loss_workspace", "msl_conv2d_nhwc", "msl_letterbox_u8", "msl_nms", "msl_volume_consensus", "msl_volume_dice_sums",
    "msl_conv2d_wgrad_nhwc", "msl_bn_act_fwd", "msl_bn_act_bwd", "msl_seg_loss", "msl_adamw", "msl_input_table_supported", "msl_lane_stamps",
)


class MslOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("dtype", C.c_int32),
        ("p", C.c_void_p * 12),
        ("i", C.c_int32 * 32),
        ("f", C.c_float * 4),
    ]


class MslError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise MslError(
                f"{LIB_PATH} is missing: build it with `python -m mslesseg_amd.build` "
                "(there is no CPU fallback for the product path)"
            )
        L = C.CDLL(str(LIB_PATH))
        L.msl_abi_version.restype = C.c_int
        L.msl_last_error.restype = C.c_char_p
        L.msl_launch.argtypes = [C.POINTER(MslOp), C.c_void_p]
        L.msl_run_program.argtypes = [C.POINTER(MslOp), C.c_int32, C.c_void_p]
        L.msl_run_program_lanes.argtypes = [C.POINTER(MslOp), C.POINTER(C.c_int32), C.c_int32, C.c_void_p]
        L.msl_graph_create.argtypes = [C.POINTER(MslOp), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.msl_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.msl_graph_destroy.argtypes = [C.c_void_p]
        L.msl_event_create.argtypes = [C.POINTER(C.c_void_p)]
        L.msl_event_record.argtypes = [C.c_void_p, C.c_void_p]
        L.msl_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.msl_event_destroy.argtypes = [C.c_void_p]
        L.msl_input_table_supported.argtypes = [C.POINTER(MslOp)]
        L.msl_lane_stamps.argtypes = [C.POINTER(C.c_float), C.c_int32]
        L.msl_seg_loss_workspace.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.msl_seg_loss_workspace.restype = C.c_int64
        # typed entry points (plain arguments; the Python host itself goes through descriptors — these are exercised by tests/test_gpu_capi_typed.py)
        L.msl_conv2d_nhwc.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 10 + [C.c_void_p]
        L.msl_letterbox_u8.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 11 + [C.c_void_p]
        L.msl_nms.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_float, C.c_void_p]
        L.msl_volume_consensus.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_void_p]
        L.msl_volume_dice_sums.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]
        if L.msl_abi_version() != ABI_VERSION:
            raise MslError(f"ABI version mismatch: library {L.msl_abi_version()}, binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise MslError(f"{what or 'libmslesseg_hip'} failed ({rc}): {lib().msl_last_error().decode(errors='replace')}")


def make_op(kind: int, dtype: int, p=(), i=None, f=()) -> MslOp:
    op = MslOp()
    op.kind, op.dtype = kind, dtype
    for k, v in enumerate(p):
        op.p[k] = v if v else None
    if i:
        for k, v in i.items():
            op.i[k] = int(v)
    for k, v in enumerate(f):
        op.f[k] = float(v)
    return op


def launch(op: MslOp, stream: int) -> None:
    check(lib().msl_launch(C.byref(op), C.c_void_p(stream)), f"op kind {op.kind}")


class Program:
    """A fixed list of ops (device pointers baked in) enqueued with one host call, optionally as a hipGraph."""

    def __init__(self, ops, lanes=None):
        self.n = len(ops)
        self.arr = (MslOp * self.n)(*ops)
        self.lanes = (C.c_int32 * self.n)(*lanes) if lanes is not None and any(lanes) else None  # per-op side-stream lane (0 = caller's stream)
        self._graph = None

    def run(self, stream: int) -> None:
        if self.lanes is not None:
            check(lib().msl_run_program_lanes(self.arr, self.lanes, self.n, C.c_void_p(stream)), "msl_run_program_lanes")
        else:
            check(lib().msl_run_program(self.arr, self.n, C.c_void_p(stream)), "msl_run_program")

    def capture(self, stream: int) -> None:
        g = C.c_void_p()
        if self.lanes is not None:
            check(lib().msl_graph_create_lanes(self.arr, self.lanes, self.n, C.c_void_p(stream), C.byref(g)), "msl_graph_create_lanes")
        else:
            check(lib().msl_graph_create(self.arr, self.n, C.c_void_p(stream), C.byref(g)), "msl_graph_create")
        self._graph = g

    def replay(self, stream: int) -> None:
        if self._graph is None:
            self.capture(stream)
        check(lib().msl_graph_launch(self._graph, C.c_void_p(stream)), "msl_graph_launch")

    def __del__(self):
        try:
            if self._graph is not None and _lib is not None:
                _lib.msl_graph_destroy(self._graph)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        check(lib().msl_event_create(C.byref(self.h)), "msl_event_create")

    def record(self, stream: int) -> None:
        check(lib().msl_event_record(self.h, C.c_void_p(stream)), "msl_event_record")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        check(lib().msl_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "msl_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if _lib is not None:
                _lib.msl_event_destroy(self.h)
        except Exception:
            pass
