"""Host logic on CPU (-m "not gpu"): geometry tables vs the oracle, checkpoint format, graph/param specs,
the C-ABI library loads and exports every symbol the header declares, and the product path fails loudly
(no CPU fallback) when there is no GPU."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from mslesseg_amd import geometry, graph, hiplib, params

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built_lib():
    from mslesseg_amd import build

    return build.build()


# ------------------------------------------------------------------------------------------- C-ABI
def test_library_exports_every_declared_symbol(built_lib):
    header = (ROOT / "include" / "mslesseg_hip.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(msl_\w+)\s*\(", header, flags=re.M))
    assert declared == set(hiplib.EXPORTS), declared ^ set(hiplib.EXPORTS)
    lib = ctypes.CDLL(str(built_lib))
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert hiplib.lib().msl_abi_version() == hiplib.ABI_VERSION == 2


def test_no_kernel_uses_the_k16_f16_matrix_instruction():
    """`v_mfma_f32_16x16x16_f16` runs at half rate on gfx950 and hipcc schedules VALU accesses behind it as for a 4-pass instruction: the 8-wave
    split-precision kernels that used it were intermittently wrong.  Single K-steps go through the K = 32 form with a zero
    second step (`msl_mfma_split`); nothing in the library's sources may bring the K = 16 f16 / bf16 forms back."""
    import re
    from pathlib import Path

    src = Path(__file__).resolve().parents[1] / "yolo-mslesseg_amd" / "csrc"
    bad = []
    for f in sorted(src.glob("*.h*")):
        code = re.sub(r"//[^\n]*", "", f.read_text())  # comments may name the instruction
        if re.search(r"mfma_f32_16x16x16_?(f16|bf16)", code):
            bad.append(f.name)
    assert not bad, bad


def test_op_struct_layout_matches_header(built_lib):
    # int32 kind, dtype; 12 pointers; 32 int32; 4 floats  → 8 + 96 + 128 + 16 (ABI version 2)
    assert ctypes.sizeof(hiplib.MslOp) == 248
    header = (ROOT / "include" / "mslesseg_hip.h").read_text()
    assert re.search(r"void\* p\[12\];\s*int32_t i\[32\];\s*float f\[4\];", header), "msl_op layout in the header changed: update hiplib.MslOp with it"
    header = (ROOT / "include" / "mslesseg_hip.h").read_text()
    kinds = dict(re.findall(r"^\s*(MSL_OP_\w+)\s*=\s*(\d+)", header, flags=re.M))
    for name, val in kinds.items():
        assert getattr(hiplib, name[4:]) == int(val), name
    assert int(re.search(r"#define MSL_PRED_STRIDE (\d+)", header).group(1)) == hiplib.PRED_STRIDE


def test_descriptor_validation_needs_no_gpu(built_lib):
    """Bad descriptors are rejected on the host before any launch."""
    op = hiplib.make_op(hiplib.OP_CONV, hiplib.MSL_BF16, p=(1, 1, 1, 0, 1),
                        i={0: 1, 1: 8, 2: 8, 3: 12, 4: 8, 5: 8, 6: 16, 7: 3, 8: 1, 9: 1, 10: 12, 12: 16, 16: 108, 17: 128, 21: 16})
    with pytest.raises(hiplib.MslError, match="multiples of 8"):
        hiplib.launch(op, 0)
    with pytest.raises(hiplib.MslError, match="unknown op kind"):
        hiplib.launch(hiplib.make_op(99, 0), 0)
    op = hiplib.make_op(hiplib.OP_NMS, 0, p=(1, 1, 1, 1), i={0: 1, 6: 20000, 7: 300})
    with pytest.raises(hiplib.MslError, match="nms"):
        hiplib.launch(op, 0)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu(golden_dir, tmp_path):
    from ultralytics import YOLO

    st = torch.load(golden_dir / "synth_n_nc1.pt", map_location="cpu", weights_only=True)
    params.save_checkpoint(tmp_path / "best.pt", st, "n", 1)
    m = YOLO(tmp_path / "best.pt")
    with pytest.raises(hiplib.MslError, match="no CPU fallback"):
        m(np.zeros((182, 182, 3), np.uint8), verbose=False)


# ------------------------------------------------------------------------------------------- geometry
@pytest.mark.parametrize("hw", [(218, 182), (182, 182), (182, 218), (640, 640), (37, 91), (700, 500), (33, 1000)])
def test_letterbox_geometry_matches_oracle(hw):
    from oracle import prepost as P

    lb = geometry.letterbox_for(*hw)
    nw, nh, top, bottom, left, right = P.letterbox_geometry(*hw)
    assert (lb.wn, lb.hn, lb.top, lb.left, lb.hlb, lb.wlb) == (nw, nh, top, left, nh + top + bottom, nw + left + right)
    assert lb.hlb % 32 == 0 and lb.wlb % 32 == 0


@pytest.mark.parametrize("src,dst", [((218, 182), (640, 534)), ((182, 218), (534, 640)), ((37, 91), (260, 640)), ((700, 500), (640, 457))])
def test_linear_tables_reproduce_oracle_resize(src, dst):
    from oracle import prepost as P

    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    want = P.cv_resize_linear_u8(img, (dst[1], dst[0]))
    xt = geometry.linear_table(dst[1], src[1], True).astype(np.int64)
    yt = geometry.linear_table(dst[0], src[0], False).astype(np.int64)
    a = img.astype(np.int64)
    rows = a[:, xt[:, 0]] * xt[:, 2][None, :, None] + a[:, xt[:, 1]] * xt[:, 3][None, :, None]
    S0, S1 = rows[yt[:, 0]], rows[yt[:, 1]]
    got = (((yt[:, 2][:, None, None] * (S0 >> 4)) >> 16) + ((yt[:, 3][:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(np.clip(got, 0, 255).astype(np.uint8), want)


@pytest.mark.parametrize("dst,src", [(218, 640), (182, 544), (182, 640), (218, 544), (7, 5), (5, 7)])
def test_nearest_table_matches_oracle(dst, src):
    from oracle import prepost as P

    assert np.array_equal(geometry.nearest_table(dst, src), P.nearest_index_table(dst, src))


# ------------------------------------------------------------------------------------------- params / graph
@pytest.mark.parametrize("scale,nc,want", [("n", 80, 2876848), ("n", 1, 2842803), ("s", 80, 10113248), ("s", 1, 10082675)])
def test_product_param_count(scale, nc, want):
    assert params.count_params(scale, nc) == want


@pytest.mark.parametrize("scale", ["n", "s", "m"])
def test_product_specs_equal_oracle_state_dict(scale):
    from oracle import yolo11seg as Y

    sd = Y.build(scale, 1).state_dict()
    shapes = params.tensor_shapes(scale, 1)
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)


def test_checkpoint_roundtrip_and_safe_loader(tmp_path):
    st = params.init_state("n", 1, seed=3)
    p = tmp_path / "trains" / "x" / "fold1" / "weights" / "best.pt"
    params.save_checkpoint(p, st, "n", 1, {0: "lesion"})
    ck = params.load_checkpoint(p)
    assert ck["scale"] == "n" and ck["nc"] == 1 and ck["names"][0] == "lesion"
    assert all(torch.equal(ck["state"][k], st[k]) for k in st)
    params.validate_state(ck["state"], "n", 1)
    torch.save({k: v for k, v in st.items()}, tmp_path / "bare.pt")  # bare state_dict dump (INTEGRATION.md converter)
    ck2 = params.load_checkpoint(tmp_path / "bare.pt")
    assert (ck2["scale"], ck2["nc"]) == ("n", 1)
    bad = dict(st)
    bad["model.0.conv.weight"] = torch.zeros(8, 3, 3, 3)
    with pytest.raises(ValueError):
        params.validate_state(bad, "n", 1)


def test_init_state_bias_init_and_determinism():
    import math

    a, b = params.init_state("n", 1, seed=0), params.init_state("n", 1, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    for i, s in enumerate((8, 16, 32)):
        assert float(a[f"model.23.cv2.{i}.2.bias"][0]) == 1.0
        assert abs(float(a[f"model.23.cv3.{i}.2.bias"][0]) - math.log(5 / 1 / (640 / s) ** 2)) < 1e-6


def test_folded_conv_bn_matches_oracle_fuse(golden_dir):
    from oracle import synth

    st = torch.load(golden_dir / "synth_n_nc1.pt", map_location="cpu", weights_only=True)
    st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
    fused = synth.model_from_state(st)
    specs = params.param_specs("n", 1)
    for name in ("model.1", "model.8.m.0.m.1.cv2", "model.10.m.0.attn.pe", "model.23.cv3.1.0.0", "model.23.proto.cv2"):
        w, b = params.folded(st, name, specs[name])
        mod = fused
        for part in name.split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        assert torch.allclose(w, mod.conv.weight, rtol=1e-5, atol=1e-7) and torch.allclose(b, mod.conv.bias, rtol=1e-5, atol=1e-6)


def test_on_load_scan_decides_from_the_library(built_lib):
    """trainprog.OnLoadScan (BatchNorm on load, round 4) asks the library which readers honour an input BatchNorm table (msl_input_table_supported: host code, no
    GPU needed) and leaves a layer pending only when every reader does: no layer with a residual of its own, no layer read by a pool / upsample / attention /
    depthwise conv / the loss, none read through the generic kernel (the one-channel class head); and the reader-pass bound is respected."""
    from mslesseg_amd import graph, trainprog
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32

    sc = trainprog.OnLoadScan(128, 640, 640, MSL_BF16, 2.0)
    graph.walk(sc, "n", 1)
    pend = sc.pending()
    bn_layers = [p for p in sc.prod if p["bn"]]
    assert len(bn_layers) == 90 and len(pend) == 56
    for p in sc.prod:
        if p["res"] or not p["bn"]:
            assert p["name"] not in pend
    for must_not in ("model.9.cv1", "model.10.m.0.attn.qkv", "model.10.cv2", "model.13.cv2", "model.16.cv2", "model.22.cv2", "model.23.proto.cv3", "model.23.cv3.0.1.1",
                     "model.23.cv2.0.0"):  # pool, attention, upsample, multi-reader pyramid features, the loss, the generic class head, the persistent 3x3 reader
        assert must_not not in pend, must_not
    for must in ("model.0", "model.1", "model.2.cv1", "model.2.m.0.cv1", "model.3", "model.23.proto.cv2", "model.23.cv3.0.0.0"):
        assert must in pend, must
    tight = trainprog.OnLoadScan(128, 640, 640, MSL_BF16, 1.0)
    graph.walk(tight, "n", 1)
    assert "model.2.cv1" not in tight.pending() and "model.1" in tight.pending()  # cv1's output is read 0.5 + 0.5 + 1 times
    f32 = trainprog.OnLoadScan(8, 64, 64, MSL_F32, 2.0)
    graph.walk(f32, "n", 1)
    assert not f32.pending()  # a bf16 form
