"""Paddle inference data formats read without Paddle (paddle_io.py) and the shim's model-directory loading
(det_model_dir / rec_model_dir of backend/tools/ocr.py:93-99 are DIRECTORIES holding inference.pdmodel + inference.pdiparams)."""
import os
import struct

import numpy as np
import pytest

from vse_amd import modelzoo, paddle_io, shim

REF_MODELS = "/root/reference/backend/models"


# ---- a minimal protobuf writer for framework.proto's ProgramDesc (only what a tiny test graph needs) -------------------
def _vi(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno, payload):
    return _vi(fno << 3 | 2) + _vi(len(payload)) + payload


def _iv(fno, v):
    return _vi(fno << 3) + _vi(v)


def _var(name, dims=None, persistable=False):
    body = _ld(1, name.encode())
    if dims is not None:
        tdesc = _iv(1, 5) + b"".join(_iv(2, d) for d in dims)                # FP32
        body += _ld(2, _iv(1, 7) + _ld(3, _ld(1, tdesc)))                    # VarType{LOD_TENSOR, lod_tensor{tensor}}
    return body + _iv(3, int(persistable))


def _op(typ, ins, outs, attrs=()):
    body = b""
    for k, names in ins.items():
        body += _ld(1, _ld(1, k.encode()) + b"".join(_ld(2, n.encode()) for n in names))
    for k, names in outs.items():
        body += _ld(2, _ld(1, k.encode()) + b"".join(_ld(2, n.encode()) for n in names))
    body += _ld(3, typ.encode())
    for name, atype, val in attrs:
        a = _ld(1, name.encode()) + _iv(2, atype)
        if atype == 0:
            a += _iv(3, val)
        elif atype == 1:
            a += _vi(4 << 3 | 5) + struct.pack("<f", val)
        elif atype == 2:
            a += _ld(5, val.encode())
        elif atype == 3:
            a += b"".join(_iv(6, x) for x in val)
        body += _ld(4, a)
    return body


def _tiny_model(tmp_path, with_params=True):
    d = tmp_path / "V9" / "tiny_det"
    d.mkdir(parents=True)
    vars_ = [_var("feed"), _var("fetch"), _var("x", [-1, 3, -1, -1]), _var("conv.w", [8, 3, 3, 3], True),
             _var("conv.b", [8], True), _var("y", [-1, 8, -1, -1]), _var("z", [-1, 8, -1, -1])]
    ops = [_op("feed", {"X": ["feed"]}, {"Out": ["x"]}, [("col", 0, 0)]),
           _op("conv2d", {"Input": ["x"], "Filter": ["conv.w"]}, {"Output": ["y"]},
               [("strides", 3, [1, 1]), ("paddings", 3, [1, 1]), ("dilations", 3, [1, 1]), ("groups", 0, 1),
                ("padding_algorithm", 2, "EXPLICIT"), ("data_format", 2, "NCHW"), ("use_cudnn", 0, 1)]),
           _op("elementwise_add", {"X": ["y"], "Y": ["conv.b"]}, {"Out": ["z"]}, [("axis", 0, 1)]),
           _op("fetch", {"X": ["z"]}, {"Out": ["fetch"]}, [("col", 0, 0)])]
    block = _iv(1, 0) + _iv(2, -1) + b"".join(_ld(3, v) for v in vars_) + b"".join(_ld(4, o) for o in ops)
    (d / "inference.pdmodel").write_bytes(_ld(1, block))
    rng = np.random.default_rng(3)
    w = {"conv.w": rng.standard_normal((8, 3, 3, 3)).astype(np.float32), "conv.b": rng.standard_normal(8).astype(np.float32)}
    if with_params:
        paddle_io.write_params(str(d / "inference.pdiparams"), w)
    return d, w


def test_params_stream_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    t = {"b.w_0": rng.standard_normal((4, 3, 3, 3)).astype(np.float32), "a.b_0": rng.standard_normal(7).astype(np.float32),
         "c.idx": np.arange(5, dtype=np.int64), "d.scalar": np.float32(2.5).reshape(())}
    path = str(tmp_path / "inference.pdiparams")
    paddle_io.write_params(path, t)
    back = paddle_io.parse_params(path, sorted(t))
    assert list(back) == sorted(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and back[k].shape == t[k].shape and (back[k] == t[k]).all()
    with open(path, "ab") as f:                     # trailing bytes = a different tensor list than the graph's: refuse
        f.write(b"\0\0\0\0")
    with pytest.raises(ValueError, match="bytes left"):
        paddle_io.parse_params(path, sorted(t))


def test_model_directory_is_read_directly(tmp_path):
    d, w = _tiny_model(tmp_path)
    desc, weights = paddle_io.load_model_dir(str(d))
    assert [op["type"] for op in desc["ops"]] == ["feed", "conv2d", "elementwise_add", "fetch"]
    conv = desc["ops"][1]
    assert conv["in"] == {"Input": ["x"], "Filter": ["conv.w"]} and conv["attrs"]["paddings"] == [1, 1]
    assert conv["attrs"]["padding_algorithm"] == "EXPLICIT" and "use_cudnn" not in conv["attrs"]      # export noise dropped
    assert desc["params"] == {"conv.w": {"dims": [8, 3, 3, 3], "dtype": 5}, "conv.b": {"dims": [8], "dtype": 5}}
    assert desc["var_shapes"]["x"] == [-1, 3, -1, -1]
    assert set(weights) == set(w) and all((weights[k] == w[k]).all() for k in w)
    # the shim's loader: the directory itself (what the reference passes) and an id resolved under config.models_root
    old = shim.config.models_root
    try:
        desc2, weights2 = shim._load_model(str(d))
        assert desc2["ops"] == desc["ops"] and (weights2["conv.w"] == w["conv.w"]).all()
        shim.config.models_root = str(tmp_path)
        desc3, weights3 = shim._load_model("V9_tiny_det")
        assert desc3["model"] == "V9_tiny_det" and (weights3["conv.b"] == w["conv.b"]).all()
    finally:
        shim.config.models_root = old


def test_model_directory_without_blob_fails_loudly(tmp_path):
    d, _ = _tiny_model(tmp_path, with_params=False)
    assert paddle_io.load_model_dir(str(d))[1] is None
    with pytest.raises(FileNotFoundError, match="pdiparams"):
        shim._load_model(str(d))
    bad = tmp_path / "V9" / "tiny_det" / "inference.pdiparams"
    paddle_io.write_params(str(bad), {"conv.w": np.zeros((8, 3, 3, 2), np.float32), "conv.b": np.zeros(8, np.float32)})
    with pytest.raises(ValueError, match="conv.w"):
        shim._load_model(str(d))


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="the reference checkout exists only in the build container")
def test_reference_model_files_give_the_committed_descriptors():
    """Every graph of the reference's backend/models decodes to the descriptor committed under models/, and the one weight
    blob it ships (V3/ch_det_fast) to the committed tensors."""
    n = 0
    for ver in sorted(os.listdir(REF_MODELS)):
        for name in sorted(os.listdir(os.path.join(REF_MODELS, ver))):
            d = os.path.join(REF_MODELS, ver, name)
            if not os.path.exists(os.path.join(d, "inference.pdmodel")):
                continue
            desc, weights = paddle_io.load_model_dir(d, f"{ver}_{name}")
            assert desc == modelzoo.load_descriptor(f"{ver}_{name}")
            if weights is not None:
                ref = modelzoo.load_weights_npz(os.path.join(modelzoo.MODELS_DIR, f"{ver}_{name}.npz"))
                assert set(ref) == set(weights) and all((ref[k] == weights[k]).all() for k in ref)
            n += 1
    assert n == 21
