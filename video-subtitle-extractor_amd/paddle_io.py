"""Readers for the two Paddle inference DATA formats — no Paddle needed.

  inference.pdmodel    protobuf ProgramDesc -> parse_program() -> build_descriptor(): the engine's model descriptor
                       (per op its type, named input / output slots and the handful of attributes the engine reads)
  inference.pdiparams  raw tensor stream    -> parse_params(): {name: ndarray} in the sorted order of the persistable names

Used at build time by tools/pdmodel_convert.py (the descriptors under models/) and at run time by the shim when it is pointed at a
Paddle model directory (shim._load_model: det_model_dir / rec_model_dir of the reference's call sites, ocr.py:91-113), so a
user's own inference.pdiparams are read directly.  write_params() writes the same stream (tests, exporting stand-in weights).
"""
import struct

import numpy as np


# ----------------------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field_no, wire_type, value) for one message.  value: int | bytes."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _sint64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _sint32(v):
    v &= 0xFFFFFFFFFFFFFFFF
    v = _sint64(v)
    return v


def _packed_varints(b):
    out = []
    pos = 0
    while pos < len(b):
        v, pos = _varint(b, pos)
        out.append(_sint64(v))
    return out


# attr type enum (framework.proto AttrType)
_INT, _FLOAT, _STRING, _INTS, _FLOATS, _STRINGS, _BOOLEAN, _BOOLEANS, _BLOCK, _LONG, _BLOCKS, _LONGS = range(12)
_FLOAT64 = 15


def _parse_attr(buf):
    name = None
    atype = None
    i = f = s = b = l = f64 = None
    ints, floats, strings, bools, longs = [], [], [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = v.decode()
        elif fno == 2:
            atype = v
        elif fno == 3:
            i = _sint32(v)
        elif fno == 4:
            f = struct.unpack("<f", v)[0]
        elif fno == 5:
            s = v.decode(errors="replace")
        elif fno == 6:
            ints += _packed_varints(v) if wt == 2 else [_sint32(v)]
        elif fno == 7:
            floats += list(struct.unpack(f"<{len(v)//4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno == 8:
            strings.append(v.decode(errors="replace"))
        elif fno == 10:
            b = bool(v)
        elif fno == 11:
            bools += [bool(x) for x in (_packed_varints(v) if wt == 2 else [v])]
        elif fno == 13:
            l = _sint64(v)
        elif fno == 15:
            longs += _packed_varints(v) if wt == 2 else [_sint64(v)]
        elif fno == 19:
            f64 = struct.unpack("<d", v)[0]
    val = {_INT: i, _FLOAT: f, _STRING: s, _INTS: ints, _FLOATS: floats, _STRINGS: strings, _BOOLEAN: b,
           _BOOLEANS: bools, _LONG: l, _LONGS: longs, _FLOAT64: f64}.get(atype)
    return name, val


def _parse_opvar(buf):
    param = None
    args = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            param = v.decode()
        elif fno == 2:
            args.append(v.decode())
    return param, args


def _parse_op(buf):
    op = {"type": None, "inputs": {}, "outputs": {}, "attrs": {}}
    for fno, wt, v in _fields(buf):
        if fno == 3:
            op["type"] = v.decode()
        elif fno == 1:
            p, a = _parse_opvar(v)
            op["inputs"][p] = a
        elif fno == 2:
            p, a = _parse_opvar(v)
            op["outputs"][p] = a
        elif fno == 4:
            n, val = _parse_attr(v)
            op["attrs"][n] = val
    return op


def _parse_tensor_desc(buf):
    dtype = None
    dims = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dtype = v
        elif fno == 2:
            dims += _packed_varints(v) if wt == 2 else [_sint64(v)]
    return dtype, dims


def _parse_var(buf):
    var = {"name": None, "persistable": False, "dtype": None, "dims": None}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            var["name"] = v.decode()
        elif fno == 3:
            var["persistable"] = bool(v)
        elif fno == 2:  # VarType
            for f2, _, v2 in _fields(v):
                if f2 == 3:  # LoDTensorDesc
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            var["dtype"], var["dims"] = _parse_tensor_desc(v3)
    return var


def parse_program(path):
    buf = open(path, "rb").read()
    blocks = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            blk = {"vars": [], "ops": []}
            for f2, _, v2 in _fields(v):
                if f2 == 3:
                    blk["vars"].append(_parse_var(v2))
                elif f2 == 4:
                    blk["ops"].append(_parse_op(v2))
            blocks.append(blk)
    assert len(blocks) == 1, "all inference graphs here are single-block"
    return blocks[0]


# ----------------------------------------------------------------------------- pdiparams
_DT = {5: np.float32, 3: np.int64, 2: np.int32, 6: np.float64}


def parse_params(path, names_sorted):
    buf = open(path, "rb").read()
    pos = 0
    out = {}
    for name in names_sorted:
        (ver,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        (lod,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        for _ in range(lod):
            (nb,) = struct.unpack_from("<Q", buf, pos)
            pos += 8 + nb
        (tver,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        (dlen,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        dtype, dims = _parse_tensor_desc(buf[pos:pos + dlen])
        pos += dlen
        if dtype not in _DT:
            raise ValueError(f"{path}: tensor {name} has unsupported Paddle dtype code {dtype}")
        npdt = _DT[dtype]
        cnt = int(np.prod(dims)) if dims else 1
        arr = np.frombuffer(buf, dtype=npdt, count=cnt, offset=pos).reshape(dims).copy()
        pos += cnt * np.dtype(npdt).itemsize
        out[name] = arr
    if pos != len(buf):
        raise ValueError(f"{path}: {len(buf) - pos} bytes left after the {len(names_sorted)} tensors the graph declares "
                         "(the weight file belongs to a different graph)")
    return out


# ----------------------------------------------------------------------------- descriptor
# attributes the engine reads, per op type (everything else is export noise)
_KEEP = {
    "conv2d": ["strides", "paddings", "dilations", "groups", "padding_algorithm", "data_format"],
    "depthwise_conv2d": ["strides", "paddings", "dilations", "groups", "padding_algorithm", "data_format"],
    "conv2d_transpose": ["strides", "paddings", "dilations", "groups", "padding_algorithm", "output_padding",
                         "output_size", "data_format"],
    "batch_norm": ["epsilon", "data_layout"],
    "pool2d": ["pooling_type", "ksize", "strides", "paddings", "ceil_mode", "exclusive", "adaptive",
               "global_pooling", "padding_algorithm", "data_format"],
    "hard_swish": ["offset", "scale", "threshold"],
    "hard_sigmoid": ["slope", "offset"],
    "swish": ["beta"],
    "elementwise_add": ["axis"],
    "elementwise_mul": ["axis"],
    "nearest_interp_v2": ["scale", "out_h", "out_w", "align_corners", "interp_method", "data_layout"],
    "layer_norm": ["epsilon", "begin_norm_axis"],
    "softmax": ["axis"],
    "scale": ["scale", "bias", "bias_after_scale"],
    "matmul_v2": ["trans_x", "trans_y"],
    "matmul": ["transpose_X", "transpose_Y", "alpha"],
    "transpose2": ["axis"],
    "reshape2": ["shape"],
    "slice": ["axes", "starts", "ends", "decrease_axis", "infer_flags"],
    "concat": ["axis"],
    "squeeze2": ["axes"],
    "flatten_contiguous_range": ["start_axis", "stop_axis"],
    "dropout": ["dropout_implementation", "is_test", "dropout_prob"],
    "fill_constant": ["shape", "value", "str_value", "dtype"],
    "fill_constant_batch_size_like": ["shape", "value", "input_dim_idx", "output_dim_idx", "dtype"],
    "rnn": ["mode", "num_layers", "is_bidirec", "hidden_size", "input_size", "is_test"],
    "feed": ["col"],
    "fetch": ["col"],
    "shape": [],
    "assign": [],
    "relu": [],
    "sigmoid": [],
}


def build_descriptor(block, model_id):
    vars_ = {v["name"]: v for v in block["vars"]}
    ops = []
    for op in block["ops"]:
        t = op["type"]
        keep = _KEEP.get(t)
        if keep is None:
            raise SystemExit(f"{model_id}: op type {t!r} not in the closed operator set (SURVEY App. E)")
        attrs = {k: op["attrs"][k] for k in keep if k in op["attrs"] and op["attrs"][k] is not None}
        ops.append({
            "type": t,
            "in": {k: v for k, v in op["inputs"].items() if v},
            "out": {k: v for k, v in op["outputs"].items() if v},
            "attrs": attrs,
        })
    params = {}
    for v in block["vars"]:
        if v["persistable"] and v["name"] not in ("feed", "fetch"):
            params[v["name"]] = {"dims": v["dims"], "dtype": v["dtype"]}
    var_shapes = {n: v["dims"] for n, v in vars_.items() if v["dims"] is not None and not v["persistable"]}
    return {"model": model_id, "ops": ops, "params": params, "var_shapes": var_shapes}


def write_params(path, tensors):
    """{name: ndarray} -> inference.pdiparams stream (sorted names, LoD level 0, the TensorDesc proto of parse_params)."""
    inv = {np.dtype(v): k for k, v in _DT.items()}

    def varint(v):
        v &= (1 << 64) - 1
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)
    with open(path, "wb") as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name], order="C")
            desc = b"\x08" + varint(inv[a.dtype]) + b"".join(b"\x10" + varint(int(d)) for d in a.shape)
            f.write(struct.pack("<IQI", 0, 0, 0) + struct.pack("<i", len(desc)) + desc + a.tobytes())


def load_model_dir(path, model_id=None):
    """A Paddle inference model directory -> (descriptor, weights | None): the graph from inference.pdmodel, the tensors from
    inference.pdiparams when that file exists (the reference checkout ships most models without it)."""
    import os
    pdm = os.path.join(path, "inference.pdmodel")
    import json
    # through JSON like the committed descriptors (tuples -> lists, one representation for both routes)
    desc = json.loads(json.dumps(build_descriptor(parse_program(pdm), model_id or os.path.basename(os.path.normpath(path)))))
    pdi = os.path.join(path, "inference.pdiparams")
    if not os.path.exists(pdi):
        return desc, None
    weights = parse_params(pdi, sorted(desc["params"].keys()))
    for n, a in weights.items():
        if list(a.shape) != list(desc["params"][n]["dims"]):
            raise ValueError(f"{pdi}: tensor {n} has shape {list(a.shape)}, the graph declares {desc['params'][n]['dims']}")
    return desc, weights
