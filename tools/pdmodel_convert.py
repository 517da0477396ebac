#!/usr/bin/env python3
"""Build-time converter: Paddle inference artefacts -> this repo's own model descriptors.

Runs ONLY in the build container (it reads /root/reference/backend/models/**, which does not exist on
the GPU box).  It decodes the two Paddle *data formats* (SURVEY.md Appendix A):

  inference.pdmodel   protobuf ProgramDesc  -> <out>/<ver>_<name>.json   (op list, attrs, var shapes)
  inference.pdiparams raw tensor stream     -> <out>/<ver>_<name>.npz    (only where the blob exists)

The JSON descriptor is a compact, engine-specific restatement of the graph: per op its type, named
input/output slots and the handful of attributes the engine reads.  Nothing of the reference's Python
is copied; the .pdmodel/.pdiparams files are model data, and only the derived descriptors are committed.

usage: python tools/pdmodel_convert.py [--ref /root/reference] [--out video-subtitle-extractor_amd/models]
"""
import argparse
import json
import os
import struct
import sys

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
        npdt = _DT[dtype]
        cnt = int(np.prod(dims)) if dims else 1
        arr = np.frombuffer(buf, dtype=npdt, count=cnt, offset=pos).reshape(dims).copy()
        pos += cnt * np.dtype(npdt).itemsize
        out[name] = arr
    assert pos == len(buf), f"pdiparams not consumed to EOF ({pos} != {len(buf)})"
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
               "global_pooling", "padding_algorithm"],
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "video-subtitle-extractor_amd", "models"))
    args = ap.parse_args()
    base = os.path.join(args.ref, "backend", "models")
    os.makedirs(args.out, exist_ok=True)
    index = {}
    for ver in sorted(os.listdir(base)):
        vdir = os.path.join(base, ver)
        if not os.path.isdir(vdir):
            continue
        for name in sorted(os.listdir(vdir)):
            mdir = os.path.join(vdir, name)
            pdm = os.path.join(mdir, "inference.pdmodel")
            if not os.path.exists(pdm):
                continue
            mid = f"{ver}_{name}"
            blk = parse_program(pdm)
            desc = build_descriptor(blk, mid)
            with open(os.path.join(args.out, mid + ".json"), "w") as f:
                json.dump(desc, f, separators=(",", ":"))
            nparam = sum(int(np.prod(p["dims"])) for p in desc["params"].values())
            has_w = False
            pdi = os.path.join(mdir, "inference.pdiparams")
            if os.path.exists(pdi):
                tensors = parse_params(pdi, sorted(desc["params"].keys()))
                for n, a in tensors.items():
                    assert list(a.shape) == list(desc["params"][n]["dims"]), (n, a.shape, desc["params"][n])
                np.savez_compressed(os.path.join(args.out, mid + ".npz"), **tensors)
                has_w = True
            index[mid] = {"ops": len(desc["ops"]), "params": nparam, "weights": has_w}
            print(f"{mid:28s} ops={len(desc['ops']):4d} params={nparam/1e6:7.3f}M weights={'yes' if has_w else 'MISSING'}")
    with open(os.path.join(args.out, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
