"""Model descriptors + weights for the product path.

Descriptors are the JSON files under models/ (converted from the reference's .pdmodel data files by
tools/pdmodel_convert.py).  Real weights exist for V3_ch_det_fast only (every other .pdiparams blob is missing
from the reference checkout); for the other models `get_model` builds deterministic stand-in
weights with the same generator the oracle uses is NOT possible here (the product must not import oracle/),
so the generator lives here and the oracle imports nothing from it: both sides are handed the SAME numpy
weight dict by the caller (tests, bench) — the engine never invents weights on its own in production use.
"""
import json
import os

import numpy as np

MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


def load_descriptor(model_id):
    with open(os.path.join(MODELS_DIR, model_id + ".json")) as f:
        return json.load(f)


def load_weights_npz(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def has_real_weights(model_id):
    return os.path.exists(os.path.join(MODELS_DIR, model_id + ".npz"))


def random_weights(desc, seed=0):
    """Plain seeded He-style stand-in weights (uncalibrated) for throughput runs where values do not matter."""
    rng = np.random.default_rng(seed)
    out = {}
    bn_var = set()
    for op in desc["ops"]:
        if op["type"] == "batch_norm":
            bn_var.add(op["in"]["Variance"][0])
            bn_var.add(op["in"]["Scale"][0])
    for name in sorted(desc["params"]):
        dims = desc["params"][name]["dims"]
        if name in bn_var:
            a = rng.uniform(0.8, 1.2, dims)
        elif len(dims) == 4:
            a = rng.standard_normal(dims) * np.sqrt(1.0 / max(1, dims[1] * dims[2] * dims[3]))
        elif len(dims) == 2:
            a = rng.standard_normal(dims) * np.sqrt(1.0 / max(1, dims[0]))
        else:
            a = rng.standard_normal(dims) * 0.05
        out[name] = a.astype(np.float32)
    return out


def get_model(model_id, seed=0):
    desc = load_descriptor(model_id)
    p = os.path.join(MODELS_DIR, model_id + ".npz")
    if os.path.exists(p):
        return desc, load_weights_npz(p)
    return desc, random_weights(desc, seed)
