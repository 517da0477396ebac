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
import sys

import numpy as np


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vse_amd.paddle_io import build_descriptor, parse_params, parse_program      # noqa: E402


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
