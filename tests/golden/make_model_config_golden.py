#!/usr/bin/env python3
"""Generates tests/golden/model_config.json by RUNNING the reference's backend/tools/paddle_model_config.py
(PaddleModelConfig.__init__, :8-106) for every (language, mode, accelerator) combination against the reference's own
backend/models tree.  Stubs: fsplit (Filesplit.merge is a no-op: the split weight blobs are missing anyway) and
backend.config (plain values).  Only inputs/outputs are written.  Needs /root/reference."""
import importlib.util
import json
import os
import sys
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_config.json")


class _Val:
    def __init__(self, v):
        self.value = v


def main():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Filesplit:
        def merge(self, input_dir=None):
            pass
    mod("fsplit")
    mod("fsplit.filesplit", Filesplit=Filesplit)
    cfg = types.SimpleNamespace(language=_Val("ch"), mode=_Val("fast"))
    mod("backend")
    mod("backend.config", BASE_DIR=os.path.join(REF, "backend"), config=cfg)
    spec = importlib.util.spec_from_file_location("pmc_ref", os.path.join(REF, "backend/tools/paddle_model_config.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    class HA:
        def __init__(self, acc):
            self.acc = acc
            self.onnx_providers = []

        def has_accelerator(self):
            return self.acc

    probe = m.PaddleModelConfig(HA(False))
    langs = list(probe.MULTI_LANG) + ["xx_unknown"]
    rows = []
    base = os.path.join(REF, "backend", "models")
    for lang in langs:
        for mode in ("fast", "auto", "accurate"):
            for acc in (False, True):
                cfg.language.value, cfg.mode.value = lang, mode
                try:
                    c = m.PaddleModelConfig(HA(acc))
                    rows.append([lang, mode, acc, os.path.relpath(c.DET_MODEL_PATH, base).replace(os.sep, "_"),
                                 os.path.relpath(c.REC_MODEL_PATH, base).replace(os.sep, "_"), c.MODEL_VERSION,
                                 c.REC_IMAGE_SHAPE, None])
                except Exception as e:                  # e.g. os.listdir on a model dir that does not exist
                    rows.append([lang, mode, acc, None, None, None, None, type(e).__name__])
    with open(OUT, "w") as f:
        json.dump({"source": "backend/tools/paddle_model_config.py @ v2.2.0 executed against backend/models",
                   "columns": ["language", "mode", "accelerator", "det", "rec", "version", "rec_image_shape", "error"],
                   "rows": rows}, f, separators=(",", ":"))
    print("wrote", OUT, len(rows), "rows;", sum(r[7] is not None for r in rows), "raise")


if __name__ == "__main__":
    main()
