#!/usr/bin/env python3
"""Which stored activations' fp16 rounding moves the real-weight detector's map across the 0.3 threshold (the residue of a8: 3 of
136 boxes at 1080p are 2 px off)?  CPU only: the engine program of V3_ch_det_fast (hi + lo weights, exact raw input) on the CPU
emulator with fp32 activations everywhere EXCEPT a chosen range of ops whose outputs are rounded to fp16, against the fp32 oracle.
usage: python tools/act_rounding_study.py [frame index in the seed-777 sweep, default 31] [height 1080] [width 1920]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ir_emul, net_ref, pipeline_ref as P
from vse_amd import compiler, pipeline, synth

pos = [a for a in sys.argv[1:] if not a.startswith("-")]
f = int(pos[0]) if len(pos) > 0 else 31
h = int(pos[1]) if len(pos) > 1 else 1080
w = int(pos[2]) if len(pos) > 2 else 1920
frame = synth.make_frames(f + 1, h, w, seed=777, p_two_lines=0.5)[f]
desc, wts = net_ref.get_weights("V3_ch_det_fast")
x, _ = P.det_preprocess(frame)
ref = net_ref.run_graph(desc, wts, x)[0].numpy()[0, 0]
rh, rw = x.shape[2], x.shape[3]
raw = np.zeros((1, rh, rw, 8), np.float32)
raw[0, ..., :3] = P.cv2_resize_linear_u8(frame, rw, rh).astype(np.float32)
raw[..., 3] = 1.0
prog = compiler.compile_model(desc, wts, 1, rh, rw, hilo=True, input_norm=pipeline.DET_NORM)
n = len(prog.ops)
print(f"frame {f}: {n} ops; oracle map: {int((ref > 0.3).sum())} pixels above 0.3")


def study(label, ops):
    got = ir_emul.Emulator(prog, round_ops=ops).run(raw)[0][0, ..., 0]
    d = np.abs(got - ref)
    print(f"{label:42s} max |dp| {d.max():.4f}  flipped pixels {int(((got > 0.3) != (ref > 0.3)).sum())}", flush=True)
    return got


q = n // 4
if "--bisect" not in sys.argv:
    study("fp32 activations everywhere", [])
    study("fp16 storage everywhere", range(n))
    for a, b in ((0, q), (q, 2 * q), (2 * q, 3 * q), (3 * q, n)):
        study(f"fp16 storage of ops {a}..{b - 1} only ({prog.names[a][:14]} .. {prog.names[b - 1][:14]})", range(a, b))

if "--bisect" in sys.argv:
    lo, hi = 0, q
    for a in range(lo, hi, 3):
        b = min(a + 3, hi)
        study(f"ops {a}..{b - 1}: " + ", ".join(f"{prog.names[k][:12]}[{int(prog.ops[k]['out']['h'])}x{int(prog.ops[k]['out']['c'])}]" for k in range(a, b)), range(a, b))
