#!/usr/bin/env python3
"""Large-map check (BASELINE configs[2] stress form): 4K frames with det_limit_side_len = 3840 -> 2176 x 3840 detector input (16x the
reference's 544 x 960).  Mobile detector vs the CPU oracle on one frame; server detector: runs, finite, timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import net_ref, pipeline_ref as P
from vse_amd import engine, pipeline, synth

ctx = engine.Context(0)
frames = synth.make_frames(2, 2160, 3840, seed=3)
dev = torch.from_numpy(frames).cuda()
rec = net_ref.get_weights("V4_en_rec_fast")
for mid, check in (("V3_ch_det_fast", True), ("V4_ch_det", False)):
    det = net_ref.get_weights(mid)
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), limit_side_len=3840, det_weights="fp16")
    maps = pipe.det_maps(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    maps = pipe.det_maps(dev)
    torch.cuda.synchronize()
    dt = time.time() - t0
    m = maps.cpu().numpy()
    print(mid, "map", m.shape, "finite", bool(np.isfinite(m).all()), "range", float(m.min()), float(m.max()), f"{dt * 1e3:.1f} ms for 2 frames")
    if check:
        x, _ = P.det_preprocess(frames[0], 3840)
        ref = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
        d = np.abs(m[0] - ref)
        print("  vs oracle: max |dp|", float(d.max()), "bitmap disagreement", float(((m[0] > 0.3) != (ref > 0.3)).mean()))
        got = ctx.db_postprocess(maps[:1], 2160, 3840)[0][0]
        want, _ = P.db_postprocess(ref, 2160, 3840)
        print("  boxes", len(got), len(want), "identical", len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(pipeline.sorted_boxes(got), P.sorted_boxes(want))))
