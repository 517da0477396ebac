"""__graft_entry__.smoke(): one tiny det+rec invocation of the hot path on cuda:0, checked against the oracle."""
import os
import sys

import numpy as np


def smoke():
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import net_ref, pipeline_ref
    from vse_amd import engine, pipeline, synth
    ctx = engine.Context(0)
    det = net_ref.get_weights("V3_ch_det_fast")       # real weights
    rec = net_ref.get_weights("V4_en_rec_fast")       # calibrated stand-in weights (blob missing in the reference)
    charset = pipeline_ref.en_charset()
    frames = synth.make_frames(2, 360, 640, seed=3)
    pipe = pipeline.OcrPipeline(ctx, det, rec, charset)
    dev = torch.from_numpy(frames).cuda()
    got = pipe.ocr(dev)
    torch.cuda.synchronize()
    nbox = 0
    for f in range(len(frames)):
        def det_fn(x):
            return net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]

        def rec_fn(x):
            return net_ref.run_graph(rec[0], rec[1], x)[0].numpy()
        rb, rr = pipeline_ref.text_system(frames[f], det_fn, rec_fn, charset)
        gb, gr = got[f]
        assert len(gb) == len(rb), f"frame {f}: {len(gb)} boxes vs oracle {len(rb)}"
        for a, b in zip(gb, rb):
            assert np.abs(np.asarray(a) - np.asarray(b)).max() <= 1.0, (a, b)
        nbox += len(gb)
        for (gt, gs), (rt, rs) in zip(gr, rr):
            assert abs(gs - rs) < 5e-2, (gs, rs)
    assert nbox > 0, "smoke frames must contain detectable text"
    print(f"smoke ok: {len(frames)} frames, {nbox} boxes match the oracle; libvse_hip.so loaded from {engine.LIB_PATH}")
