import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from vse_amd import engine, modelzoo, pipeline, synth
ctx = engine.Context(0)
frames = torch.from_numpy(synth.make_frames(64, 1080, 1920, seed=1)).cuda()
for mid, hilo in (("V4_ch_det", False), ("V3_ch_det_fast", True)):
    desc, w = modelzoo.get_model(mid, seed=0)
    for fuse in (False, True):
        net = engine.Net(ctx, desc, w, fetch_cols=(0,), hilo=hilo, input_norm=pipeline.DET_NORM, fuse_preprocess=fuse)
        net.det_forward(frames, 544, 960); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for _ in range(5):
            ev[0].record(); net.det_forward(frames, 544, 960); ev[1].record(); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
        ms, prog, names = net.profile_frames(frames, 544, 960)
        ms2, _, _ = net.profile_frames(frames, 544, 960)
        print(mid, "fused" if fuse else "pass+plan", "det_forward %.3f ms (min of 5)" % min(ts), "| plan ops %.3f ms, stem op %.3f ms (%s)" % (min(ms.sum(), ms2.sum()), min(ms[0], ms2[0]), names[0]))
