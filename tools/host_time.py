#!/usr/bin/env python3
"""Where the HOST spends a step of the pipeline (one batch at a time, wall clock around each stage, GPU drained between stages so
that waits are not mistaken for host work): detector launch, wait for the detector, DB post-processing (device CCL + host geometry +
device scoring), recognition (crop specs, grouping, launches, final read-back + string decode).
usage: python tools/host_time.py [--models server|fast-real] [--batch 64]"""
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from vse_amd import engine, modelzoo, pipeline, shim, synth

models = sys.argv[sys.argv.index("--models") + 1] if "--models" in sys.argv else "server"
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 64
ctx = engine.Context(0)
det_id, rec_id, lang = {"server": ("V4_ch_det", "V4_ch_rec", "ch"), "fast": ("V4_ch_det_fast", "V4_ch_rec_fast", "ch")}.get(models, ("V3_ch_det_fast", "V4_en_rec_fast", "en"))
det, rec = modelzoo.get_model(det_id, seed=0), modelzoo.get_model(rec_id, seed=1)
overlay = None
frames_np, truth = synth.make_frames(batch, 1080, 1920, seed=100, return_truth=True)
if not modelzoo.has_real_weights(det_id):
    det = (det[0], bench.empty_det_head(det[0], det[1]))
    overlay = torch.from_numpy(bench.text_kernel_maps(truth, 1080, 1920, 544, 960)).cuda()
pipe = pipeline.OcrPipeline(ctx, det, rec, shim.standin_charset(lang, shim._ncls(rec[0])), bucket=256, batch_round=4, min_rec_group=8)
pipe.rec_streams = 4
frames = torch.from_numpy(frames_np).cuda()


def step(T=None):
    t0 = time.perf_counter()
    maps = pipe.det_maps(frames)
    if overlay is not None:
        torch.maximum(maps, overlay, out=maps)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    db = ctx.db_postprocess(maps, 1080, 1920, **pipe.db); t3 = time.perf_counter()
    boxes = [pipeline.sorted_boxes(b[0]) for b in db]; t4 = time.perf_counter()
    specs = pipe._crop_specs(boxes); groups = pipe._groups(specs); t5 = time.perf_counter()
    res = pipe.recognize(frames, boxes); t6 = time.perf_counter()
    if T is not None:
        T.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, sum(len(b) for b in boxes)))


for _ in range(3):
    step()
T = []
for _ in range(6):
    step(T)
a = np.array(T).mean(0)
print(f"{models}, {batch} frames, {a[6]:.0f} boxes per step — host ms: detector launch {1e3 * a[0]:.2f} | wait for the detector (GPU) {1e3 * a[1]:.2f} | "
      f"DB post-process (device passes + host geometry) {1e3 * a[2]:.2f} | box sort {1e3 * a[3]:.2f} | crop specs + grouping {1e3 * a[4]:.2f} | "
      f"recognize (the same again + launches + read-back + decode; GPU time included) {1e3 * a[5]:.2f}")
