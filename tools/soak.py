#!/usr/bin/env python3
"""Soak run of the streamed pipeline over a long synthetic video whose subtitles keep changing (line count, width, position): the
recogniser meets ever new (crops, width) plan keys, so this watches what a real film would stress — the plan cache, the per-plan
workspaces (LRU under VSE_WS_BUDGET_GB), device memory and the steady-state frame rate.
usage: python tools/soak.py [--batches 300] [--batch 16] [--models fast|server]"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vse_amd import engine, modelzoo, pipeline, shim, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=300)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--models", default="fast")
    a = ap.parse_args()
    ctx = engine.Context(0)
    det_id, rec_id = ("V3_ch_det_fast", "V4_en_rec_fast") if a.models == "fast" else ("V4_ch_det", "V4_ch_rec")
    det, rec = modelzoo.get_model(det_id, seed=0), modelzoo.get_model(rec_id, seed=1)
    charset = shim.standin_charset("en" if a.models == "fast" else "ch", shim._ncls(rec[0]))
    pipe = pipeline.OcrPipeline(ctx, det, rec, charset, batch_round=4)
    pipe.rec_streams = 2
    pool = [synth.make_frames(a.batch, 720, 1280, seed=1000 + k, p_two_lines=0.1 + 0.8 * (k % 5) / 4) for k in range(12)]
    rng = np.random.default_rng(0)

    def batches():
        for k in range(a.batches):
            fr = pool[int(rng.integers(len(pool)))].copy()
            if rng.random() < 0.3:                      # a stretch without subtitles
                fr[:, int(0.7 * fr.shape[1]):] = fr[:, :int(0.3 * fr.shape[1]) + 1][:, :fr.shape[1] - int(0.7 * fr.shape[1])]
            if rng.random() < 0.5:                      # shorter lines: blank the right part of the subtitle band
                cut = int(rng.integers(fr.shape[2] // 2, fr.shape[2]))
                fr[:, int(0.75 * fr.shape[1]):, cut:] = 40
            yield torch.from_numpy(fr).to(ctx.tdev, non_blocking=True)
    t0 = time.time()
    nframes = nlines = 0
    marks = []
    for k, out in enumerate(pipe.ocr_stream(batches(), depth=2, rec_span=2)):
        nframes += len(out)
        nlines += sum(len(r[1]) for r in out)
        if (k + 1) % 50 == 0:
            torch.cuda.synchronize()
            marks.append((k + 1, round(nframes / (time.time() - t0), 1), len(pipe.rec.plans), len(pipe.rec.ws),
                          round(torch.cuda.memory_allocated() / 1e9, 2), round(torch.cuda.max_memory_allocated() / 1e9, 2)))
            print("batches %4d: %7.1f frames/s since start, rec plans %3d, rec workspaces %3d, device memory %.2f GB (peak %.2f)" % marks[-1],
                  flush=True)
    torch.cuda.synchronize()
    print(f"soak: {nframes} frames, {nlines} text lines in {time.time() - t0:.1f} s; rec plan keys: {sorted(pipe.rec.plans)[:6]} ... ({len(pipe.rec.plans)})")


if __name__ == "__main__":
    main()
