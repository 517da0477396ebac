#!/usr/bin/env python3
"""Long streamed runs of the product pipeline (`OcrPipeline.ocr_stream` + `parallel.gather_records`): one rank's share of a long
job, generator-fed, with subtitles that keep changing (line count, width) so that the recogniser meets ever new (crops, width) plan
keys — what a film stresses: the plan cache, the per-plan workspaces (LRU under VSE_WS_BUDGET_GB), device memory, the steady-state
frame rate.

* BASELINE configs[4] (C5): 2 h x 24 fps = 172 800 frames of 1080p over 8 ranks -> rank 0's contiguous range
  `parallel.shard_range(172800, 0, 8)` = 21 600 frames: `--frames 21600` (tests/test_gpu_stream.py runs exactly this).
* BASELINE configs[3] (C4): one 4 096-frame 1080p clip per rank: `--frames 4096`.
* soak: `--batches 600 --models server|fast`, summary as JSON (`--json PATH`).

Frames come from a small resident pool of synthetic batches; every batch of the stream is a fresh device tensor spliced from two
pool entries at a random cut (so consecutive batches differ in line count and widths).  The server pair runs on stand-in weights:
as in bench.py, the detector's map is max-overlaid (on the detector's stream) with the text-kernel map of the generator's lines,
so DB post-processing and the recogniser do real work on every frame; `--models fast-real` uses the one real-weight detector and
no overlay.

usage: python tools/soak.py [--frames N | --batches B] [--batch 64] [--models server|fast|fast-real] [--height 1080 --width 1920] [--json out.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run_stream(ctx, models="server", total_frames=21600, batch=64, height=1080, width=1920, pool_batches=4, seed=0, depth=2,
               rec_span=2, first_frame=0, log=None, ws_budget_gb=None):
    """-> (records sorted by frame number, summary dict, sources).  Frame numbers run from first_frame (a rank's shard start);
    sources[k] = (pool entry, frame in it) of streamed frame k: the same source frame must give the same record whatever batch it
    rides in (the detector works per image, the ragged recogniser is bit-identical across batch compositions)."""
    import gc
    import torch
    import bench
    from vse_amd import modelzoo, parallel, pipeline, shim, synth
    gc.collect()                       # an earlier pipeline of this process (its closures form cycles) would count into the peak
    torch.cuda.empty_cache()
    det_id, rec_id, lang = {"server": ("V4_ch_det", "V4_ch_rec", "ch"), "fast": ("V4_ch_det_fast", "V4_ch_rec_fast", "ch"),
                            "fast-real": ("V3_ch_det_fast", "V4_en_rec_fast", "en")}[models]
    det, rec = modelzoo.get_model(det_id, seed=0), modelzoo.get_model(rec_id, seed=1)
    standin = not modelzoo.has_real_weights(det_id)
    if standin:
        det = (det[0], bench.empty_det_head(det[0], det[1]))
    charset = shim.standin_charset(lang, shim._ncls(rec[0]))
    pipe = pipeline.OcrPipeline(ctx, det, rec, charset, rec_mode="ragged", bucket=256, batch_round=4, min_rec_group=8)   # bench.py's settings
    pipe.rec_streams = 4
    pipe.rec_stream_priority = -1
    if ws_budget_gb is not None:       # the recogniser's workspace LRU (engine.Net._workspace; default VSE_WS_BUDGET_GB = 64)
        pipe.rec.ws_budget = int(ws_budget_gb * (1 << 30))
    mh, mw = pipeline.det_resize_shape(height, width, pipe.limit)
    pool, overlays, lines = [], [], []
    for k in range(pool_batches):
        fr, truth = synth.make_frames(batch, height, width, seed=1000 + 17 * k + seed, p_two_lines=0.1 + 0.8 * k / max(1, pool_batches - 1),
                                      return_truth=True)
        pool.append(torch.from_numpy(fr).to(ctx.tdev))
        lines.append([len(t) for t in truth])
        if standin:
            overlays.append(torch.from_numpy(bench.text_kernel_maps(truth, height, width, mh, mw, unclip_ratio=pipe.db["unclip_ratio"])).to(ctx.tdev))
    rng = np.random.default_rng(seed)
    pending = {}                       # data_ptr of a streamed batch -> its overlay (consumed by the detector wrapper)
    plain_det_maps = pipe.det_maps

    def det_maps(frames, slot=0):
        maps = plain_det_maps(frames, slot=slot)
        ov = pending.pop(frames.data_ptr(), None)
        if ov is not None:
            ov.record_stream(torch.cuda.current_stream(ctx.tdev))      # spliced on the main stream, consumed on the detector's
            torch.maximum(maps, ov, out=maps)       # scaffolding for stand-in weights only, on the detector's stream (as in bench.py)
        return maps
    pipe.det_maps = det_maps
    expect_lines = [0]
    sources = []

    def batches():
        left = total_frames
        while left > 0:
            n = min(batch, left)
            i, j = (int(v) for v in rng.integers(len(pool), size=2))
            cut = int(rng.integers(0, n + 1))
            fr = torch.cat([pool[i][:cut], pool[j][cut:n]])
            if standin:
                pending[fr.data_ptr()] = torch.cat([overlays[i][:cut], overlays[j][cut:n]])
            expect_lines[0] += sum(lines[i][:cut]) + sum(lines[j][cut:n])
            sources.extend([(i, f) for f in range(cut)] + [(j, f) for f in range(cut, n)])
            left -= n
            yield fr

    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    records, marks = [], []
    nframes = nlines = 0
    t0 = time.perf_counter()
    for k, out in enumerate(pipe.ocr_stream(batches(), depth=depth, rec_span=rec_span)):
        for boxes, res in out:
            records.append((first_frame + nframes, np.asarray(boxes, np.float32).reshape(-1, 4, 2), res))
            nframes += 1
            nlines += len(boxes)
        # one mark per finished batch: (frames done, seconds, plans, workspaces, bytes allocated) — host clock, no device sync added
        marks.append((nframes, time.perf_counter() - t0, len(pipe.rec.plans), len(pipe.rec.ws), torch.cuda.memory_allocated()))
        if log and nframes % (50 * batch) == 0:
            log(f"frames {nframes:6d}: {nframes / marks[-1][1]:7.1f} frames/s since start, rec plans {marks[-1][2]:3d}, "
                f"rec workspaces {marks[-1][3]:3d}, device memory {marks[-1][4] / 1e9:.2f} GB")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gathered = parallel.gather_records(records, device=ctx.tdev)        # THE exchange step of the path (local sort on one rank)

    def rate(lo, hi):                  # frames/s between the marks at fractions lo..hi of the stream
        a = marks[min(len(marks) - 1, int(lo * len(marks)))]
        b = marks[min(len(marks) - 1, max(int(hi * len(marks)) - 1, 0))]
        return (b[0] - a[0]) / max(b[1] - a[1], 1e-9)
    third = len(marks) // 3
    summary = {
        "models": f"{det_id} + {rec_id}", "frames": nframes, "frame_size": [height, width], "batch": batch, "text_lines": nlines,
        "text_lines_expected": expect_lines[0], "wall_s": round(wall, 3), "frames_per_s": round(nframes / wall, 1),
        "frames_per_s_first_third": round(rate(0.02, 1 / 3), 1), "frames_per_s_last_third": round(rate(2 / 3, 1.0), 1),
        "frames_per_s_first_decile": round(rate(0.02, 0.1), 1), "frames_per_s_last_decile": round(rate(0.9, 1.0), 1),
        "rec_plans": marks[-1][2], "rec_plans_at_one_third": marks[third][2], "rec_workspaces": marks[-1][3],
        "workspace_evictions": int(getattr(pipe.rec, "ws_evictions", 0) + getattr(pipe.det, "ws_evictions", 0)),
        "memory_allocated_gb": round(marks[-1][4] / 1e9, 3), "memory_allocated_at_one_third_gb": round(marks[third][4] / 1e9, 3),
        "memory_peak_gb": round(torch.cuda.max_memory_allocated() / 1e9, 3),
        "rec_workspace_budget_gb": round(pipe.rec.ws_budget / (1 << 30), 1),
        "rec_workspace_gb": round(sum(int(w[0].untyped_storage().nbytes()) for w in pipe.rec.ws.values()) / 1e9, 3),
        "det_workspace_gb": round(sum(int(w[0].untyped_storage().nbytes()) for w in pipe.det.ws.values()) / 1e9, 3),
        "pool_gb": round((sum(int(p.numel()) for p in pool) + sum(4 * int(o.numel()) for o in overlays)) / 1e9, 3),
        "det_plans": len(pipe.det.plans), "streaming": f"ocr_stream(depth={depth}, rec_span={rec_span})",
    }
    pipe.det_maps = plain_det_maps
    return gathered, summary, sources


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--batches", type=int, default=None)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--models", default="server", choices=["server", "fast", "fast-real"])
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--pool", type=int, default=4)
    ap.add_argument("--json", default=None)
    ap.add_argument("--ws-budget-gb", type=float, default=None)
    a = ap.parse_args()
    from vse_amd import engine
    total = a.frames if a.frames is not None else (a.batches or 300) * a.batch
    ctx = engine.Context(0)
    recs, summary, _src = run_stream(ctx, a.models, total, a.batch, a.height, a.width, pool_batches=a.pool, ws_budget_gb=a.ws_budget_gb,
                               log=lambda m: print(m, file=sys.stderr, flush=True))
    assert [r[0] for r in recs] == list(range(total))
    print(json.dumps(summary))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
