#!/usr/bin/env python3
"""Whole-clip throughput of the callers' row: extractor.SubtitleExtractor.run() (frame selection -> batched OCR -> raw.txt
filters -> SRT) on a synthetic 1080p clip held in HOST memory (so the PCIe upload of every frame is inside the timed
region), real-weight detector (V3_ch_det_fast) + stand-in en recogniser, fully data-driven boxes.

usage: python tools/bench_extract.py [--frames 1024] [--batch 64] [--hold 12]
Prints one JSON line per mode: fps sampler looking at every frame, batched and frame by frame (the reference's order of
work), and the accurate mode (detector loop over every frame + OCR of the selected ones)."""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vse_amd import engine, extractor, modelzoo, pipeline, shim, staging, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hold", type=int, default=12, help="frames one subtitle stays on screen")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--single", type=int, default=96, help="frames of the frame-by-frame run (slow)")
    ap.add_argument("--workers", type=int, default=4, help="copy threads of the pinned-slab uploader")
    ap.add_argument("--staged-only", action="store_true")
    a = ap.parse_args()
    ctx = engine.Context(0)
    det = modelzoo.get_model("V3_ch_det_fast", seed=0)
    rec = modelzoo.get_model("V4_en_rec_fast", seed=1)
    pipe = pipeline.OcrPipeline(ctx, det, rec, shim.en_charset(), rec_mode="bucketed")
    n_sub = (a.frames + 2 * a.hold - 1) // (2 * a.hold)
    lit = synth.make_frames(min(n_sub, 48), a.height, a.width, seed=9)
    dark = np.full((a.height, a.width, 3), 40, np.uint8)
    clip = []
    for k in range(n_sub):                                  # subtitle k for `hold` frames, then `hold` dark frames
        clip += [lit[k % len(lit)]] * a.hold + [dark] * a.hold
    clip = clip[:a.frames]
    fps = 24.0

    class Ocr:
        def predict(self, frame):
            b, r = pipe.ocr(torch.from_numpy(np.ascontiguousarray(frame)).to(ctx.tdev)[None])[0]
            return shim.OcrRecogniser.arrange(b, r)

    class OcrBatched(Ocr):
        def predict_batch(self, frames):
            return [shim.OcrRecogniser.arrange(b, r) for b, r in pipe.ocr(frames)]

    def detect_stream(batches):
        for dets in pipe.detect_stream(batches):
            yield [np.asarray(b, np.float32).reshape(-1, 4, 2) for b in dets]

    class OcrStreamed(OcrBatched):
        def predict_with_dets(self, frames, dets):
            return [shim.OcrRecogniser.arrange(b, r) for b, r in pipe.ocr_from_det(frames, dets)]

        def predict_stream(self, batches):
            for out in pipe.ocr_stream(batches):
                yield [shim.OcrRecogniser.arrange(b, r) for b, r in out]

    def detect(frames):
        dev = frames if torch.is_tensor(frames) else torch.from_numpy(np.stack(frames)).to(ctx.tdev)
        return [np.asarray(b, np.float32).reshape(-1, 4, 2) for b in pipe.detect(dev)]

    area = extractor.SubtitleArea(ymin=int(0.75 * a.height), ymax=a.height, xmin=0, xmax=a.width)
    up = staging.Uploader(ctx.tdev, workers=a.workers)
    runs = [
        ("fps sampler, every frame, staged upload + streamed detector", clip, OcrStreamed(),
         dict(sub_area=None, mode="fast", extract_frequency=fps, uploader=up)),
        ("fps sampler, every frame, batched, staged upload", clip, OcrBatched(),
         dict(sub_area=None, mode="fast", extract_frequency=fps, uploader=up)),
        ("accurate mode, staged upload + streamed detector, boxes reused", clip, OcrStreamed(),
         dict(sub_area=area, mode="accurate", uploader=up, detect_stream=detect_stream)),
        ("accurate mode, batched, staged upload", clip, OcrBatched(), dict(sub_area=area, mode="accurate", uploader=up)),
        ("fps sampler, every frame, batched", clip, OcrBatched(), dict(sub_area=None, mode="fast", extract_frequency=fps)),
        ("fps sampler, every frame, frame by frame", clip[:a.single], Ocr(), dict(sub_area=None, mode="fast", extract_frequency=fps)),
        ("accurate mode, batched", clip, OcrBatched(), dict(sub_area=area, mode="accurate")),
    ]
    if a.staged_only:
        runs = runs[:4]
    for name, frames, ocr, kw in runs:
        src = extractor.ArraySource(frames, fps)
        ex = extractor.SubtitleExtractor(src, ocr, detect_batch=detect, drop_score=0.0, batch=a.batch, **kw)
        ex_warm = extractor.SubtitleExtractor(extractor.ArraySource(frames[:5 * a.batch], fps), ocr, detect_batch=detect,
                                              drop_score=0.0, batch=a.batch, **kw)
        ex_warm.run()
        torch.cuda.synchronize()
        t0 = time.time()
        text = ex.run()
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(json.dumps({"mode": name, "frames": len(frames), "seconds": round(dt, 3), "frames_per_s": round(len(frames) / dt, 1),
                          "raw_lines": len(ex.raw_lines), "srt_blocks": text.count(" --> "),
                          "frame_mb": round(a.height * a.width * 3 / 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()
