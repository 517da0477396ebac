#!/usr/bin/env python3
"""Debug: the crops of tests/test_gpu_pipeline.py::test_text_recognizer_call_site through the oracle and through the engine net (plain plan at
the chunk width, probabilities out), per sample: |delta log p| and where it sits along the sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ir_emul, net_ref, pipeline_ref as P
from vse_amd import engine, synth

frames, truth = synth.make_frames(3, 720, 1280, seed=21, p_two_lines=1.0, return_truth=True)
crops = []
for f, tr in enumerate(truth):
    for (x0, y0, x1, y1, _t) in tr:
        crops.append(np.ascontiguousarray(frames[f][y0 - 3:y1 + 3, x0 - 3:x1 + 3]))
crops.append(np.ascontiguousarray(crops[0][:, :40]))
rec = net_ref.get_weights(sys.argv[1] if len(sys.argv) > 1 else "V4_en_rec_fast")
ctx = engine.Context(0)
net = engine.Net(ctx, rec[0], rec[1], want_probs=True)
for idx, img_w in P.rec_batches(crops, 3):
    batch = np.stack([P.resize_norm_img(crops[i], img_w) for i in idx]).astype(np.float32)
    ref = net_ref.run_graph(rec[0], rec[1], batch)[0].numpy().astype(np.float64)
    x = torch.from_numpy(ir_emul.to_nhwc8(batch).astype(np.float16)).cuda()
    got = net.run(x)[0].cpu().numpy()[:, 0].astype(np.float64)
    for k, i in enumerate(idx):
        dl = np.abs(np.log(np.maximum(got[k], 1e-300)) - np.log(ref[k]))
        per_t = dl.max(-1)
        print(f"crop {i} shape {crops[i].shape} chunk width {img_w}: T {ref.shape[1]} max|dlogp| {dl.max():.3e} at t={int(per_t.argmax())}; per-step max:",
              " ".join(f"{v:.1e}" for v in per_t[::4]), " ref maxp", " ".join(f"{v:.3f}" for v in ref[k].max(-1)[::4]),
              " got maxp", " ".join(f"{v:.3f}" for v in got[k].max(-1)[::4]), flush=True)

# ---- the call-site path itself: shim.TextRecognizer in both modes vs the oracle's strings
from types import SimpleNamespace
from vse_amd import shim
shim.config.allow_standin_weights = True
charset = P.en_charset()
want = {}
for idx, img_w in P.rec_batches(crops, 3):
    batch = np.stack([P.resize_norm_img(crops[i], img_w) for i in idx])
    probs = net_ref.run_graph(rec[0], rec[1], batch)[0].numpy()
    for k, i in enumerate(idx):
        ids, conf = P.ctc_greedy(probs[k])
        want[i] = (P.decode_text(ids, charset), conf, img_w)
for mode in ("ragged", "reference"):
    tr = shim.TextRecognizer(SimpleNamespace(rec_model_dir="V4_en_rec_fast", rec_image_shape="3,48,320", lang="en", rec_batch_num=3, rec_mode=mode))
    got, _ = tr(crops)
    for i, (text, score) in enumerate(got):
        print(mode, i, crops[i].shape, "chunk", want[i][2], "engine", repr(text), round(score, 4), "oracle", repr(want[i][0]), round(want[i][1], 4), flush=True)
    print("groups:", [(list(g[0]), g[1], list(g[2]) if g[2] is not None else None) for g in tr.pipe._groups(
        [dict(frame=i, gframe=0, ratio=c.shape[1] / float(c.shape[0])) for i, c in enumerate(crops)])])
# ragged net with probabilities, the three narrow crops in one tensor
netr = engine.Net(ctx, rec[0], rec[1], want_probs=True, ragged=True)
for idx, img_w in P.rec_batches(crops, 3):
    wt = (img_w + 63) // 64 * 64
    batch = np.zeros((len(idx), 3, 48, wt), np.float32)
    for k, i in enumerate(idx):
        batch[k, :, :, :img_w] = P.resize_norm_img(crops[i], img_w)
    ref = net_ref.run_graph(rec[0], rec[1], batch[..., :img_w])[0].numpy().astype(np.float64)
    outs = netr.run(torch.from_numpy(ir_emul.to_nhwc8(batch).astype(np.float16)).cuda(), widths=np.full(len(idx), img_w, np.int32))
    got = outs[0].cpu().numpy()[:, 0].astype(np.float64)
    tl = netr.last_tlen.cpu().numpy()
    for k, i in enumerate(idx):
        tn = int(tl[k])
        dl = np.abs(np.log(np.maximum(got[k, :tn], 1e-300)) - np.log(ref[k]))
        print(f"ragged net crop {i} width {img_w} in a {wt}-px tensor: T {tn} vs oracle {ref.shape[1]}, max |dlogp| {dl.max():.3e}", flush=True)
