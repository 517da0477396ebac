#!/bin/bash
# recogniser invocations as HIP graphs (bench.py --rec-graphs) vs plain launches, alternating on one box; + result equality
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from oracle import net_ref, pipeline_ref as P
from vse_amd import engine, pipeline, synth
ctx = engine.Context(0)
det = net_ref.get_weights("V3_ch_det_fast"); rec = net_ref.get_weights("V4_en_rec_fast")
pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), bucket=256, batch_round=4)
pipe.rec_streams = 2
frames = torch.from_numpy(synth.make_frames(24, 720, 1280, seed=9, p_two_lines=0.6)).cuda()
want = pipe.ocr(frames)
pipe.rec_graphs = True
for _ in range(3):
    got = pipe.ocr(frames)
    assert [r for _, r in got] == [r for _, r in want]
print("graph results identical over 3 runs:", sum(len(r) for _, r in want), "lines")
PY
for i in 1 2; do
run
run --rec-graphs
run --models fast-real
run --models fast-real --rec-graphs
done
