import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--steps", "1", "--warmup", "1"]
# re-create the pieces of bench.main
from vse_amd import engine, modelzoo, pipeline, shim, synth
ctx = engine.Context(0)
det = modelzoo.get_model("V4_ch_det", seed=0); rec = modelzoo.get_model("V4_ch_rec", seed=1)
det = (det[0], bench.empty_det_head(det[0], det[1]))
charset = shim.standin_charset("ch", shim._ncls(rec[0]))
pipe = pipeline.OcrPipeline(ctx, det, rec, charset, rec_mode="bucketed", bucket=256, batch_round=4)
pipe.rec_streams = 2
frames_np, truth = synth.make_frames(64, 1080, 1920, seed=100, return_truth=True)
frames = torch.from_numpy(frames_np).cuda()
quads = bench.gt_quads(truth)
def step(timing=None):
    t0 = time.perf_counter()
    maps = pipe.det_maps(frames); t1 = time.perf_counter()
    db = ctx.db_postprocess(maps, 1080, 1920, **pipe.db); t2 = time.perf_counter()
    res = pipe.recognize(frames, quads); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if timing is not None: timing.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(3): step()
T = []
for _ in range(5): step(T)
a = np.array(T).mean(0) * 1e3
print("host ms: det launch %.2f | db_postprocess (incl. wait for det) %.2f | recognize (launch+final D2H syncs) %.2f | tail sync %.2f" % tuple(a))
# recognize prep only (no GPU): specs + groups
t0 = time.perf_counter()
for _ in range(20):
    specs = pipe._crop_specs(quads); groups = pipe._groups(specs)
print("recognize host prep (specs+groups): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
