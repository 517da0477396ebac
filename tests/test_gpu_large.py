"""Full-size detector inputs on the HIP engine vs the CPU oracle:
  * V4_ch_det (the server detector of BASELINE configs[1]) at the reference's own input size 1 x 544 x 960,
  * BASELINE configs[2]'s stress form: a 4K frame with det_limit_side_len = 3840 -> 2176 x 3840 detector input (16x the
    pixels of the reference's default), the real-weight mobile detector end to end (map + identical boxes) and the server
    detector's map.
The oracle needs seconds (544 x 960) to about a minute (2176 x 3840, server model) of host time."""
import numpy as np
import pytest

from oracle import net_ref, pipeline_ref as P

pytestmark = pytest.mark.gpu


def test_server_detector_at_the_reference_input_size(ctx):
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V4_ch_det")
    rec = net_ref.get_weights("V4_en_rec_fast")
    frames = synth.make_frames(1, 1080, 1920, seed=9)
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset())
    got = pipe.det_maps(torch.from_numpy(frames).cuda()).cpu().numpy()[0]
    x, _ = P.det_preprocess(frames[0])
    assert x.shape == (1, 3, 544, 960)
    ref = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
    d = np.abs(got - ref)
    assert ref.max() - ref.min() > 0.2                      # the stand-in head is not saturated
    # (522 k pixels of a stand-in head that sits near 0.5: single-pixel maxima of 0.018-0.025 measured, fp16 activation storage)
    assert d.max() < 4e-2 and d.mean() < 1e-3, (d.max(), d.mean())
    # (a live stand-in map crosses the 0.3 threshold all over the frame: the pixels that land on the other side are the ones whose
    # oracle value lies within the map error of it — measured 1.0e-3 of them)
    assert ((got > 0.3) != (ref > 0.3)).mean() < 3e-3
    assert not (((got > 0.3) != (ref > 0.3)) & (np.abs(ref - 0.3) > d.max())).any()


def test_4k_frame_with_limit_side_3840(ctx):
    import torch
    from vse_amd import pipeline, synth
    frames = synth.make_frames(1, 2160, 3840, seed=3)
    dev = torch.from_numpy(frames).cuda()
    rec = net_ref.get_weights("V4_en_rec_fast")
    x, _ = P.det_preprocess(frames[0], 3840)
    assert x.shape == (1, 3, 2176, 3840)
    # real-weight mobile detector: map and boxes
    det = net_ref.get_weights("V3_ch_det_fast")
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), limit_side_len=3840)
    maps = pipe.det_maps(dev)
    got = maps.cpu().numpy()[0]
    ref = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
    assert np.isfinite(got).all() and np.abs(got - ref).max() < 1e-1
    assert ((got > 0.3) != (ref > 0.3)).mean() < 1e-4
    gb = pipeline.sorted_boxes(ctx.db_postprocess(maps, 2160, 3840)[0][0])
    wb = P.sorted_boxes(P.db_postprocess(ref, 2160, 3840)[0])
    assert len(gb) == len(wb) > 0
    same = sum(np.array_equal(a, b) for a, b in zip(gb, wb))
    assert same == len(wb), (same, len(wb))                 # every box the oracle's integers (chained detector with pair tensors)
    del pipe, maps
    # server detector (stand-in weights): the 16x map against the oracle
    det = net_ref.get_weights("V4_ch_det")
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), limit_side_len=3840)
    got = pipe.det_maps(dev).cpu().numpy()[0]
    ref = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
    d = np.abs(got - ref)
    # (8.4 M pixels of a stand-in head that sits near 0.5 everywhere: the largest single deviation measured is 0.031)
    assert np.isfinite(got).all() and d.max() < 6e-2 and d.mean() < 2e-3, (d.max(), d.mean())      # (live stand-in, round 5: mean 1.0e-3)
    # (a live stand-in map crosses the 0.3 threshold all over the frame: the pixels that land on the other side are the ones whose
    # oracle value lies within the map error of it — measured 1.0e-3 of them)
    assert ((got > 0.3) != (ref > 0.3)).mean() < 3e-3
    assert not (((got > 0.3) != (ref > 0.3)) & (np.abs(ref - 0.3) > d.max())).any()
