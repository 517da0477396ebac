"""Frames -> raw.txt -> SRT driver (vse_amd.extractor) against tests/golden/extract.json: the reference's own fps sampler
(backend/main.py:228-253) and OCR task producer / consumer (backend/tools/subtitle_ocr.py) executed on scripted inputs
(tests/golden/make_extract_golden.py).  Host logic only: the recogniser is the same scripted fake."""
import json
import os

import numpy as np
import pytest

from vse_amd import extractor

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "extract.json"), encoding="utf-8"))
H, W, _ = G["frame_shape"]


def frame_of(no):
    f = np.zeros((H, W, 3), np.uint8)
    f[:, :, 0] = no & 255
    f[:, :, 1] = no >> 8
    return f


class ScriptedOcr:
    def __init__(self, script, batched):
        self.script, self.seen = script, []
        if batched:
            self.predict_batch = self._predict_batch

    def predict(self, img):
        no = int(img[0, 0, 0]) | (int(img[0, 0, 1]) << 8)
        self.seen.append([no, list(img.shape)])
        o = self.script.get(str(no), [])
        return [q for q, _t, _s in o], [(t, s) for _q, t, s in o]

    def _predict_batch(self, frames):
        return [self.predict(np.asarray(f)) for f in frames]


@pytest.mark.parametrize("fc", G["fps_cases"], ids=lambda c: f"{c['n_frames']}f@{c['fps']}/{c['extract_frequency']}")
def test_fps_sampler_matches_reference(fc):
    got = extractor.fps_tasks(fc["n_frames"], fc["fps"], fc["extract_frequency"], "AREA")
    assert [[t[0], t[1], t[5]] for t in got] == fc["tasks"]
    assert all(t[2] is None and t[3] is None and t[4] is None for t in got)


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("k", range(len(G["ocr_cases"])))
def test_ocr_tasks_match_reference_raw_txt(k, batched, monkeypatch):
    c = G["ocr_cases"][k]
    monkeypatch.setattr(extractor, "_stack", lambda frames: frames)           # scripted recogniser: no device
    src = extractor.ArraySource([frame_of(i + 1) for i in range(c["n_frames"])], 25.0)
    ocr = ScriptedOcr(c["ocr"], batched)
    tasks = []
    for no, cached, default_area in c["tasks"]:
        o = c["ocr"].get(str(no), [])
        dt, rr = ([q for q, _t, _s in o], [(t, s) for _q, t, s in o]) if cached else (None, None)
        tasks.append((c["n_frames"], no, dt, rr, None, default_area))
    tasks.append((c["n_frames"], -1, None, None, None, None))
    area = None if c["area"] is None else extractor.SubtitleArea(**c["area"])
    lines = extractor.run_ocr_tasks(src, tasks, ocr, area, c["lang"], c["drop_score"], c["deviation"], batch=3)
    assert "".join(lines) == c["raw"]
    assert sorted(ocr.seen) == sorted(c["seen"])              # same frames recognised, same (cropped) shapes


def test_extractor_run_end_to_end_host():
    """fps sampler -> OCR -> scene-text filter -> SRT, and the same with an area (no filters); SRT numbering / time codes from
    srt.generate_subtitle_file (pinned separately by golden/srt.json)."""
    script = {}
    for no in range(1, 101):
        items = [[[[10, 30], [50, 30], [50, 36], [10, 36]], "first" if no <= 40 else "second subtitle", 0.9]]
        if no % 2:
            items.append([[[5, 2], [30, 2], [30, 8], [5, 8]], "LOGO", 0.99])
        script[str(no)] = items
    src = extractor.ArraySource([frame_of(i + 1) for i in range(100)], 25.0)
    ex = extractor.SubtitleExtractor(src, ScriptedOcr(script, False), extract_frequency=5)
    text = ex.run()
    assert text == "1\n00:00:00,001 --> 00:00:01,011\nfirst\n\n2\n00:00:01,016 --> 00:00:03,021\nsecond subtitle\n\n"
    assert extractor.SubtitleExtractor.srt2txt(text) == "first\nsecond subtitle\n"
    assert all("LOGO" not in ln for ln in ex.raw_lines)        # scene-text filter: most frequent vertical band only
    area = extractor.SubtitleArea(ymin=28, ymax=38, xmin=0, xmax=64)
    ex2 = extractor.SubtitleExtractor(src, ScriptedOcr(script, False), sub_area=area, extract_frequency=5)
    assert ex2.run() == text


def test_uncompressed_avi_roundtrip_and_refusals(tmp_path):
    from vse_amd import ingest
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (7, 18, 37, 3), dtype=np.uint8)          # odd width: rows padded to 4 bytes
    p = str(tmp_path / "clip.avi")
    ingest.write_avi_bgr24(p, frames, 23.976)
    src = ingest.open_source(p)
    assert (src.frame_count, src.width, src.height) == (7, 37, 18) and abs(src.fps - 23.976) < 1e-9
    assert all(np.array_equal(a, b) for a, b in zip(src.frames(), frames))
    assert np.array_equal(src.read(5), frames[4]) and src.read(0) is None and src.read(8) is None
    assert src.pos_msec(7) is None and src.pos_msec(0) == 0.0 and abs(src.pos_msec(3) - 3000 / 23.976) < 1e-9
    src.close()
    np.save(str(tmp_path / "clip.npy"), frames)
    nsrc = ingest.open_source(str(tmp_path / "clip.npy"), fps=25)
    assert nsrc.frame_count == 7 and np.array_equal(nsrc.read(7), frames[6])
    with pytest.raises(ValueError, match="fps"):
        ingest.open_source(str(tmp_path / "clip.npy"))
    # a compressed stream is refused with a reason instead of decoded wrongly
    raw = bytearray(open(p, "rb").read())
    i = raw.index(b"vidsDIB ")
    raw[i + 4:i + 8] = b"H264"
    (tmp_path / "h264.avi").write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="needs a codec"):
        ingest.AviBgr24Source(str(tmp_path / "h264.avi"))
    (tmp_path / "junk.avi").write_bytes(b"not a video at all")
    with pytest.raises(ValueError, match="RIFF"):
        ingest.AviBgr24Source(str(tmp_path / "junk.avi"))


def test_avi_opendml_segments_and_dropped_frames(tmp_path):
    """ffmpeg's AVI muxer opens a new `RIFF....AVIX` segment about every GiB (~170 frames of 1080p bgr24): all segments are
    read, trailing bytes that are not a segment are an error (never a silently shorter clip), and a zero-length frame chunk
    (dropped frame) shows the previous picture."""
    from vse_amd import ingest
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (11, 10, 21, 3), dtype=np.uint8)
    p = str(tmp_path / "seg.avi")
    ingest.write_avi_bgr24(p, frames, 25.0, riff_frames=4, dropped=(0, 5, 6))
    raw = open(p, "rb").read()
    assert raw.count(b"AVIX") == 2
    src = ingest.open_source(p)
    assert src.frame_count == 11
    got = list(src.frames())
    assert got[0] is None                                         # a dropped FIRST frame has no predecessor
    for k in range(1, 11):
        want = frames[4] if k in (5, 6) else frames[k]
        assert np.array_equal(got[k], want), k
    assert np.array_equal(src.read(7), frames[4]) and np.array_equal(src.read(11), frames[10])
    src.close()
    (tmp_path / "tail.avi").write_bytes(raw + b"JUNKJUNKJUNKJUNK")
    with pytest.raises(ValueError, match="not an AVI segment"):
        ingest.AviBgr24Source(str(tmp_path / "tail.avi"))


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = G["ocr_cases"][0]
    extractor._stack = lambda frames: frames
    src = extractor.ArraySource([frame_of(i + 1) for i in range(c["n_frames"])], 25.0)
    ocr = ScriptedOcr(c["ocr"], True)
    tasks = [(c["n_frames"], no, None, None, None, None) for no, _c, _a in c["tasks"]]
    lines = extractor.run_ocr_tasks(src, tasks, ocr, extractor.SubtitleArea(**c["area"]), c["lang"], c["drop_score"],
                                    c["deviation"], batch=2, shard=(rank, world), gather_device="cpu")
    q.put((rank, "".join(lines), sorted(n for n, _s in ocr.seen)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ocr_tasks_world2_gloo():
    """N > 1: each rank recognises a contiguous slice of the tasks, one variable-length gather, every rank ends up with the
    reference's raw.txt."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = G["ocr_cases"][0]
    assert got[0][1] == got[1][1] == c["raw"]
    assert got[0][2] == [1, 2, 3, 4, 5] and got[1][2] == [6, 7, 8, 9, 10]          # disjoint halves of the work


def test_prefetch_keeps_order_and_propagates_errors():
    """staging.prefetch with a stand-in uploader: batches arrive in order, staged by the producer thread; a failing source
    raises in the consumer; an abandoned iteration does not leave the producer blocked."""
    import threading
    from vse_amd import staging

    class FakeUploader:
        def __init__(self):
            self.threads = set()

        def bind_thread(self):
            self.threads.add(threading.get_ident())

        def stage(self, frames):
            return ("staged", [int(f.sum()) for f in frames])

    def batches(n, fail_at=None):
        for b in range(n):
            if b == fail_at:
                raise IOError("decoder died")
            yield [(10 * b + i, np.full((2, 2, 3), b + i, np.uint8)) for i in range(3)]

    up = FakeUploader()
    got = list(staging.prefetch(batches(5), up))
    assert [[k for k, _ in items] for items, _ in got] == [[10 * b + i for i in range(3)] for b in range(5)]
    assert got[3][1] == ("staged", [12 * (3 + i) for i in range(3)])
    assert up.threads and threading.get_ident() not in up.threads
    with pytest.raises(IOError):
        list(staging.prefetch(batches(5, fail_at=2), up))
    it = staging.prefetch(batches(50), up, ahead=1)
    next(it)
    it.close()                                  # consumer gives up: the producer must terminate
    assert threading.active_count() < 8


def test_run_ocr_tasks_with_uploader_equals_host_stacking():
    """The staged route (producer thread + uploader) gives the same lines as the host-stacking route."""
    from vse_amd import extractor

    class Staged:
        def __init__(self, frames):
            self.frames = frames

        def tensor(self):
            return np.stack(self.frames)

    class FakeUploader:
        def bind_thread(self):
            pass

        def stage(self, frames):
            return Staged([np.array(f) for f in frames])

    class Ocr:
        def predict(self, frame):
            v = int(frame[0, 0, 0])
            return [[(v, v), (v + 30, v), (v + 30, v + 10), (v, v + 10)]], [(f"t{v}", 0.9)]

    class OcrBatched(Ocr):
        def predict_batch(self, frames):
            assert isinstance(frames, np.ndarray) and frames.ndim == 4        # what the uploader staged, not a list
            return [self.predict(f) for f in frames]

    frames = [np.full((8 + 2 * (i // 5), 40, 3), i, np.uint8) for i in range(13)]
    src = extractor.ArraySource(frames, 10.0)
    tasks = extractor.fps_tasks(13, 10.0, 10)
    a = extractor.run_ocr_tasks(src, tasks, Ocr(), batch=4)
    b = extractor.run_ocr_tasks(src, tasks, OcrBatched(), batch=4, uploader=FakeUploader())
    assert a == b and len(a) == 13


def test_run_ocr_tasks_uses_predict_stream_in_batch_order():
    """A recogniser with predict_stream (detector of the next batches in flight) is fed the staged batches as a generator and
    its results are matched back in batch order; same lines as frame by frame."""
    from vse_amd import extractor

    class Staged:
        def __init__(self, frames):
            self.frames = frames

        def tensor(self):
            return np.stack(self.frames)

    class FakeUploader:
        def bind_thread(self):
            pass

        def stage(self, frames):
            return Staged([np.array(f) for f in frames])

    class Ocr:
        def predict(self, frame):
            v = int(frame[0, 0, 0])
            return [[(v, v), (v + 30, v), (v + 30, v + 10), (v, v + 10)]], [(f"t{v}", 0.9)]

    class OcrStreamed(Ocr):
        seen = []

        def predict_batch(self, frames):
            raise AssertionError("the streamed form must be preferred")

        def predict_stream(self, batches):
            held = []
            for b in batches:                          # one batch ahead, like a detector in flight
                held.append(b)
                if len(held) > 1:
                    f = held.pop(0)
                    self.seen.append(len(f))
                    yield [self.predict(x) for x in f]
            for f in held:
                self.seen.append(len(f))
                yield [self.predict(x) for x in f]

    frames = [np.full((8 + 2 * (i // 5), 40, 3), i, np.uint8) for i in range(13)]
    src = extractor.ArraySource(frames, 10.0)
    tasks = extractor.fps_tasks(13, 10.0, 10)
    a = extractor.run_ocr_tasks(src, tasks, Ocr(), batch=4)
    ocr = OcrStreamed()
    b = extractor.run_ocr_tasks(src, tasks, ocr, batch=4, uploader=FakeUploader())
    assert a == b and ocr.seen == [4, 1, 4, 1, 3]          # shape changes close a batch: 5 + 5 + 3 frames of three shapes


def test_mjpeg_avi_is_decoded_by_pillow(tmp_path):
    """Motion-JPEG AVI — the one compressed format with a decoder on these boxes (Pillow / libjpeg): every frame of the file equals
    Pillow's own decode of that frame's JPEG bytes (BGR order like cv2), random access works, and the stream still refuses other
    codecs.  (FFmpeg's MJPEG decoder, which the reference reaches through cv2.VideoCapture, may differ by a grey level or two.)"""
    import io
    from PIL import Image
    from vse_amd import ingest, synth
    frames = synth.make_frames(5, 120, 200, seed=3)
    p = str(tmp_path / "clip_mjpeg.avi")
    ingest.write_avi_mjpeg(p, frames, 25.0, quality=92)
    src = ingest.open_source(p)
    assert src.mjpeg and (src.frame_count, src.width, src.height) == (5, 200, 120) and abs(src.fps - 25.0) < 1e-9
    got = list(src.frames())
    for k, g in enumerate(got):
        assert g.shape == (120, 200, 3) and g.dtype == np.uint8
        assert np.abs(g.astype(int) - frames[k].astype(int)).mean() < 8.0            # lossy (noisy synthetic background), but the same picture
    # equals Pillow's decode of the very bytes stored in the file
    raw = open(p, "rb").read()
    off = raw.index(b"00dc")
    n = int.from_bytes(raw[off + 4:off + 8], "little")
    ref = np.asarray(Image.open(io.BytesIO(raw[off + 8:off + 8 + n])).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(got[0], ref) and np.array_equal(src.read(1), ref)
    assert src.read(6) is None and src.pos_msec(2) == 80.0
    src.close()
