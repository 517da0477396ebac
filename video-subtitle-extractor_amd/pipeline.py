"""Host orchestration of the per-frame OCR hot path on top of the C-ABI engine (Python, as the reference's is).

frames (uint8 BGR, device-resident) -> det pre-process -> DB det net -> DB post-process -> box ordering ->
perspective crops + rec pre-process -> CTC rec net -> arg-max / CTC collapse -> (boxes, (text, score)).

Replaces paddleocr's TextDetector / TextRecognizer / TextSystem as called from backend/tools/subtitle_detect.py:25
and backend/tools/ocr.py:27.  Unlike the reference (one frame per call) every stage takes a BATCH of
frames: frames are independent, so det runs N frames per launch and rec runs the crops of all N frames.

Recognition batching.  The reference recognises the crops of ONE frame in chunks of rec_batch_num (6), each chunk
zero-padded to its own widest crop, and what a crop's logits are depends on that padded width (conv borders, SVTR
attention span).  Three modes:
  * "reference": exactly that launch structure — one network run per chunk.
  * "ragged" (default): every crop keeps the padded width of ITS reference chunk, but crops of all frames share
    launches: the recogniser plans are compiled for ragged batches (compiler.compile_model(ragged=True)), every
    kernel treats x >= a sample's width as outside the image, and a sample's outputs are bit-identical to the
    "reference" mode's whatever batch it rides in.  Groups are formed for throughput only (width buckets).
  * "bucketed": crops padded to the width of their bucket — faster grouping of round 1-2, NOT the reference's
    padding (logits near the right edge differ); kept for A/B.
"""
import math

import numpy as np

from . import engine


DET_NORM = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))      # paddleocr NormalizeImage of the DB detectors (mean, std; scale 1/255)


def det_resize_shape(h, w, limit_side_len=960):
    """paddleocr DetResizeForTest type0, limit_type='max' (SURVEY App. C.1)."""
    ratio = float(limit_side_len) / max(h, w) if max(h, w) > limit_side_len else 1.0
    rh, rw = int(h * ratio), int(w * ratio)
    return max(int(round(rh / 32) * 32), 32), max(int(round(rw / 32) * 32), 32)


def sorted_boxes(boxes):
    """paddleocr predict_system.sorted_boxes: top-to-bottom, then left-to-right within 10 px rows."""
    bs = sorted(list(boxes), key=lambda b: (b[0][1], b[0][0]))
    for i in range(len(bs) - 1):
        for j in range(i, -1, -1):
            if abs(bs[j + 1][0][1] - bs[j][0][1]) < 10 and bs[j + 1][0][0] < bs[j][0][0]:
                bs[j], bs[j + 1] = bs[j + 1], bs[j]
            else:
                break
    return bs


def crop_geometry(q):
    q = np.asarray(q, dtype=np.float32)
    cw = int(max(np.linalg.norm(q[0] - q[1]), np.linalg.norm(q[2] - q[3])))
    ch = int(max(np.linalg.norm(q[0] - q[3]), np.linalg.norm(q[1] - q[2])))
    rotate = 1 if (cw > 0 and ch * 1.0 / cw >= 1.5) else 0
    return cw, ch, rotate


def resolve_det_weights(det_weights, weights):
    """"auto" -> "fp16x2" (fp16 hi + lo weight pairs) for detectors under 4 M parameters, else "fp16"."""
    if det_weights == "auto":
        nparam = sum(int(np.prod(v.shape)) for v in weights.values())
        det_weights = "fp16x2" if nparam < 4_000_000 else "fp16"
    assert det_weights in ("fp16", "fp16x2"), det_weights
    return det_weights


class OcrPipeline:
    def __init__(self, ctx, det_model, rec_model, charset, rec_batch_num=6, rec_h=48, rec_base_w=320,
                 limit_side_len=960, db_thresh=0.3, db_box_thresh=0.6, db_unclip_ratio=1.5, drop_score=0.0,
                 rec_mode="ragged", bucket=64, batch_round=1, max_rec_batch=64, det_weights="auto", min_rec_group=0,
                 det_input="raw", det_chains=None):
        """det_model / rec_model: (descriptor, weights dict).
        det_weights: "fp16" | "fp16x2" | "auto".  fp16x2 stores the detector's conv weights as fp16 hi + lo pairs (two K
        passes into the same fp32 accumulators): the rounding of BN-folded weights to fp16 is what moves box borders against
        an fp32 reference.  "auto" uses it for the mobile detectors (< 4 M parameters), where the second pass
        hides behind the memory traffic, and plain fp16 for the server models.
        det_chains: None (default) — detectors that run with fp16x2 weights (the mobile models) also run their 1x1 / depthwise runs as
        LDS-resident chains with fp16 hi + lo pair tensors between ops (compiler chains.py, csrc/chain.hip): a tensor is rounded to 11
        bits a handful of times per network instead of once per layer — the box parity of DESIGN 4 — at 1.2-1.3x the detector time;
        False keeps the layer-by-layer program (faster, 2-3 boxes per 400 a pixel row off the fp32 reference).
        det_input: "raw" — the detector is fed the resized uint8 pixels themselves (+ a ones channel) and its stem conv carries
        paddleocr's (x/255 - mean)/std, so it computes on the reference's exact input values; the resize itself happens inside
        the stem kernel's patch staging (no pre-processing pass, no fp16 input tensor); "normalized" — the fp16 rounding
        of the normalised image (rounds 1-2; moves the real detector's map by up to 4e-3)."""
        self.ctx = ctx
        assert det_input in ("raw", "normalized")
        self.det_input = det_input
        self.det_weights = det_weights = resolve_det_weights(det_weights, det_model[1])
        self.det = engine.Net(ctx, det_model[0], det_model[1], fetch_cols=(0,), hilo=det_weights == "fp16x2",
                              input_norm=DET_NORM if det_input == "raw" else None, fuse_preprocess=det_input == "raw", chain=det_chains)
        # ragged plans for every mode: "reference" runs them with uniform widths, so the modes share kernels and summation orders
        self.rec = engine.Net(ctx, rec_model[0], rec_model[1], want_probs=False, ragged=True)
        self.charset = charset
        self.rec_batch_num = rec_batch_num
        self.rec_h = rec_h
        self.rec_base_w = rec_base_w
        self.limit = limit_side_len
        self.db = dict(thresh=db_thresh, box_thresh=db_box_thresh, unclip_ratio=db_unclip_ratio)
        self.drop_score = drop_score
        self.rec_mode = rec_mode
        self.bucket = bucket
        self.batch_round = batch_round        # bucketed mode: pad group sizes to a multiple (bounds the plan cache)
        self.max_rec_batch = max_rec_batch
        self.min_rec_group = min_rec_group    # bucketed mode: buckets with fewer crops absorb the next narrower bucket
        self.profile_sink = None              # list: when set, every net run is profiled per op and appended here
        self.rec_streams = 1                  # >1: width groups of the recogniser run on that many side streams
        # ragged grouping (_ragged_partition): fixed cost of one launch sequence in crop-pixels.  The mobile recognisers (< 8 M
        # parameters: ~80 launches of ~10 us per sequence, latency-bound) do better with MORE, smaller groups spread over the side
        # streams than the server model (MI355X, bench.py --models fast, 4 streams: 8.42-8.51 k frames/s at 5000, 8.58-8.67 k at 2000)
        nparam_rec = sum(int(np.prod(v.shape)) for v in rec_model[1].values())
        self.ragged_launch_cost = 5000 if nparam_rec >= 8_000_000 else 2000

    def _run(self, net, x, slot=0, widths=None):
        if getattr(self, "profile_sink", None) is not None:
            # the profiled run IS the run (a second, asynchronous run would overlap the next group's profile on another stream
            # and inflate its per-op times: rec groups read 1.5x too long in round 3's first bench lines)
            self.profile_sink.append(net.profile(x, slot, widths=widths))
            return net.last_outs
        return net.run(x, slot, widths=widths)

    # ---- detection ---------------------------------------------------------------------------------------
    def det_maps(self, frames, slot=0):
        """frames: cuda uint8 [N,H,W,3] -> cuda fp32 prob maps [N,h,w].  `slot` selects the detector workspace: two
        batches may be in flight on different streams when they use different slots."""
        n, h, w, _ = frames.shape
        rh, rw = det_resize_shape(h, w, self.limit)
        if getattr(self, "profile_sink", None) is None:
            return self.det.det_forward(frames, rh, rw, slot)          # vse_det_forward: pre-process + network in one call
        self.profile_sink.append(self.det.profile_frames(frames, rh, rw, slot))
        return self.det.last_outs[0].view(n, rh, rw)

    def detect(self, frames):
        """-> list per frame of float32 [k,4,2] boxes (paddleocr TextDetector output, unsorted)."""
        n, h, w, _ = frames.shape
        prob = self.det_maps(frames)
        res = self.ctx.db_postprocess(prob, h, w, **self.db)
        return [r[0] for r in res]

    # ---- recognition -------------------------------------------------------------------------------------
    def _crop_specs(self, boxes_per_frame):
        specs = []
        # crop sizes for all boxes at once when the corners are integer-valued (DB boxes always are): the squared side
        # lengths are then exact in float32, so the batched sqrt(x0^2 + x1^2) equals np.linalg.norm's sqrt(dot(x, x)) bit
        # for bit; otherwise the per-box form of the reference is kept.  (This runs while the GPU waits for the crops.)
        flat = [np.asarray(q, np.float32).reshape(4, 2) for boxes in boxes_per_frame for q in boxes]
        geo = None
        if flat:
            Q = np.stack(flat)
            if np.all(Q == np.rint(Q)) and np.abs(Q).max() < 4000:
                def side(a, b):
                    d = Q[:, a] - Q[:, b]
                    return np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
                cws = np.maximum(side(0, 1), side(2, 3)).astype(np.int64)
                chs = np.maximum(side(0, 3), side(1, 2)).astype(np.int64)
                geo = [(int(a), int(b), 1 if (a > 0 and b * 1.0 / a >= 1.5) else 0) for a, b in zip(cws, chs)]
        gi = 0
        for f, boxes in enumerate(boxes_per_frame):
            for k, q in enumerate(boxes):
                cw, ch, rot = geo[gi] if geo is not None else crop_geometry(q)
                gi += 1
                iw, ih = (ch, cw) if rot else (cw, ch)
                specs.append(dict(frame=f, slot=k, quad=np.asarray(q, np.float32), crop_w=max(cw, 1),
                                  crop_h=max(ch, 1), rotate=rot, ratio=(iw / float(ih)) if ih > 0 else 1.0,
                                  iw=max(iw, 1), ih=max(ih, 1)))
        return specs

    def _reference_chunks(self, specs):
        """The reference's grouping: per frame, crops sorted by w/h, chunks of rec_batch_num, every chunk as wide as its
        widest crop (at least rec_base_w).  -> list of (spec indices, img_w)."""
        chunks = []
        base = self.rec_base_w / float(self.rec_h)
        by_frame = {}
        for i, s in enumerate(specs):
            by_frame.setdefault(s.get("gframe", (s.get("src", 0), s["frame"])), []).append(i)
        for f in sorted(by_frame):
            idx = by_frame[f]
            order = [idx[j] for j in np.argsort(np.array([specs[i]["ratio"] for i in idx]), kind="stable")]
            for b in range(0, len(order), self.rec_batch_num):
                chunk = order[b:b + self.rec_batch_num]
                mx = max([base] + [specs[i]["ratio"] for i in chunk])
                chunks.append((chunk, int(self.rec_h * mx)))
        return chunks

    def _ragged_partition(self, need):
        """Ragged mode: results do not depend on the grouping, so groups are chosen for cost alone.  Crops sorted by width are
        cut into runs by dynamic programming over   cost(run) = max(n * W, floor) + launch,   n = crops of the run rounded up to
        batch_round, W = its widest crop rounded up to 64 px (the tensor every sample of the run is computed on: work is
        proportional to n * W), `floor` = the n * W below which a launch sequence no longer gets faster (a handful of crops fill
        a fraction of the chip), `launch` = the fixed cost of one ~80-kernel sequence in the same unit (MI355X: ~0.8 ms against
        ~7.5 k crop-pixels per ms).  Fixed 256-px buckets padded the headline workload to 70.7 k crop-pixels of tensor for 53 k of
        crops; this partition to 64.3 k (end to end +0.6 %: the recogniser's sequences hide beside the detector streams)."""
        order = sorted(range(len(need)), key=lambda i: (need[i], i))
        n = len(order)
        if n == 0:
            return []
        rnd = max(1, getattr(self, "batch_round", 1))
        floor_, launch = getattr(self, "ragged_floor", 6000), getattr(self, "ragged_launch_cost", 5000)
        cap = self.max_rec_batch
        best = [0.0] + [float("inf")] * n
        cut = [0] * (n + 1)
        for j in range(1, n + 1):
            wj = (need[order[j - 1]] + 63) // 64 * 64
            for i in range(max(0, j - cap), j):
                c = best[i] + max(-(-(j - i) // rnd) * rnd * wj, floor_) + launch
                if c < best[j]:
                    best[j], cut[j] = c, i
        groups, j = [], n
        while j > 0:
            i = cut[j]
            part = order[i:j]
            groups.append((part, (need[order[j - 1]] + 63) // 64 * 64, [need[k] for k in part]))
            j = i
        return groups[::-1]

    def _groups(self, specs):
        """-> list of (spec indices, tensor width, per-sample widths): one recogniser run each."""
        if self.rec_mode == "reference":
            return [(chunk, w, [w] * len(chunk)) for chunk, w in self._reference_chunks(specs)]
        if self.rec_mode == "ragged":
            own = {}
            for chunk, w in self._reference_chunks(specs):
                for i in chunk:
                    own[i] = w                      # the padded width crop i has in the reference
            need = [own[i] for i in range(len(specs))]
            return self._ragged_partition(need)
        need = [max(self.rec_base_w, int(math.ceil(self.rec_h * s["ratio"]))) for s in specs]
        buckets = {}
        for i, wn in enumerate(need):
            wb = (wn + self.bucket - 1) // self.bucket * self.bucket
            buckets.setdefault(wb, []).append(i)
        # a bucket with a handful of crops runs ~80 launches that fill a fraction of the chip (4 crops x 1280 px: 2.9 ms,
        # 24 x 1024: 4.2 ms on MI355X): fold the next narrower bucket into it (its crops are padded further) until the
        # group is worth its launches — from the widest bucket down, since only wider buckets can hold narrower crops
        min_group = getattr(self, "min_rec_group", 0)
        if min_group > 1:
            widths = sorted(buckets, reverse=True)
            k = 0
            while k < len(widths) - 1:
                if len(buckets[widths[k]]) < min_group:
                    buckets[widths[k]] += buckets.pop(widths[k + 1])
                    del widths[k + 1]
                else:
                    k += 1
        groups = []
        for wb in sorted(buckets):
            idx = buckets[wb]
            for b in range(0, len(idx), self.max_rec_batch):
                part = idx[b:b + self.max_rec_batch]
                groups.append((part, wb, [wb] * len(part)))
        return groups

    def recognize(self, frames, boxes_per_frame):
        """boxes_per_frame: list (len N) of sequences of 4x2 quads (already in the order results are wanted).
        -> list per frame of [(text, score)]."""
        specs = self._crop_specs(boxes_per_frame)
        results = [[("", 0.0)] * len(b) for b in boxes_per_frame]
        for s, r in zip(specs, self._recognize_specs(frames, specs)):
            results[s["frame"]][s["slot"]] = r
        return results

    def recognize_multi(self, frames_list, boxes_list):
        """recognize() over SEVERAL frame tensors at once: frames_list[k] = cuda uint8 [N_k,H,W,3], boxes_list[k] = its boxes per
        frame.  In the ragged mode results do not depend on how crops are grouped, so the crops of consecutive frame batches of a
        video share launch sequences: the small maps of the recogniser (12 / 6 / 3 rows) fill the chip only when a launch carries
        many crops (MI355X, V4_ch_rec, 78 crops per 64-frame batch: 9.3 ms of sequential GPU time per batch in their own sequences,
        7.7 ms per batch together with the next batch's; end to end +0.9 %).
        -> list (per tensor) of list (per frame) of [(text, score)]."""
        return self.recognize_multi_collect(self.recognize_multi_launch(frames_list, boxes_list))

    def recognize_multi_launch(self, frames_list, boxes_list):
        """The launch half of recognize_multi(): crops are cut and every recogniser sequence is enqueued on the side streams; NOTHING
        is read back — the host does not wait for the GPU.  -> a handle for recognize_multi_collect().  A streaming caller launches
        span s + 1 before it collects span s (ocr_stream, bench.py): the read-back then finds its results finished instead of
        stalling the host — and with it the detector batches it has yet to enqueue — behind the recogniser's launch chains."""
        specs = []
        for k, boxes_per_frame in enumerate(boxes_list):
            for s in self._crop_specs(boxes_per_frame):
                s["src"] = k
                specs.append(s)
        shape = [[len(b) for b in boxes_per_frame] for boxes_per_frame in boxes_list]
        return dict(specs=specs, shape=shape, run=self._launch_specs(list(frames_list), specs))

    def recognize_multi_collect(self, handle):
        results = [[[("", 0.0)] * nb for nb in per_frame] for per_frame in handle["shape"]]
        for s, r in zip(handle["specs"], self._collect_specs(handle["run"])):
            results[s["src"]][s["frame"]][s["slot"]] = r
        return results

    def recognize_crops(self, crops):
        """paddleocr TextRecognizer.__call__(img_list): crops = list of uint8 BGR [h,w,3] arrays of any sizes ->
        [(text, score)] in input order.  The whole list is ONE grouping unit (sorted by w/h, chunks of rec_batch_num, each
        chunk padded to its own widest member), as the reference's recogniser treats the crops of one frame.  The crops ride
        through the same device path as boxes cut from frames: they are placed on a common canvas and "cropped" with the
        identity quad (bicubic weights at integer coordinates are exactly (0,1,0,0): a pixel copy)."""
        t = self.ctx.torch
        if not crops:
            return []
        mh = max(int(c.shape[0]) for c in crops)
        mw = max(int(c.shape[1]) for c in crops)
        canvas = np.zeros((len(crops), mh, mw, 3), np.uint8)
        specs = []
        for i, c in enumerate(crops):
            c = np.asarray(c)
            if c.dtype != np.uint8 or c.ndim != 3 or c.shape[2] != 3 or c.shape[0] < 1 or c.shape[1] < 1:
                raise ValueError("expected uint8 BGR HxWx3 crops")
            h, w = int(c.shape[0]), int(c.shape[1])
            canvas[i, :h, :w] = c
            quad = np.array([[0, 0], [w, 0], [w, h], [0, h]], np.float32)
            specs.append(dict(frame=i, gframe=0, slot=0, quad=quad, crop_w=w, crop_h=h, rotate=0, ratio=w / float(h),
                              iw=w, ih=h))
        return self._recognize_specs(t.from_numpy(canvas).to(self.ctx.tdev), specs)

    def _recognize_specs(self, frames, specs):
        """-> [(text, score)] per spec.  frames: one cuda uint8 [N,H,W,3] tensor, or a list of them with spec["src"] naming the
        tensor a crop is cut from."""
        return self._collect_specs(self._launch_specs(frames, specs))

    def _launch_specs(self, frames, specs):
        """Crops + recogniser sequences of `specs` enqueued (side streams when rec_streams > 1); -> run handle for _collect_specs.
        The main stream does NOT wait for the side streams: every group records an event, the read-back waits for those."""
        t = self.ctx.torch
        frames_list = list(frames) if isinstance(frames, (list, tuple)) else [frames]
        if not specs:
            return dict(n=0, pending=[])
        pending = []
        groups = self._groups(specs)
        # width groups are independent: run them on side streams so the latency-bound launches of small groups overlap.
        # Every stream owns a workspace slot of the recogniser: two groups with the same (n, h, w) plan key may be in flight
        # at once (a bucket split into max_rec_batch chunks, reference-mode chunks of equal shape) and must not share one.
        nstreams = min(len(groups), getattr(self, "rec_streams", 1))
        main = t.cuda.current_stream(self.ctx.tdev)
        if nstreams > 1:
            if len(getattr(self, "_streams", [])) < nstreams:
                # streams VERIFIED to run beside each other and beside the main stream, shared by every pipeline of the context
                # (engine.Context.side_streams: torch's pool streams may alias one hardware queue and then execute in order)
                self._streams = self.ctx.side_streams(nstreams, priority=getattr(self, "rec_stream_priority", -1), role="rec")
            for st in self._streams[:nstreams]:
                st.wait_stream(main)
        try:
            for gi, (idx, img_w, widths) in enumerate(groups):
                if nstreams > 1:
                    t.cuda.set_stream(self._streams[gi % nstreams])
                if len(frames_list) > 1:                 # rows of one source tensor next to each other: one crop launch per tensor
                    order = sorted(range(len(idx)), key=lambda k: specs[idx[k]].get("src", 0))
                    idx, widths = [idx[k] for k in order], [widths[k] for k in order]
                crops, srcs = [], []
                for i, wi in zip(idx, widths):
                    s = specs[i]
                    rw = min(wi, int(math.ceil(self.rec_h * s["ratio"])))
                    crops.append(dict(quad=s["quad"], frame=s["frame"], crop_w=s["crop_w"], crop_h=s["crop_h"],
                                      resized_w=max(rw, 1), rotate=s["rotate"]))
                    srcs.append(s.get("src", 0))
                widths = list(widths)
                if self.rec_mode != "reference" and self.batch_round > 1:
                    while len(crops) % self.batch_round:
                        crops.append(crops[-1])           # dummy rows; their results are never read
                        widths.append(widths[-1])
                        srcs.append(srcs[-1])
                slot = gi % nstreams if nstreams > 1 else 0
                use_graph = (getattr(self, "rec_graphs", False) and nstreams > 1 and getattr(self, "profile_sink", None) is None)
                bufs = self.rec.rec_graph_buffers(len(crops), self.rec_h, img_w, slot) if use_graph else None
                if len(frames_list) == 1:
                    x = self.ctx.rec_preprocess(frames_list[0], crops, self.rec_h, img_w, out=bufs["x"] if bufs else None)
                else:
                    x = bufs["x"] if bufs else t.empty((len(crops), self.rec_h, img_w, 8), dtype=t.float16, device=self.ctx.tdev)
                    a = 0
                    while a < len(crops):
                        b = a
                        while b < len(crops) and srcs[b] == srcs[a]:
                            b += 1
                        self.ctx.rec_preprocess(frames_list[srcs[a]], crops[a:b], self.rec_h, img_w, out=x[a:b])
                        a = b
                if bufs is not None:
                    oi, ol, oc = self.rec.rec_forward_graph(bufs, np.asarray(widths, np.int32))     # the same as ONE HIP graph launch
                elif getattr(self, "profile_sink", None) is None:
                    oi, ol, oc = self.rec.rec_forward(x, np.asarray(widths, np.int32), slot)      # vse_rec_forward: network + CTC collapse
                else:
                    idx_maxp = self._run(self.rec, x, slot=slot, widths=np.asarray(widths, np.int32))[-1]          # [B,1,T,2]
                    oi, ol, oc = self.ctx.ctc_collapse(idx_maxp, self.rec.last_tlen)
                ev = t.cuda.Event()
                ev.record(t.cuda.current_stream(self.ctx.tdev))
                pending.append((idx, oi, ol, oc, ev))
        finally:
            if nstreams > 1:
                t.cuda.set_stream(main)
                # the frames were allocated on the caller's stream and are read by the crop kernels on the side streams: tell the
                # allocator (the main stream no longer waits for the side streams here, so nothing else orders a later free behind them)
                for st in self._streams[:nstreams]:
                    for fr in frames_list:
                        fr.record_stream(st)
        return dict(n=len(specs), pending=pending)

    def _collect_specs(self, run):
        """Read-back + string decode of a launched run: waits for each group's event (long finished when the caller launched the
        next span in between), one device -> host copy per output."""
        out = [("", 0.0)] * run["n"]
        for idx, oi, ol, oc, ev in run["pending"]:
            ev.synchronize()
            oi, ol, oc = oi.cpu().numpy(), ol.cpu().numpy(), oc.cpu().numpy()
            for k, i in enumerate(idx):
                ids = oi[k, :ol[k]]
                out[i] = ("".join(self.charset[j] for j in ids), float(oc[k]))
        return out

    # ---- TextSystem.__call__(img, cls=False) for a batch ------------------------------------------------
    def ocr(self, frames):
        """-> list per frame of (list of float32 [4,2] boxes, list of (text, score)) — paddleocr TextSystem output."""
        det = self.detect(frames)
        return self._finish(frames, det)

    def ocr_stream(self, batches, depth=2, rec_span=1):
        """Generator over an iterable of frame batches (cuda uint8 [N,H,W,3]): yields ocr(batch) for each, in order, with
        the detectors of the next `depth` batches in flight (own HIP streams, own workspace slots) while batch k is
        post-processed and recognised — the steady state of a whole-video extraction.  Two detector batches in flight fill each
        other's launch tails (MI355X, 64 x 1080p: +3.6 % over one, three: +1.6 %).  Results are identical to calling ocr() per
        batch.  rec_span > 1 (ragged mode): the crops of that many consecutive batches are recognised together
        (recognize_multi) — same results, larger launches."""
        t = self.ctx.torch
        depth = max(1, int(depth))
        rec_span = max(1, int(rec_span)) if self.rec_mode == "ragged" else 1
        if len(getattr(self, "_det_streams", [])) < depth:
            self._det_streams = self.ctx.side_streams(depth, role="det")      # verified concurrent with each other and with the main stream
        main = t.cuda.current_stream(self.ctx.tdev)
        queue, ready = [], []

        def boxes_of(frames, maps, ev):
            main.wait_event(ev)
            maps.record_stream(main)
            n, h, w, _ = frames.shape
            return frames, [sorted_boxes(r[0]) for r in self.ctx.db_postprocess(maps, h, w, **self.db)]

        prev = []                                             # the span launched last: (boxes per batch, launch handle)

        def flush():
            # launch this span's recognition, THEN read the previous span's results back: the host never waits for launch chains it has
            # just enqueued (results come out one span later, in order)
            cur = ([b for _, b in ready], self.recognize_multi_launch([f for f, _ in ready], [b for _, b in ready]))
            ready.clear()
            outs = collect()
            prev.append(cur)
            return outs

        def collect():
            if not prev:
                return []
            boxes_list, handle = prev.pop()
            rec = self.recognize_multi_collect(handle)
            return [self._filter(b, r) for b, r in zip(boxes_list, rec)]
        for k, frames in enumerate(batches):
            # the iterator may have produced this batch asynchronously on the main stream (GPU decode, crop, non-blocking
            # upload): order the detector stream after it for EVERY batch
            st = self._det_streams[k % depth]
            st.wait_stream(main)
            with t.cuda.stream(st):
                maps = self.det_maps(frames, slot=k % (depth + 1))
                ev = t.cuda.Event()
                ev.record(st)
            frames.record_stream(st)
            queue.append((frames, maps, ev))
            if len(queue) > depth:
                ready.append(boxes_of(*queue.pop(0)))
                if len(ready) >= rec_span:
                    yield from flush()
        while queue:
            ready.append(boxes_of(*queue.pop(0)))
            if len(ready) >= rec_span or not queue:
                yield from flush()
        yield from collect()

    def detect_stream(self, batches, depth=2):
        """ocr_stream()'s detector half: yields detect(batch) for each batch, in order, with the detectors of the next `depth`
        batches in flight — for clients that look at the boxes first and recognise a subset (accurate-mode frame selection)."""
        t = self.ctx.torch
        depth = max(1, int(depth))
        if len(getattr(self, "_det_streams", [])) < depth:
            self._det_streams = self.ctx.side_streams(depth, role="det")      # verified concurrent with each other and with the main stream
        main = t.cuda.current_stream(self.ctx.tdev)
        queue = []

        def boxes_of(frames, maps, ev):
            main.wait_event(ev)
            maps.record_stream(main)
            n, h, w, _ = frames.shape
            return [r[0] for r in self.ctx.db_postprocess(maps, h, w, **self.db)]
        for k, frames in enumerate(batches):
            st = self._det_streams[k % depth]
            st.wait_stream(main)
            with t.cuda.stream(st):
                maps = self.det_maps(frames, slot=k % (depth + 1))
                ev = t.cuda.Event()
                ev.record(st)
            frames.record_stream(st)
            queue.append((frames, maps, ev))
            if len(queue) > depth:
                yield boxes_of(*queue.pop(0))
        while queue:
            yield boxes_of(*queue.pop(0))

    def ocr_from_det(self, frames, det):
        """ocr(frames) for frames whose detect() result `det` is already known (the same boxes come out of the same frame):
        box ordering, crops, recognition, drop_score filter."""
        return self._finish(frames, [list(np.asarray(b, np.float32).reshape(-1, 4, 2)) for b in det])

    def _finish_maps(self, frames, maps, ev):
        main = self.ctx.torch.cuda.current_stream(self.ctx.tdev)
        main.wait_event(ev)
        maps.record_stream(main)
        n, h, w, _ = frames.shape
        res = self.ctx.db_postprocess(maps, h, w, **self.db)
        return self._finish(frames, [r[0] for r in res])

    def _finish(self, frames, det):
        ordered = [sorted_boxes(b) for b in det]
        return self._filter(ordered, self.recognize(frames, ordered))

    def _filter(self, ordered, rec):
        """TextSystem's drop_score filter over ordered boxes + their recognition results -> per frame (boxes, results)."""
        out = []
        for boxes, res in zip(ordered, rec):
            fb, fr = [], []
            for b, r in zip(boxes, res):
                if r[1] >= self.drop_score:
                    fb.append(b)
                    fr.append(r)
            out.append((fb, fr))
        return out
