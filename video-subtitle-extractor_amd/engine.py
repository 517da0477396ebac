"""ctypes binding of libvse_hip.so (include/vse_hip.h) + plan cache on top of the graph compiler.

PyTorch is plumbing here: device memory (torch.empty on cuda), streams and torch.distributed.  All compute
goes through the C ABI; there is NO fallback — if the library or a gfx950 GPU is missing, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import compiler, ir

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSE_LIB_PATH") or os.path.join(_HERE, "libvse_hip.so")      # (VSE_LIB_PATH: ablation builds of tools/)

EXPORTS = [
    "vse_init", "vse_destroy", "vse_last_error", "vse_sizeof_op", "vse_sizeof_view", "vse_abi_version", "vse_is_dev_build",
    "vse_weights_upload", "vse_weights_free", "vse_plan_create", "vse_plan_destroy", "vse_plan_run", "vse_plan_run_ragged",
    "vse_plan_width_levels", "vse_plan_profile", "vse_plan_op_variant", "vse_plan_op_kernel_name", "vse_det_preprocess", "vse_db_workspace_bytes",
    "vse_db_postprocess", "vse_rec_preprocess", "vse_rec_preprocess_scratch_bytes", "vse_ctc_collapse", "vse_ctc_collapse_ragged",
    "vse_det_forward", "vse_rec_forward", "vse_plan_set_source", "vse_plan_takes_frames",
    "vse_rec_graph_create", "vse_graph_launch", "vse_graph_destroy",
]


class VseError(RuntimeError):
    pass


class DbParams(C.Structure):
    _fields_ = [("box_thresh", C.c_double), ("unclip_ratio", C.c_double), ("thresh", C.c_float),
                ("max_candidates", C.c_int), ("min_size", C.c_int)]


class Box(C.Structure):
    _fields_ = [("pts", C.c_float * 2 * 4), ("score", C.c_float), ("frame", C.c_int)]


class Crop(C.Structure):
    _fields_ = [("quad", C.c_float * 2 * 4), ("frame", C.c_int), ("crop_w", C.c_int), ("crop_h", C.c_int),
                ("resized_w", C.c_int), ("rotate", C.c_int)]


CROP_DT = np.dtype([("quad", "<f4", (4, 2)), ("frame", "<i4"), ("crop_w", "<i4"), ("crop_h", "<i4"), ("resized_w", "<i4"),
                    ("rotate", "<i4")])
assert CROP_DT.itemsize == C.sizeof(Crop)

_lib = None


def load_library(path=None):
    """dlopen the C-ABI library and bind signatures.  Works without a GPU (used by the CPU test-suite to check
    that every symbol of include/vse_hip.h is exported and that the record layouts agree)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported BEFORE libvse_hip.so: the torch wheel bundles its own libamdhip64; if ours (linked against
    # /opt/rocm) initialises first, two HIP runtimes end up in the process and one of them sees no device
    import torch  # noqa: F401
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise VseError(f"{path} not found: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first; "
                       "there is no CPU fallback for the OCR hot path")
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise VseError(f"libvse_hip.so does not export {name}")
    if os.environ.get("VSE_DEV_BUILD", "0") == "1" and not lib.vse_is_dev_build():
        # ir.dev_switch honours the compiler's experiment switches under VSE_DEV_BUILD=1; a PRODUCT library ignores its half of them
        # (vse_dev_getenv) and refuses the experimental kernels: the two halves of an A/B arm would silently disagree
        raise VseError(f"VSE_DEV_BUILD=1 but {path} is a product build: rebuild with `VSE_DEV_BUILD=1 python __graft_entry__.py --force` or "
                       "point VSE_LIB_PATH at a development build (tools/build_ab.sh), or unset VSE_DEV_BUILD")
    lib.vse_last_error.restype = C.c_char_p
    lib.vse_sizeof_op.restype = C.c_size_t
    lib.vse_sizeof_view.restype = C.c_size_t
    lib.vse_db_workspace_bytes.restype = C.c_size_t
    lib.vse_db_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.vse_rec_preprocess_scratch_bytes.restype = C.c_size_t
    lib.vse_rec_preprocess_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.vse_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.vse_destroy.argtypes = [C.c_void_p]
    lib.vse_destroy.restype = None
    lib.vse_weights_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vse_weights_free.argtypes = [C.c_void_p, C.c_int]
    lib.vse_plan_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.vse_plan_destroy.argtypes = [C.c_void_p]
    lib.vse_plan_destroy.restype = None
    lib.vse_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.vse_plan_run_ragged.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]
    lib.vse_plan_width_levels.argtypes = [C.c_void_p]
    lib.vse_rec_graph_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.vse_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    lib.vse_graph_destroy.argtypes = [C.c_void_p]
    lib.vse_graph_destroy.restype = None
    lib.vse_plan_set_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64]
    lib.vse_plan_takes_frames.argtypes = [C.c_void_p]
    lib.vse_det_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vse_rec_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vse_plan_profile.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_float)]
    lib.vse_plan_op_variant.argtypes = [C.c_void_p, C.c_int]
    lib.vse_plan_op_kernel_name.argtypes = [C.c_void_p, C.c_int]
    lib.vse_plan_op_kernel_name.restype = C.c_char_p
    lib.vse_det_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.c_void_p]
    lib.vse_db_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(DbParams), C.c_void_p, C.c_size_t, C.POINTER(Box), C.c_int,
                                       C.POINTER(C.c_int), C.c_void_p]
    lib.vse_rec_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                       C.POINTER(Crop), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_size_t, C.c_void_p]
    lib.vse_ctc_collapse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    lib.vse_ctc_collapse_ragged.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
    if lib.vse_sizeof_op() != ir.OP_DT.itemsize or lib.vse_sizeof_view() != ir.VIEW_DT.itemsize:
        raise VseError(f"ABI mismatch: vse_op {lib.vse_sizeof_op()} vs {ir.OP_DT.itemsize}, "
                       f"vse_view {lib.vse_sizeof_view()} vs {ir.VIEW_DT.itemsize}")
    _lib = lib
    return lib


def _check(rc, what):
    if rc < 0:
        raise VseError(f"{what} failed (rc={rc}): {load_library().vse_last_error().decode(errors='replace')}")
    return rc


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise VseError("no HIP device visible: the OCR hot path runs only on MI355X (gfx950); no CPU fallback")
    return torch


class Context:
    """One per (process, device)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.torch = _torch()
        self.device = device
        self.handle = C.c_void_p()
        _check(self.lib.vse_init(device, C.byref(self.handle)), "vse_init")
        self.tdev = self.torch.device("cuda", device)

    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)

    # ---- streams ----------------------------------------------------------------------------------------
    def streams_concurrent(self, a, b, spin_cycles=1_500_000):
        """Do streams a and b really run side by side?  torch hands streams out of a round-robin pool of 32 per priority and the HIP
        runtime multiplexes them over a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default; a queue is bound at a stream's
        FIRST use): two streams on one hardware queue execute in order whatever the program says.  Measured: a short kernel on b is
        launched behind a long spin on a; concurrent = it finishes while the spin still runs."""
        t = self.torch
        x = self.__dict__.setdefault("_probe_buf", t.zeros(64, device=self.tdev))
        for s in (a, b):                       # bind both streams to their hardware queues first (a first use costs milliseconds)
            with t.cuda.stream(s):
                x.add_(1)
        t.cuda.synchronize(self.tdev)
        a0, a1, b1 = (t.cuda.Event(enable_timing=True) for _ in range(3))
        with t.cuda.stream(a):
            a0.record()
            t.cuda._sleep(spin_cycles)
            a1.record()
        with t.cuda.stream(b):
            x.add_(1)
            b1.record()
        t.cuda.synchronize(self.tdev)
        return a0.elapsed_time(b1) < 0.5 * a0.elapsed_time(a1)

    def side_streams(self, n, priority=0, tries=24, role="default"):
        """n streams of the given priority that run concurrently with each other AND with the current stream (verified by
        streams_concurrent), cached per (priority, role): every pipeline of the process gets the same ones for the same role, and two
        roles — the detector batches in flight ("det") and the recogniser's width groups ("rec") — never receive the same stream objects,
        whatever their priorities (with one list per priority, equal priorities made the detector / recogniser overlap serialise
        silently).  A candidate is also checked against the streams of the OTHER roles: preferred when it runs beside all of them
        (the device has a handful of hardware queues: not always possible), accepted after half the tries when it only runs beside
        its own list and the main stream (`side_streams_cross_verified` tells)."""
        t = self.torch
        cache = self.__dict__.setdefault("_side_streams", {})
        have = cache.setdefault((priority, role), [])
        others = [s for k, lst in cache.items() if k != (priority, role) for s in lst]
        main = t.cuda.current_stream(self.tdev)
        budget = tries
        while len(have) < n and tries > 0:
            tries -= 1
            s = t.cuda.Stream(device=self.tdev, priority=priority)
            if any(s.cuda_stream == h.cuda_stream for h in have + others):
                continue
            if not (self.streams_concurrent(main, s) and all(self.streams_concurrent(h, s) for h in have)):
                continue
            cross = all(self.streams_concurrent(o, s) for o in others)
            if cross or tries < budget // 2:
                if not cross:
                    self.side_streams_cross_verified = False
                have.append(s)
        if len(have) < n:
            # nothing on this device runs side by side right now — a profiler that serialises dispatches (rocprofv3 --pmc), one hardware
            # queue, a debugger: concurrency is an optimisation, not a requirement, so the plain pool streams do (results are identical)
            import warnings
            warnings.warn(f"only {len(have)} of {n} HIP streams of priority {priority} verified to run concurrently; using unverified ones")
            self.side_streams_verified = False
            while len(have) < n:
                s = t.cuda.Stream(device=self.tdev, priority=priority)
                if not any(s.cuda_stream == h.cuda_stream for h in have + others) or tries < -64:
                    have.append(s)
                tries -= 1
        return have[:n]

    def close(self):
        if self.handle:
            self.lib.vse_destroy(self.handle)
            self.handle = C.c_void_p()

    # ---- stand-alone stages ---------------------------------------------------------------------------
    def det_preprocess(self, frames_u8, dst_h, dst_w, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), raw=False):
        """frames_u8: cuda uint8 [N,H,W,3] (contiguous or row-pitched view) -> fp16 [N,dst_h,dst_w,8].
        raw=True: resized uint8 values + a ones channel for a net built with input_norm (see vse_det_preprocess)."""
        t = self.torch
        assert frames_u8.dtype == t.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3
        assert frames_u8.stride(3) == 1 and frames_u8.stride(2) == 3
        n, h, w, _ = frames_u8.shape
        out = t.empty((n, dst_h, dst_w, 8), dtype=t.float16, device=self.tdev)
        m = None if raw else (C.c_float * 3)(*mean)
        s = None if raw else (C.c_float * 3)(*std)
        _check(self.lib.vse_det_preprocess(self.handle, C.c_void_p(frames_u8.data_ptr()), n, h, w,
                                           frames_u8.stride(1), frames_u8.stride(0), C.c_void_p(out.data_ptr()),
                                           dst_h, dst_w, m, s, self.stream()), "vse_det_preprocess")
        return out

    def db_postprocess(self, prob, src_h, src_w, thresh=0.3, box_thresh=0.6, unclip_ratio=1.5, max_candidates=1000,
                       min_size=3, max_boxes=None):
        """prob: cuda fp32 [N,H,W] -> list (per frame) of (ndarray[k,4,2] float32, ndarray[k] float32)."""
        t = self.torch
        assert prob.dtype == t.float32 and prob.is_contiguous()
        n, h, w = prob.shape
        nbytes = self.lib.vse_db_workspace_bytes(n, h, w)
        if getattr(self, "_db_ws", None) is None or self._db_ws.numel() < nbytes:
            self._db_ws = t.empty(nbytes, dtype=t.uint8, device=self.tdev)
        max_boxes = max_boxes or n * 1024
        if getattr(self, "_db_boxes_cap", 0) < max_boxes:           # reused across calls (2.6 MB for 64 frames; only nb.value
            self._db_boxes = (Box * max_boxes)()                    # records are read back, and they are copied below)
            self._db_boxes_cap = max_boxes
        boxes = self._db_boxes
        nb = C.c_int(0)
        prm = DbParams(box_thresh, unclip_ratio, thresh, max_candidates, min_size)
        _check(self.lib.vse_db_postprocess(self.handle, C.c_void_p(prob.data_ptr()), n, h, w, src_h, src_w,
                                           C.byref(prm), C.c_void_p(self._db_ws.data_ptr()), nbytes, boxes,
                                           max_boxes, C.byref(nb), self.stream()), "vse_db_postprocess")
        arr = np.frombuffer(boxes, dtype=np.dtype([("pts", "<f4", (4, 2)), ("score", "<f4"), ("frame", "<i4")]),
                            count=nb.value)
        # records come grouped by frame in ascending order: split by counts instead of 64 boolean masks
        out = []
        if nb.value and np.all(np.diff(arr["frame"]) >= 0):
            cnt = np.bincount(arr["frame"], minlength=n)
            end = np.cumsum(cnt)
            pts, sc = arr["pts"].copy(), arr["score"].copy()
            for f in range(n):
                out.append((pts[end[f] - cnt[f]:end[f]], sc[end[f] - cnt[f]:end[f]]))
        else:
            for f in range(n):
                sel = arr[arr["frame"] == f]
                out.append((sel["pts"].copy(), sel["score"].copy()))
        return out

    def rec_preprocess(self, frames_u8, crops, rec_h, rec_w, out=None):
        """crops: list of dict(quad[4][2], frame, crop_w, crop_h, resized_w, rotate) -> fp16 [n,rec_h,rec_w,8].
        out: write into this contiguous fp16 [n,rec_h,rec_w,8] tensor (rows of a larger batch whose other rows are cut from
        another frame tensor) instead of allocating one."""
        t = self.torch
        n = len(crops)
        # the vse_crop records are filled through a numpy view of the ctypes array (same bytes, no per-field Python loop:
        # this sits on the GPU-idle path between the detector's boxes and the first recogniser launch)
        arr = (Crop * n)()
        rec = np.frombuffer(arr, dtype=CROP_DT, count=n)
        rec["quad"] = np.asarray([c["quad"] for c in crops], dtype=np.float32).reshape(n, 4, 2)
        for name in ("frame", "crop_w", "crop_h", "resized_w", "rotate"):
            rec[name] = [c[name] for c in crops]
        mw = max(1, int(rec["crop_w"].max())) if n else 1
        mh = max(1, int(rec["crop_h"].max())) if n else 1
        nbytes = self.lib.vse_rec_preprocess_scratch_bytes(n, mw, mh)
        crop_ws = t.empty(nbytes, dtype=t.uint8, device=self.tdev)     # per call: groups may run on different streams
        if out is None:
            out = t.empty((n, rec_h, rec_w, 8), dtype=t.float16, device=self.tdev)
        assert out.dtype == t.float16 and out.is_contiguous() and tuple(out.shape) == (n, rec_h, rec_w, 8)
        nf, h, w, _ = frames_u8.shape
        _check(self.lib.vse_rec_preprocess(self.handle, C.c_void_p(frames_u8.data_ptr()), nf, h, w,
                                           frames_u8.stride(1), frames_u8.stride(0), arr, n,
                                           C.c_void_p(out.data_ptr()), rec_h, rec_w,
                                           C.c_void_p(crop_ws.data_ptr()), nbytes, self.stream()),
               "vse_rec_preprocess")
        return out

    def ctc_collapse(self, idx_maxp, tlen=None):
        """idx_maxp: cuda [B,1,T,2] (int32 idx / fp32 prob bit-pairs, as float32 tensor) ->
        (idx int32 [B,T], len int32 [B], conf fp32 [B]) cuda tensors.  tlen: cuda int32 [B] sequence lengths of a ragged batch."""
        t = self.torch
        b, tt = idx_maxp.shape[0], idx_maxp.shape[2]
        oi = t.zeros((b, tt), dtype=t.int32, device=self.tdev)
        ol = t.empty((b,), dtype=t.int32, device=self.tdev)
        oc = t.empty((b,), dtype=t.float32, device=self.tdev)
        if tlen is not None:
            assert tlen.dtype == t.int32 and tlen.is_contiguous() and tlen.numel() == b
        _check(self.lib.vse_ctc_collapse_ragged(self.handle, C.c_void_p(idx_maxp.data_ptr()), b, tt,
                                                C.c_void_p(tlen.data_ptr()) if tlen is not None else None,
                                                C.c_void_p(oi.data_ptr()), C.c_void_p(ol.data_ptr()),
                                                C.c_void_p(oc.data_ptr()), self.stream()), "vse_ctc_collapse")
        return oi, ol, oc


class Net:
    """One model (descriptor + fp32 weights) with plans compiled per static input shape."""

    def __init__(self, ctx: Context, desc, weights, fetch_cols=(0,), want_probs=True, hilo=False, ragged=False, input_norm=None,
                 fuse_preprocess=False, chain=None, tail2=None):
        """hilo=True: conv weights as fp16 hi + lo pairs (compiler.compile_model): ~22-bit weights, twice the MFMA work.
        ragged=True (recognisers): every plan takes a per-sample width vector (run(x, widths=...)); a sample's outputs do
        not depend on the batch it rides in (compiler.compile_model(ragged=True))."""
        self.ctx = ctx
        self.ragged = bool(ragged)
        self.input_norm = input_norm          # (mean3, std3): the net takes det_preprocess(raw=True) input (compiler.compile_model)
        # the stem conv resizes the uint8 frames itself (needs input_norm): the plans take FRAMES — det_forward / profile_frames;
        # run(x) with a pre-processed tensor is refused by the library
        self.fuse_preprocess = bool(fuse_preprocess) and input_norm is not None
        self.desc = desc
        self.weights = weights
        self.fetch_cols = fetch_cols
        self.want_probs = want_probs
        self.hilo = bool(hilo)
        self.tail2 = tail2                    # False: the server detector's two head deconvs as separate launches (compiler F_TAIL2; tests)
        self.chain = chain                    # None: 1x1 / depthwise chains (OP_CHAIN) for hi + lo nets; False / True forces it
        self.store = compiler.WeightStore()
        self.plans = {}
        self.wid = None
        self.wbytes = 0
        self.ws = {}          # one workspace per plan: plans of one net may run concurrently on different streams
        self.ws_budget = int(float(os.environ.get("VSE_WS_BUDGET_GB", "64")) * (1 << 30))     # per net; LRU beyond it
        self.ws_evictions = 0 # workspaces dropped by that LRU so far (tools/soak.py reports it)
        self.fallbacks = {}   # rewrites compile_model had to abandon for this graph (tail2 / se_lateral): remembered for the next shape

    def program(self, n, h, w):
        key = (n, h, w)
        if key not in self.plans:
            try:
                prog = compiler.compile_model(self.desc, self.weights, n, h, w, self.fetch_cols, self.want_probs,
                                              self.store, hilo=self.hilo, ragged=self.ragged, input_norm=self.input_norm,
                                              fuse_preprocess=self.fuse_preprocess, chain=self.chain,
                                              tail2=self.fallbacks.get("tail2", self.tail2),
                                              se_lateral=self.fallbacks.get("se_lateral"), fallbacks=self.fallbacks)
            except compiler.UnsupportedGraph:
                if not self.fuse_preprocess or self.plans:
                    raise
                self.fuse_preprocess = False      # a graph whose feed is not read by one 3x3 stem conv: keep the pre-processing pass
                return self.program(n, h, w)
            self.plans[key] = [prog, None]
        return self.plans[key][0]

    def _ensure(self, key):
        prog, handle = self.plans[key]
        lib = self.ctx.lib
        if self.wid is None or len(self.store.blob) != self.wbytes:
            # (re-)upload the shared weight blob; plans created against an older blob are rebuilt lazily
            blob = self.store.array()
            self._drop_graphs()               # captured graphs replay kernels against the plan handles / weight blob freed below
            if self.wid is not None:
                for k, (p_, h_) in self.plans.items():
                    if h_ is not None:
                        lib.vse_plan_destroy(h_)
                        self.plans[k][1] = None
                lib.vse_weights_free(self.ctx.handle, self.wid)
            self.wid = _check(lib.vse_weights_upload(self.ctx.handle, blob.ctypes.data_as(C.c_void_p), blob.nbytes),
                              "vse_weights_upload")
            self.wbytes = len(self.store.blob)
            handle = None
        if handle is None:
            h = C.c_void_p()
            ops = np.ascontiguousarray(prog.ops)
            _check(lib.vse_plan_create(self.ctx.handle, self.wid, ops.ctypes.data_as(C.c_void_p), len(ops),
                                       prog.ws_bytes, C.byref(h)), "vse_plan_create")
            self.plans[key][1] = h
            handle = h
        return prog, handle

    GRAPH_CACHE_MAX = 32      # captured recogniser graphs kept per net (each pins input, workspace and output buffers)

    def _drop_graphs(self, keep=0):
        """Destroy captured recogniser graphs, oldest first, until `keep` are left (after a device synchronisation: they may be
        in flight)."""
        graphs = self.__dict__.get("_graphs")
        if not graphs or len(graphs) <= keep:
            return
        self.ctx.torch.cuda.synchronize(self.ctx.tdev)
        while len(graphs) > keep:
            bufs = graphs.pop(next(iter(graphs)))
            if bufs["graph"] is not None:
                self.ctx.lib.vse_graph_destroy(bufs["graph"])

    def _workspace(self, key, prog, slot):
        """One workspace per (plan, slot): a plan is stateless between runs, so the same plan may run concurrently on
        several streams as long as every in-flight run has its own slot."""
        t = self.ctx.torch
        k = (key, slot)
        ent = self.ws.pop(k, None)            # (view handed to the plan, stream of its last use)
        cur = t.cuda.current_stream(self.ctx.tdev)
        if ent is None:
            # a long video meets many recogniser shapes (crops x width bucket), each with a workspace of its own (they are
            # zero-initialised and their padding regions must stay zero, so plans cannot share one): the total is kept under a
            # budget, least recently used first.  No device synchronisation and, where possible, no allocator round trip:
            #  * a victim last used on THIS stream and large enough is re-zeroed and reused in place — the memset and the new plan's
            #    launches queue behind its last launch on that stream (a recogniser slot always runs on the same side stream);
            #  * otherwise victims are dropped (every use records its stream, so the caching allocator holds the memory back until
            #    its launches finished) and a fresh buffer is allocated.
            # (Round 6: a synchronising eviction drained the detector batches in flight, -4 % on a stream at its budget; dropping
            # without reuse made a host that runs a span ahead wait in hipMalloc for blocks still held back: -4 % again.)
            need = max(prog.ws_bytes, 256)
            total = sum(int(e[0].untyped_storage().nbytes()) for e in self.ws.values())
            ws = None
            if self.ws and total + need > self.ws_budget:
                for vk, (vws, vstream) in self.ws.items():           # least recently used first
                    cap = int(vws.untyped_storage().nbytes())
                    if vstream == cur.cuda_stream and need <= cap <= 4 * need:
                        del self.ws[vk]
                        self.ws_evictions += 1
                        base = t.empty(0, dtype=t.uint8, device=self.ctx.tdev).set_(vws.untyped_storage(), 0, (cap,))
                        ws = base[:need]
                        ws.zero_()
                        break
            if ws is None:
                while self.ws and total + need > self.ws_budget:
                    total -= int(self.ws.pop(next(iter(self.ws)))[0].untyped_storage().nbytes())
                    self.ws_evictions += 1
                ws = t.zeros(need, dtype=t.uint8, device=self.ctx.tdev)
        else:
            ws = ent[0]
        ws.record_stream(cur)
        self.ws[k] = (ws, cur.cuda_stream)    # (re-)inserted last: dict order = least recently used first
        return ws

    def _ext(self, prog, x):
        t = self.ctx.torch
        outs = [t.empty((o["n"], o["h"], o["w"], o["ld"]), dtype=t.float32, device=self.ctx.tdev)
                for o in prog.outputs]
        ptrs = (C.c_void_p * (1 + len(outs)))(x.data_ptr(), *[o.data_ptr() for o in outs])
        return outs, ptrs

    def _width_table(self, prog, widths, n, w):
        """Device int32 [levels][n] for a ragged plan (None for an ordinary one)."""
        if not self.ragged:
            if widths is not None:
                raise VseError("per-sample widths given to a net that was not built with ragged=True")
            return None
        if widths is None:
            widths = np.full(n, w, np.int32)          # a uniform batch: every sample as wide as the tensor
        return self._upload_i32(prog.width_table(widths))

    def _upload_i32(self, tab, out=None):
        """Stream-ordered upload of a small int32 host table through a ring of reusable pinned buffers.  A pageable
        `tensor.to(device)` blocks the host until the stream has drained (every recogniser group would wait for the previous
        group's ~80 kernels); a pinned buffer allocated per call costs as much as that stall (round 3 log) — the ring costs
        neither: slot i is reused after 8 further uploads, guarded by the event recorded behind its last copy."""
        t = self.ctx.torch
        ring = self.__dict__.setdefault("_pin_ring", {"slots": [None] * 8, "next": 0})
        i = ring["next"] % len(ring["slots"])
        ring["next"] += 1
        slot = ring["slots"][i]
        if slot is None or slot[0].numel() < tab.size:
            if slot is not None:
                slot[1].synchronize()
            slot = ring["slots"][i] = (t.empty(max(int(tab.size), 1024), dtype=t.int32).pin_memory(), t.cuda.Event())
        else:
            slot[1].synchronize()             # the copy that last read this slot (8 uploads ago) has long finished
        pin = slot[0][:tab.size].view(tab.shape)
        pin.numpy()[...] = tab
        dev = out if out is not None else t.empty(tab.shape, dtype=t.int32, device=self.ctx.tdev)
        dev.copy_(pin, non_blocking=True)
        slot[1].record(t.cuda.current_stream(self.ctx.tdev))
        return dev

    def run(self, x, slot=0, widths=None):
        """x: cuda fp16 [N,H,W,8] NHWC (3 real channels).  Returns list of fp32 cuda tensors (prog.outputs order).
        widths (ragged nets): per-sample input width (<= W; x is zero right of it); self.last_tlen then holds the device
        int32 [N] sequence lengths of the head's output."""
        t = self.ctx.torch
        assert x.dtype == t.float16 and x.is_contiguous() and x.shape[3] == 8, (x.dtype, x.shape)
        n, h, w, _ = x.shape
        self.program(n, h, w)
        prog, handle = self._ensure((n, h, w))
        outs, ptrs = self._ext(prog, x)
        ws = self._workspace((n, h, w), prog, slot)
        wt = self._width_table(prog, widths, n, w)
        self.last_tlen = wt[prog.out_level] if wt is not None else None
        _check(self.ctx.lib.vse_plan_run_ragged(handle, C.c_void_p(ws.data_ptr()), ptrs, len(ptrs),
                                                C.c_void_p(wt.data_ptr()) if wt is not None else None, self.ctx.stream()),
               "vse_plan_run")
        return outs

    def det_forward(self, frames_u8, dst_h, dst_w, slot=0):
        """Model-level call of a detector net (vse_det_forward): cuda uint8 [N,H,W,3] -> cuda fp32 probability maps [N,dst_h,dst_w]."""
        t = self.ctx.torch
        assert frames_u8.dtype == t.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3
        assert frames_u8.stride(3) == 1 and frames_u8.stride(2) == 3
        n, h, w, _ = frames_u8.shape
        self.program(n, dst_h, dst_w)
        prog, handle = self._ensure((n, dst_h, dst_w))
        assert len(prog.outputs) == 1 and prog.outputs[0]["kind"] == "map" and prog.outputs[0]["ld"] == 1, "not a one-map detector plan"
        x = None if self.fuse_preprocess else t.empty((n, dst_h, dst_w, 8), dtype=t.float16, device=self.ctx.tdev)
        prob = t.empty((n, dst_h, dst_w), dtype=t.float32, device=self.ctx.tdev)
        ws = self._workspace((n, dst_h, dst_w), prog, slot)
        _check(self.ctx.lib.vse_det_forward(self.ctx.handle, handle, C.c_void_p(ws.data_ptr()), C.c_void_p(frames_u8.data_ptr()), n, h, w,
                                            frames_u8.stride(1), frames_u8.stride(0), dst_h, dst_w, 1 if self.input_norm is not None else 0,
                                            C.c_void_p(x.data_ptr()) if x is not None else None, C.c_void_p(prob.data_ptr()),
                                            self.ctx.stream()), "vse_det_forward")
        return prob

    def profile_frames(self, frames_u8, dst_h, dst_w, slot=0):
        """profile() of a detector net fed FRAMES: pre-processing pass + plan for an ordinary net, the plan alone when its stem
        pre-processes.  -> (ms per op, program, kernel names); self.last_outs = the outputs."""
        if not self.fuse_preprocess:
            return self.profile(self.ctx.det_preprocess(frames_u8, dst_h, dst_w, raw=self.input_norm is not None), slot)
        t = self.ctx.torch
        n, h, w, _ = frames_u8.shape
        self.program(n, dst_h, dst_w)
        prog, handle = self._ensure((n, dst_h, dst_w))
        _check(self.ctx.lib.vse_plan_set_source(handle, h, w, frames_u8.stride(1), frames_u8.stride(0)), "vse_plan_set_source")
        outs = [t.empty((o["n"], o["h"], o["w"], o["ld"]), dtype=t.float32, device=self.ctx.tdev) for o in prog.outputs]
        ptrs = (C.c_void_p * (1 + len(outs)))(frames_u8.data_ptr(), *[o.data_ptr() for o in outs])
        ms = (C.c_float * len(prog.ops))()
        ws = self._workspace((n, dst_h, dst_w), prog, slot)
        self.last_outs = outs
        _check(self.ctx.lib.vse_plan_profile(handle, C.c_void_p(ws.data_ptr()), ptrs, len(ptrs), None, self.ctx.stream(), ms),
               "vse_plan_profile")
        names = [self.ctx.lib.vse_plan_op_kernel_name(handle, i).decode() for i in range(len(prog.ops))]
        return np.array(ms[:], dtype=np.float32), prog, names

    def rec_forward(self, x, widths=None, slot=0):
        """Model-level call of a recogniser net built with want_probs=False (vse_rec_forward): fp16 [B,h,w,8] (+ per-sample widths of
        a ragged net) -> (class ids int32 [B,T], lengths int32 [B], mean confidences fp32 [B]) cuda tensors."""
        t = self.ctx.torch
        assert x.dtype == t.float16 and x.is_contiguous() and x.shape[3] == 8
        n, h, w, _ = x.shape
        self.program(n, h, w)
        prog, handle = self._ensure((n, h, w))
        assert len(prog.outputs) == 1 and prog.outputs[0]["kind"] == "idx_maxp"
        tt = prog.outputs[0]["w"]
        wt = self._width_table(prog, widths, n, w)
        idx_maxp = t.empty((n, 1, tt, 2), dtype=t.float32, device=self.ctx.tdev)
        oi = t.zeros((n, tt), dtype=t.int32, device=self.ctx.tdev)
        ol = t.empty((n,), dtype=t.int32, device=self.ctx.tdev)
        oc = t.empty((n,), dtype=t.float32, device=self.ctx.tdev)
        ws = self._workspace((n, h, w), prog, slot)
        _check(self.ctx.lib.vse_rec_forward(self.ctx.handle, handle, C.c_void_p(ws.data_ptr()), C.c_void_p(x.data_ptr()),
                                            C.c_void_p(wt.data_ptr()) if wt is not None else None, prog.out_level,
                                            C.c_void_p(idx_maxp.data_ptr()), n, tt, C.c_void_p(oi.data_ptr()), C.c_void_p(ol.data_ptr()),
                                            C.c_void_p(oc.data_ptr()), self.ctx.stream()), "vse_rec_forward")
        return oi, ol, oc

    def rec_graph_buffers(self, n, h, w, slot=0):
        """Fixed buffers of the HIP-graph form of rec_forward for plan key (n, h, w) and workspace slot: the caller fills
        bufs["x"] (recogniser input, e.g. rec_preprocess(out=bufs["x"])) and calls rec_forward_graph(bufs, widths)."""
        t = self.ctx.torch
        key = (n, h, w, slot)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        if key in self._graphs:
            self._graphs[key] = self._graphs.pop(key)      # most recently used last
        else:
            self._drop_graphs(keep=self.GRAPH_CACHE_MAX - 1)
            self.program(n, h, w)
            prog, handle = self._ensure((n, h, w))
            tt = prog.outputs[0]["w"]
            dev = self.ctx.tdev
            self._graphs[key] = dict(
                x=t.zeros((n, h, w, 8), dtype=t.float16, device=dev), wt=t.zeros((len(prog.wlevels), n), dtype=t.int32, device=dev),
                idx_maxp=t.empty((n, 1, tt, 2), dtype=t.float32, device=dev), oi=t.zeros((n, tt), dtype=t.int32, device=dev),
                ol=t.zeros((n,), dtype=t.int32, device=dev), oc=t.zeros((n,), dtype=t.float32, device=dev), graph=None, tt=tt,
                wsbuf=t.zeros(max(prog.ws_bytes, 256), dtype=t.uint8, device=dev), prog=prog, handle=handle)
        return self._graphs[key]

    def rec_forward_graph(self, bufs, widths):
        """rec_forward through a HIP graph captured on first use (on the CURRENT stream, which must not be the default one);
        -> clones of (class ids, lengths, confidences) (the fixed output buffers are overwritten by the next launch)."""
        t = self.ctx.torch
        prog = bufs["prog"]
        self._upload_i32(prog.width_table(widths), out=bufs["wt"])
        if bufs["graph"] is None:
            g = C.c_void_p()
            n = bufs["x"].shape[0]
            _check(self.ctx.lib.vse_rec_graph_create(self.ctx.handle, bufs["handle"], C.c_void_p(bufs["wsbuf"].data_ptr()),
                                                     C.c_void_p(bufs["x"].data_ptr()), C.c_void_p(bufs["wt"].data_ptr()), prog.out_level,
                                                     C.c_void_p(bufs["idx_maxp"].data_ptr()), n, bufs["tt"], C.c_void_p(bufs["oi"].data_ptr()),
                                                     C.c_void_p(bufs["ol"].data_ptr()), C.c_void_p(bufs["oc"].data_ptr()), self.ctx.stream(),
                                                     C.byref(g)), "vse_rec_graph_create")
            bufs["graph"] = g
        _check(self.ctx.lib.vse_graph_launch(bufs["graph"], self.ctx.stream()), "vse_graph_launch")
        return bufs["oi"].clone(), bufs["ol"].clone(), bufs["oc"].clone()

    def profile(self, x, slot=0, widths=None):
        """Per-op milliseconds (HIP events) for one run; returns (ms ndarray, program, kernel name per op)."""
        n, h, w, _ = x.shape
        self.program(n, h, w)
        prog, handle = self._ensure((n, h, w))
        outs, ptrs = self._ext(prog, x)
        ms = (C.c_float * len(prog.ops))()
        ws = self._workspace((n, h, w), prog, slot)
        wt = self._width_table(prog, widths, n, w)
        self.last_tlen = wt[prog.out_level] if wt is not None else None
        self.last_outs = outs                 # the profiled run's outputs (same values as run())
        _check(self.ctx.lib.vse_plan_profile(handle, C.c_void_p(ws.data_ptr()), ptrs, len(ptrs),
                                             C.c_void_p(wt.data_ptr()) if wt is not None else None, self.ctx.stream(), ms),
               "vse_plan_profile")
        variants = [self.ctx.lib.vse_plan_op_kernel_name(handle, i).decode() for i in range(len(prog.ops))]
        return np.array(ms[:], dtype=np.float32), prog, variants
