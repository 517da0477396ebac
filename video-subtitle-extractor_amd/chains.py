"""Chain lowering: runs of 1x1 convs and depthwise convs -> ONE `OP_CHAIN` launch (csrc/chain.hip).

The mobile detectors the reference runs by default (backend/tools/paddle_model_config.py:53-58: V4/ch_det_fast, V3/ch_det_fast)
are stacks of  1x1 expand -> k x k depthwise -> 1x1 project (+ residual)  (MobileNetV3) or  depthwise -> 1x1  (PP-LCNetV3) units.
Layer by layer every unit writes and re-reads its widest tensors and rounds each of them to fp16; as a chain the intermediates
live in LDS in (effectively) fp32 and HBM sees the chain's input and the tensors other ops need — see the kernel's header.

Host side of the design (this file):
  * chain discovery: from a conv whose input exists as a tensor, follow the single data path while the next consumer is a
    1x1 stride-1 conv or a 3x3 / 5x5 depthwise conv with 'same' padding (stride 1 or 2) over multiples of 8 channels; the
    epilogue of every stage (bias / BN / learnable affine / activation / residual) is absorbed exactly like a conv op's;
    a residual must come from a channel-minor buffer of the same chain (the block input);
  * regions: the final stage's output tile is walked BACK through the stages: a depthwise stage of kernel k, stride s, padding p
    turns an output region (origin t*T - a, extent E) into the input region (origin t*(T s) - (a s + p), extent (E - 1) s + k);
  * LDS plan: weight image first, then the stage buffers placed first-fit by liveness; the tile size is the largest one whose
    plan fits the budget of two blocks per CU (falls back to one);
  * the blob: descriptor words + LDS image (MFMA weight fragments in lane order, hi and lo passes; depthwise weight records).
"""
import os
import struct

import numpy as np

from . import ir
from .ir import dev_switch as _dev_switch

CHAIN = _dev_switch("VSE_CHAIN", "1") != "0"
CHAIN_MAX_STAGES = int(_dev_switch("VSE_CHAIN_MAXSTAGES", "6"))
CHAIN_HEAD = _dev_switch("VSE_CHAIN_HEAD", "1") != "0"               # the DB head's two transposed convs as one chain (try_lower_head_tail)
CHAIN_LDS_2 = 76 * 1024          # two blocks per CU (160 KiB of LDS)
CHAIN_LDS_1 = 150 * 1024
CHAIN_TILES = [(8, 32), (16, 16), (8, 16), (4, 32), (4, 16), (2, 32), (4, 8), (2, 16), (2, 8)]
if _dev_switch("VSE_CHAIN_TILE"):                  # experiments: force the tile, e.g. "4,16"
    CHAIN_TILES = [tuple(int(v) for v in os.environ["VSE_CHAIN_TILE"].split(","))]
CHAIN_TILE_CYC = float(_dev_switch("VSE_CHAIN_TILECYC", "2400"))    # fixed cycles of a tile (input store, prefetch issue, turn-over)
CHAIN_STAGE_CYC = float(_dev_switch("VSE_CHAIN_STAGECYC", "600"))   # fixed cycles of a stage (descriptor lanes, barrier)
CHAIN_BLOCKS_CU = int(_dev_switch("VSE_CHAIN_BLOCKS", "3"))         # blocks per CU the kernel's registers allow (VSE_CHAIN_LB in chain.hip)
CHAIN_ONE_BLOCK = float(_dev_switch("VSE_CHAIN_ONEBLOCK", "2.4"))   # cost factor of a plan that leaves one block per CU
CHAIN_TWO_BLOCKS = float(_dev_switch("VSE_CHAIN_TWOBLOCKS", "1.4"))  # ... two
# segmentation: a chain's estimated time is weighted against the layer-by-layer ops it replaces; < 1 prefers chains (they round a
# tensor to fp16 once per chain instead of once per layer: the detector's box parity) even where they are not faster
CHAIN_TIME_WEIGHT = float(_dev_switch("VSE_CHAIN_WEIGHT", "0.5"))
CHAIN_OP_TBS = 2.5e6            # bytes per microsecond an un-fused streaming op reaches on this chip (measured: 2-3 TB/s), + 6 us per launch


def rup(x, m):
    return (x + m - 1) // m * m


def conv_wrow(f):
    """csrc/conv_common.h conv_wrow: swap bits 2 and 3 (the lane that supplies weight row f reads cout conv_wrow(f))."""
    return (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1)


def f2i(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def pw_fragments(mat):
    """[coutp32][Kp16] float64 -> fp16 fragments [pass hi, lo][ct][ks][lane 64][8] of v_mfma_f32_32x32x16_f16's A operand."""
    coutp, kp = mat.shape
    nct, nks = coutp // 32, kp // 16
    hi = mat.astype(np.float16)
    lo = (mat - hi.astype(np.float64)).astype(np.float16)
    lane = np.arange(64)
    rows = np.array([conv_wrow(int(f)) for f in lane & 31])
    out = np.zeros((2, nct, nks, 64, 8), np.float16)
    for p_, src in enumerate((hi, lo)):
        for ct in range(nct):
            for ks in range(nks):
                k0 = ks * 16 + 8 * (lane >> 5)
                out[p_, ct, ks] = src[ct * 32 + rows][np.arange(64)[:, None], k0[:, None] + np.arange(8)[None, :]]
    return out


def dw_rec(k):
    """floats per channel record of a depthwise stage in the LDS image: k*k weights, the bias, padding to whole 16-byte vectors."""
    return rup(k * k + 1, 4)


def regions_backward(stages, th, tw):
    """Per stage OUTPUT region and chain-input region for a final tile th x tw: list of dict(T, a, E) per axis, index 0 = chain input,
    index j + 1 = output of stage j."""
    regs = [None] * (len(stages) + 1)
    cur = {"Th": th, "ah": 0, "Eh": th, "Tw": tw, "aw": 0, "Ew": tw}
    regs[len(stages)] = dict(cur)
    for j in range(len(stages) - 1, -1, -1):
        st = stages[j]
        if st["type"] == "dw":
            k, s, p = st["k"], st["s"], st["k"] // 2
            cur = {"Th": cur["Th"] * s, "ah": cur["ah"] * s + p, "Eh": (cur["Eh"] - 1) * s + k,
                   "Tw": cur["Tw"] * s, "aw": cur["aw"] * s + p, "Ew": (cur["Ew"] - 1) * s + k}
        regs[j] = dict(cur)
    return regs


class ChainMixin:
    """Methods of compiler.Compiler (kept in their own file)."""

    def _chain_candidate(self, i):
        op = self.ops[i]
        t = op["type"]
        if t not in ("conv2d", "depthwise_conv2d") or not self.live[i] or i in self.done:
            return None
        a = op["attrs"]
        if a.get("out_gate") is not None:
            return None
        w = self.W[op["in"]["Filter"][0]]
        sh, sw = a["strides"]
        pads = a["paddings"]
        ph, pw = (pads[0], pads[1]) if len(pads) == 2 else (pads[0], pads[2])
        groups = a.get("groups", 1)
        if t == "depthwise_conv2d" or (groups > 1 and groups == w.shape[0] and w.shape[1] == 1):
            if self.dwpw_eligible(i):
                return None          # depthwise -> 1x1 runs as ONE streaming kernel (conv_dwpw.hip): faster than an LDS-resident chain
            k = w.shape[2]
            if w.shape[2] == w.shape[3] and k in (3, 5) and sh == sw and sh in (1, 2) and ph == pw == k // 2 and w.shape[0] % 8 == 0:
                return dict(type="dw", k=int(k), s=int(sh), cin=int(w.shape[0]), cout=int(w.shape[0]))
            return None
        if groups == 1 and tuple(w.shape[2:]) == (1, 1) and (sh, sw) == (1, 1) and (ph, pw) == (0, 0) and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
            return dict(type="pw", k=1, s=1, cin=int(w.shape[1]), cout=int(w.shape[0]))
        return None

    def _chain_build(self, i0, max_stages):
        """Greedy chain from conv op i0 (mutates self.done through absorb_epilogue; the caller snapshots).  -> list of stages."""
        stages = []
        op = self.ops[i0]
        cur_name = op["in"]["Input"][0]
        cm_names = {}                         # tensor name -> index of the buffer (0 = chain input, j + 1 = output of stage j)
        i = i0
        first = self._chain_candidate(i0)
        if first is None:
            return []
        if first["type"] == "pw":
            cm_names[cur_name] = 0
        cur_c = first["cin"]
        while len(stages) < max_stages:
            cand = self._chain_candidate(i)
            if cand is None or self.ops[i]["in"]["Input"][0] != cur_name or cand["cin"] != cur_c:
                break
            if stages and stages[-1]["type"] == "dw" and cand["type"] == "dw":
                break
            o = self.ops[i]
            outname = o["out"]["Output"][0]
            before = set(self.done)
            self.chain_res = cm_names if cand["type"] == "pw" else None
            try:
                ep = self.absorb_epilogue(outname, i, cand["cout"], allow_res=cand["type"] == "pw")
            finally:
                self.chain_res = None
            res = ep["res"]
            if res is not None and not (isinstance(res, tuple) and res[0] == "chain"):
                self.done = before                         # a residual from outside the chain: this conv stays an ordinary op
                break
            if ep["act2"] not in (ir.ACT_NONE, ir.ACT_RELU):
                self.done = before
                break
            rb = cm_names[res[1]] if res is not None else -1
            st = dict(cand, op=i, wname=o["in"]["Filter"][0], ep=ep, in_name=cur_name, out_name=ep["out_name"], res_buf=rb, res_abs=rb)
            stages.append(st)
            self.done.add(i)
            cur_name, cur_c = ep["out_name"], cand["cout"]
            if cur_name in self.placement or cur_name in self.fetched_names:
                break
            nxt = [j for j in self._live_consumers(cur_name)
                   if j not in self.done and self._chain_candidate(j) is not None and self.ops[j]["in"]["Input"][0] == cur_name]
            if not nxt:
                break
            i = min(nxt)
            nc = self._chain_candidate(i)
            if nc["type"] == "pw":
                cm_names[cur_name] = len(stages)          # output buffer of the stage just added feeds a PW: channel-minor
        return stages

    def _chain_plan(self, stages, H, W, in_lo):
        """Pick the tile and place the LDS buffers.  -> plan dict or None when nothing fits."""
        n = len(stages)
        # image size at every buffer's resolution
        dims = [(H, W)]
        for st in stages:
            h, w = dims[-1]
            if st["type"] == "dw" and st["s"] == 2:
                h, w = (h + 2 * (st["k"] // 2) - st["k"]) // 2 + 1, (w + 2 * (st["k"] // 2) - st["k"]) // 2 + 1
            dims.append((h, w))
        # weights image
        wbytes = 0
        for st in stages:
            if st["type"] == "pw":
                st["nks"], st["nct"] = rup(st["cin"], 16) // 16, rup(st["cout"], 32) // 32
                st["w_lds"] = wbytes
                wbytes += 2 * st["nct"] * st["nks"] * 1024
                st["b_lds"] = wbytes
                wbytes += st["nct"] * 32 * 4
            else:
                st["w_lds"] = wbytes                      # per-channel records [k*k weights, bias, padding] of 12 / 28 floats
                wbytes += st["cin"] * dw_rec(st["k"]) * 4
        wbytes = rup(wbytes, 16)
        # buffer kinds: buffer j feeds stage j (j < n); kind follows the consumer; the last stage has no LDS output
        best = None
        oh, ow = dims[-1]
        for th, tw in CHAIN_TILES:
            if th > rup(oh, 2) * 2 and (th, tw) != CHAIN_TILES[-1]:
                continue
            regs = regions_backward(stages, th, tw)
            bufs = []
            for j in range(n):
                r = regs[j]
                P = r["Eh"] * r["Ew"]
                c = stages[j]["cin"]
                if stages[j]["type"] == "pw":
                    cp = rup(c, 16)
                    stride = 2 * cp + 16
                    lo = in_lo if j == 0 else True
                    size = P * stride * (2 if lo else 1)
                    bufs.append(dict(kind=0, C=c, Cp=cp, stride=stride, P=P, lo=lo, size=rup(size, 16), reg=r, dims=dims[j]))
                else:
                    stride = rup(P, 32)
                    bufs.append(dict(kind=1, C=c, Cp=c, stride=stride, P=P, lo=False, size=rup(c * stride * 4, 16), reg=r, dims=dims[j]))
            # liveness: buffer j is written by stage j - 1 (input load = -1), read by stage j and by any residual reader
            last = list(range(n))
            for j, st in enumerate(stages):
                if st["res_buf"] >= 0:
                    last[st["res_buf"]] = max(last[st["res_buf"]], j)
            placed, total = [], wbytes
            order = sorted(range(n), key=lambda j: (j - 1, -bufs[j]["size"]))
            for j in order:
                first_, last_ = j - 1, last[j]
                off = wbytes
                for (po, ps, pf, pl) in sorted(placed):
                    if pl < first_ or pf > last_:
                        continue
                    if off + bufs[j]["size"] <= po:
                        break
                    off = max(off, po + ps)
                bufs[j]["off"] = off
                placed.append((off, bufs[j]["size"], first_, last_))
                total = max(total, off + bufs[j]["size"])
            if total > CHAIN_LDS_1:
                continue
            # cost: estimated shader cycles of one tile per owned output pixel, from the s_memtime traces of the kernel (tools/
            # trace_chain.sh): ~2.4 k per tile (input store, turn-over), ~0.6 k per stage (descriptor, barrier), a PW item (32 pixels x
            # 32 couts) ~3 k + 0.6 k per 16-deep K slice, a DW item (64 pixels x 8 channels) ~0.7 k + 0.21 k per tap; four waves share
            # the items of a stage; one block per CU instead of two costs its latency hiding
            cyc = CHAIN_TILE_CYC
            for j, st in enumerate(stages):
                rp = regs[j + 1]["Eh"] * regs[j + 1]["Ew"]
                if st["type"] == "pw":
                    items, per = -(-rp // 32) * st["nct"], 3000.0 + 600.0 * st["nks"]
                else:
                    items, per = -(-rp // 64) * (st["cin"] // 8), 700.0 + 210.0 * st["k"] ** 2
                cyc += CHAIN_STAGE_CYC + -(-items // 4) * per
            blocks_cu = min(CHAIN_BLOCKS_CU, (160 * 1024) // (total + 2048))         # 146 VGPRs: three 4-wave blocks per CU at most
            cost = cyc / float(th * tw) * {1: CHAIN_ONE_BLOCK, 2: CHAIN_TWO_BLOCKS}.get(blocks_cu, 1.0)
            if best is None or cost < best["cost"]:
                best = dict(cost=cost, th=th, tw=tw, regs=regs, bufs=[dict(b) for b in bufs], lds_total=total, wbytes=wbytes, dims=dims)
        return best

    def _chain_gouts(self, stages, inside):
        """Which stage outputs some op outside the chain still reads (the last one always is stored)."""
        gouts = []
        for j, st in enumerate(stages):
            ext = [c for c in self._live_consumers(st["out_name"]) if c not in inside]
            need = j == len(stages) - 1 or bool(ext) or st["out_name"] in self.fetched_names or st["out_name"] in self.placement
            st["gout"] = len(gouts) if need else -1
            if need:
                gouts.append(j)
        return gouts

    def try_lower_chain(self, i0):
        """Called for a conv op before the ordinary lowering; True when op i0 (and the ops behind it) became one OP_CHAIN.
        The whole 1x1 / depthwise path that starts at i0 is laid out first, then cut into chains by dynamic programming over
        HBM bytes (chain [i..j]: its input once — times the halo its tile reads —, its last output and every inner tensor
        another op needs; a lone stage: an ordinary conv op's reads and writes) under the LDS budget; the first segment is
        emitted here, the rest of the path is cut again (same optimum) when the lowering reaches it."""
        if not (CHAIN and getattr(self, "chain", False)) or self.ragged:
            return False
        cand = self._chain_candidate(i0)
        if cand is None:
            return False
        op0 = self.ops[i0]
        inv = self.resolve(op0["in"]["Input"][0])
        if (inv is None or inv.tag != "nchw" or inv.parts is not None or inv.up or inv.segs != [(0, inv.c)] or inv.c != cand["cin"]
                or inv.buf.esize != 2 or op0["in"]["Input"][0] in self.pending_gate or op0["in"]["Input"][0] in self.pending_wgate):
            return False
        snapshot = set(self.done)
        path = self._chain_build(i0, 64)
        inside_all = {st["op"] for st in path} | (self.done - snapshot)
        self.done = set(snapshot)
        n = len(path)
        if n < 2:
            return False
        in_lo = bool(getattr(inv.buf, "lo_off", 0))
        # tensor sizes along the path (bytes per image): t[0] = path input, t[j + 1] = output of stage j
        dims = [(inv.h, inv.w)]
        for st in path:
            h, w = dims[-1]
            if st["type"] == "dw" and st["s"] == 2:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            dims.append((h, w))
        t = [dims[0][0] * dims[0][1] * path[0]["cin"] * 2.0] + [dims[j + 1][0] * dims[j + 1][1] * st["cout"] * 2.0 for j, st in enumerate(path)]
        ext = [any(c not in inside_all for c in self._live_consumers(st["out_name"])) or st["out_name"] in self.fetched_names
               or st["out_name"] in self.placement for st in path]
        INF = float("inf")
        nimg = 64          # a NOMINAL batch: the cut must not depend on the batch a frame rides in (its values would)

        def seg_cost(i, j):               # estimated microseconds of stages i..j inclusive (0-based) as ONE op
            if i == j:
                st = path[i]
                byts = t[i] + t[i + 1] + (t[st["res_abs"]] if st["res_abs"] >= 0 else 0.0)
                return nimg * byts / CHAIN_OP_TBS + 6.0
            if j - i + 1 > CHAIN_MAX_STAGES:
                return INF
            sub = []
            for k in range(i, j + 1):
                st = dict(path[k])
                if st["res_abs"] >= 0:
                    if st["res_abs"] < i:
                        return INF
                    st["res_buf"] = st["res_abs"] - i
                sub.append(st)
            # inner outputs some op OUTSIDE THIS SEGMENT reads are stored by the chain (at most two beside its last output): ops outside
            # the whole path (ext) and later stages of the path that read them as a residual from another segment
            if sum(1 for k in range(i, j) if ext[k] or any(path[m]["res_abs"] == k + 1 for m in range(j + 1, n))) > 2:
                return INF
            plan = self._chain_plan(sub, dims[i][0], dims[i][1], in_lo if i == 0 else False)
            if plan is None:
                return INF
            oh_, ow_ = dims[j + 1]
            ntiles = nimg * -(-oh_ // plan["th"]) * -(-ow_ // plan["tw"])
            bcu = min(CHAIN_BLOCKS_CU, (160 * 1024) // (plan["lds_total"] + 2048))
            blocks = 256 * bcu
            # (plan["cost"] carries the occupancy factor of its block count: undo it, the rounds of tiles below account for the blocks)
            us = -(-ntiles // blocks) * plan["cost"] * plan["th"] * plan["tw"] / {1: CHAIN_ONE_BLOCK, 2: CHAIN_TWO_BLOCKS}.get(bcu, 1.0) / 2100.0
            return CHAIN_TIME_WEIGHT * us + 6.0

        best = [0.0] * (n + 1)
        cut = [0] * (n + 1)
        for i in range(n - 1, -1, -1):
            best[i] = INF
            for j in range(i, min(n, i + CHAIN_MAX_STAGES)):
                c = seg_cost(i, j)
                if c + best[j + 1] < best[i]:
                    best[i], cut[i] = c + best[j + 1], j
        L = cut[0] + 1
        if L < 2:
            return False
        stages = self._chain_build(i0, L)
        inside = {st["op"] for st in stages} | (self.done - snapshot)
        gouts = self._chain_gouts(stages, inside) if len(stages) == L else []
        plan = self._chain_plan(stages, inv.h, inv.w, in_lo) if len(stages) == L else None
        if plan is None or len(gouts) > 3:
            # the segment as EMITTED differs from the segment as PRICED (a consumer the cost model counted as internal, a shorter greedy
            # build): leave these convs to the layer-by-layer lowering instead of failing the whole graph
            self.done = set(snapshot)
            return False
        self._chain_emit(stages, plan, inv, gouts)
        return True

    def try_lower_head_tail(self, i0):
        """The DB head's tail: conv2d_transpose(c0 -> c1, 2x2 s2) + BN + relu -> conv2d_transpose(c1 -> 1, 2x2 s2) + sigmoid -> fetch
        as ONE chain of two 1x1 convs: stage A = the first transposed conv as a 1x1 conv to 4 c1 channels ordered (dy, dx, co);
        stage B = the second one applied to each of the four sub-pixels: a block-diagonal 1x1 conv 4 c1 -> 16 whose channel
        4 r + c is output pixel (4 y + r, 4 x + c) of input pixel (y, x); the kernel stores those 4 x 4 blocks of the fp32 map itself
        (CHS_SHUF).  The c1-channel tensor at twice the resolution (24 x 272 x 480 per frame, written and read back: 16 MB of the
        mobile detectors' 200 MB per frame) never exists."""
        # (also in the layer-by-layer program of a hi + lo net, chain=False: the register form chain_pw2_kernel replaces two launches)
        if not (CHAIN and CHAIN_HEAD and (getattr(self, "chain", False) or getattr(self, "hilo", False))) or self.ragged:
            return False
        op0 = self.ops[i0]
        if op0["type"] != "conv2d_transpose":
            return False
        w1 = self.W[op0["in"]["Filter"][0]]
        a0 = op0["attrs"]
        if tuple(w1.shape[2:]) != (2, 2) or list(a0["strides"]) != [2, 2] or any(a0["paddings"]) or w1.shape[0] % 8:
            return False
        inv = self.resolve(op0["in"]["Input"][0])
        if inv is None or inv.parts is not None or inv.up or inv.segs != [(0, inv.c)] or inv.c != w1.shape[0] or inv.buf.esize != 2:
            return False
        snapshot = set(self.done)
        c0, c1 = int(w1.shape[0]), int(w1.shape[1])
        ep1 = self.absorb_epilogue(op0["out"]["Output"][0], i0, c1, allow_res=False)
        cons = self._live_consumers(ep1["out_name"])
        ok = len(cons) == 1 and self.ops[cons[0]]["type"] == "conv2d_transpose" and ep1["out_name"] not in self.placement
        if ok:
            op1 = self.ops[cons[0]]
            w2 = self.W[op1["in"]["Filter"][0]]
            a1 = op1["attrs"]
            ok = (tuple(w2.shape) == (c1, 1, 2, 2) and list(a1["strides"]) == [2, 2] and not any(a1["paddings"])
                  and ep1["post_a"] == 1.0 and ep1["post_b"] == 0.0 and ep1["act2"] == ir.ACT_NONE)
        if ok:
            ep2 = self.absorb_epilogue(op1["out"]["Output"][0], cons[0], 1, allow_res=False)
            fc = self._live_consumers(ep2["out_name"])
            ok = (len(fc) == 1 and self.ops[fc[0]]["type"] == "fetch" and ep2["act2"] == ir.ACT_NONE
                  and ep2["post_a"] == 1.0 and ep2["post_b"] == 0.0)
        if not ok:
            self.done = snapshot
            return False
        self.done.add(i0)
        self.done.add(cons[0])
        c1p = rup(c1, 8)
        # stage A: W_A[(dy, dx, co), ci] = w1[ci, co, dy, dx] * scale1[co]
        wa = np.zeros((4 * c1p, c0), np.float64)
        ba = np.zeros(4 * c1p, np.float64)
        for dy in range(2):
            for dx in range(2):
                q = (dy * 2 + dx) * c1p
                wa[q:q + c1] = (w1[:, :, dy, dx].astype(np.float64) * ep1["scale"].reshape(1, -1)).T
                ba[q:q + c1] = ep1["shift"]
        # stage B: channel 4 r + c (r = 2 dy + ey, c = 2 dx + ex) reads sub-pixel (dy, dx) through tap (ey, ex) of the second conv
        wb = np.zeros((16, 4 * c1p), np.float64)
        for r in range(4):
            for c in range(4):
                q = ((r >> 1) * 2 + (c >> 1)) * c1p
                wb[4 * r + c, q:q + c1] = w2[:, 0, r & 1, c & 1].astype(np.float64) * float(ep2["scale"][0])
        bb = np.full(16, float(ep2["shift"][0]), np.float64)
        na, nb = op0["in"]["Filter"][0] + ":head_tail_a", op1["in"]["Filter"][0] + ":head_tail_b"
        self.W[na], self.W[nb] = wa.reshape(4 * c1p, c0, 1, 1), wb.reshape(16, 4 * c1p, 1, 1)
        one = lambda n_: np.ones(n_, np.float64)
        epa = dict(ep1, scale=one(4 * c1p), shift=ba, res=None, out_name=ep1["out_name"])
        epb = dict(ep2, scale=one(16), shift=bb, res=None, out_name=ep2["out_name"])
        stages = [dict(type="pw", k=1, s=1, cin=c0, cout=4 * c1p, op=i0, wname=na, ep=epa, in_name=op0["in"]["Input"][0],
                       out_name=ep1["out_name"], res_buf=-1, res_abs=-1, gout=-1),
                  dict(type="pw", k=1, s=1, cin=4 * c1p, cout=16, op=cons[0], wname=nb, ep=epb, in_name=ep1["out_name"],
                       out_name=ep2["out_name"], res_buf=-1, res_abs=-1, gout=0, shuf=1)]
        plan = self._chain_plan(stages, inv.h, inv.w, bool(getattr(inv.buf, "lo_off", 0)))
        if plan is None:
            self.done = snapshot
            return False
        self._chain_emit(stages, plan, inv, [1], map_out=True, macs_override=inv.n * inv.h * inv.w * (c0 * c1 * 4 + 4 * c1 * 4))
        return True

    def _chain_emit(self, stages, plan, inv, gouts, map_out=False, macs_override=None):
        n = len(stages)
        bufs, regs, dims = plan["bufs"], plan["regs"], plan["dims"]
        N = inv.n
        oh, ow = dims[-1]
        tiles_h, tiles_w = -(-oh // plan["th"]), -(-ow // plan["tw"])
        hdr = np.zeros(ir.CH_HDR, np.int32)
        hdr[[ir.CHH_MAGIC, ir.CHH_NSTAGES, ir.CHH_NBUFS, ir.CHH_LDSW_BYTES, ir.CHH_LDS_TOTAL, ir.CHH_TH, ir.CHH_TW, ir.CHH_TILES_H,
             ir.CHH_TILES_W]] = [ir.CH_MAGIC, n, n + 1, plan["wbytes"], plan["lds_total"], plan["th"], plan["tw"], tiles_h, tiles_w]
        bw = np.zeros((n + 1, ir.CH_BUF), np.int32)
        # buffer n = the REGION of the last stage's output (kind 2: it is stored to global memory only; the kernel needs its tile
        # geometry all the same — a depthwise last stage indexes its output pixels by it)
        rl = regs[n]
        bw[n, :15] = [2, 0, -1, stages[-1]["cout"], stages[-1]["cout"], 0, rl["Th"], rl["ah"], rl["Eh"], rl["Tw"], rl["aw"], rl["Ew"],
                      rl["Eh"] * rl["Ew"], dims[n][0], dims[n][1]]
        for j, b in enumerate(bufs):
            r = b["reg"]
            bw[j, :15] = [b["kind"], b["off"], (b["off"] + b["size"] // 2) if b["lo"] else -1, b["C"], b["Cp"], b["stride"],
                          r["Th"], r["ah"], r["Eh"], r["Tw"], r["aw"], r["Ew"], b["P"], b["dims"][0], b["dims"][1]]
        sw = np.zeros((n, ir.CH_STAGE), np.int32)
        lds_img = np.zeros(plan["wbytes"], np.uint8)
        meta = []
        macs = 0.0
        for j, st in enumerate(stages):
            ep = st["ep"]
            w = self.W[st["wname"]].astype(np.float64)
            row = sw[j]
            is_pw = st["type"] == "pw"
            nxt_dw = j + 1 < n and stages[j + 1]["type"] == "dw"
            row[[ir.CHS_TYPE, ir.CHS_IN, ir.CHS_OUT, ir.CHS_RES, ir.CHS_CIN, ir.CHS_COUT, ir.CHS_K, ir.CHS_S, ir.CHS_PAD, ir.CHS_ACT,
                 ir.CHS_GOUT, ir.CHS_MASK, ir.CHS_ACT2]] = [ir.CH_PW if is_pw else ir.CH_DW, j, j + 1, st["res_buf"],
                                                            st["cin"], st["cout"], st["k"], st["s"], st["k"] // 2, ep["act"],
                                                            st["gout"], 1 if nxt_dw else 0, ep["act2"]]
            row[ir.CHS_SHUF] = int(st.get("shuf", 0))
            row[ir.CHS_ACT_A], row[ir.CHS_ACT_B] = f2i(ep["act_a"]), f2i(ep["act_b"])
            row[ir.CHS_POST_A], row[ir.CHS_POST_B] = f2i(ep["post_a"]), f2i(ep["post_b"])
            h_out, w_out = dims[j + 1]
            if is_pw:
                mat = np.zeros((st["nct"] * 32, st["nks"] * 16), np.float64)
                mat[:st["cout"], :st["cin"]] = w[:, :, 0, 0] * ep["scale"].reshape(-1, 1)
                frags = pw_fragments(mat)
                lds_img[st["w_lds"]:st["w_lds"] + frags.nbytes] = frags.reshape(-1).view(np.uint8)
                bias = np.zeros(st["nct"] * 32, np.float32)
                bias[:st["cout"]] = ep["shift"]
                lds_img[st["b_lds"]:st["b_lds"] + bias.nbytes] = bias.view(np.uint8)
                row[[ir.CHS_NKS, ir.CHS_NCT, ir.CHS_WLDS, ir.CHS_BLDS, ir.CHS_HASLO]] = [st["nks"], st["nct"], st["w_lds"], st["b_lds"],
                                                                                           1 if bufs[j]["lo"] else 0]
                macs += N * h_out * w_out * st["cin"] * st["cout"]
                meta.append(dict(type="pw", w=mat[:st["cout"], :st["cin"]].copy(), b=np.asarray(ep["shift"], np.float64).copy()))
            else:
                k2 = st["k"] ** 2
                wk = (w[:, 0] * ep["scale"].reshape(-1, 1, 1)).reshape(st["cin"], k2).astype(np.float32)
                bk = np.asarray(ep["shift"], np.float32)
                rec = np.zeros((st["cin"], dw_rec(st["k"])), np.float32)
                rec[:, :k2] = wk
                rec[:, k2] = bk
                lds_img[st["w_lds"]:st["w_lds"] + rec.nbytes] = rec.reshape(-1).view(np.uint8)
                row[ir.CHS_WLDS] = st["w_lds"]
                macs += N * h_out * w_out * st["cin"] * k2
                meta.append(dict(type="dw", w=wk.astype(np.float64).reshape(st["cin"], st["k"], st["k"]), b=bk.astype(np.float64)))
        words = np.concatenate([hdr, bw.reshape(-1), sw.reshape(-1)])
        img_off = rup(words.nbytes, 16)
        hdr[ir.CHH_LDSIMG_OFF] = img_off
        words = np.concatenate([hdr, bw.reshape(-1), sw.reshape(-1)])
        blob = np.zeros(img_off + lds_img.nbytes, np.uint8)
        blob[:words.nbytes] = words.view(np.uint8)
        blob[img_off:] = lds_img
        # outputs
        outs = []
        for g, j in enumerate(gouts):
            st = stages[j]
            h_out, w_out = dims[j + 1]
            if map_out:
                # CHS_SHUF: the stage's 16 channels are the 4 x 4 output pixels of every input pixel: the 1-channel fp32 map itself
                from .compiler import View
                ob = self.new_buf(N, 4 * h_out, 4 * w_out, 1, esize=4, ext=len(self.outputs) + 1)
                self.outputs.append(dict(name=st["out_name"], kind="map", n=N, h=4 * h_out, w=4 * w_out, c=1, ld=1, esize=4))
                v = View(ob, 0, N, 4 * h_out, 4 * w_out, [(0, 1)], 1)
            else:
                v = self.alloc_out(st["out_name"], N, h_out, w_out, st["cout"], lo=self.wants_lo(st["out_name"]))
            outs.append(v)
            self.env[st["out_name"]] = v
        key = ("chain", tuple(st["wname"] for st in stages), tuple(st["out_name"] for st in stages), plan["th"], plan["tw"], inv.h, inv.w,
               bool(bufs[0]["lo"]))
        w_off = self.add_weights(key, blob)
        ins = [inv, None, outs[2] if len(outs) > 2 else None]
        name = "chain:" + "+".join(st["out_name"] for st in stages)
        self.emit(ir.OP_CHAIN, name[:200], ins, outs[0], p={ir.P_CH_TILES_H: tiles_h, ir.P_CH_TILES_W: tiles_w, ir.P_CH_LDS: plan["lds_total"],
                                                             ir.P_CH_NSTAGES: n, ir.P_CH_NBUFS: n + 1,
                                                             ir.P_CH_PW2: int(bool(map_out and n == 2 and all(st["type"] == "pw" for st in stages)
                                                                                   and stages[0]["cin"] <= 64 and stages[1]["cout"] <= 32)),
                                                             ir.P_CH_IMG: int(lds_img.nbytes),
                                                             ir.P_CH_LO_IN: int(getattr(inv.buf, "lo_off", 0) or 0),
                                                             ir.P_CH_LO_OUT0: outs[0].buf.lo_off,
                                                             ir.P_CH_LO_OUT1: outs[1].buf.lo_off if len(outs) > 1 else 0,
                                                             ir.P_CH_LO_OUT2: outs[2].buf.lo_off if len(outs) > 2 else 0},
                  w_off=w_off, out2=outs[1] if len(outs) > 1 else None)
        self.add_gmacs((macs if macs_override is None else macs_override) / 1e9)
        self.ir_ops[-1]["chain"] = dict(stages=[dict(type=st["type"], k=st["k"], s=st["s"], cin=st["cin"], cout=st["cout"], gout=st["gout"],
                                                     res_buf=st["res_buf"]) for st in stages], th=plan["th"], tw=plan["tw"],
                                        lds=plan["lds_total"], meta=meta)
