"""Parity tests proper: HIP engine (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import os
import numpy as np
import pytest

from oracle import ir_emul, net_ref

pytestmark = pytest.mark.gpu

CASES = [("V4_ch_det", (2, 3, 96, 160)), ("V4_ch_det", (1, 3, 128, 256)),    # 2nd shape: 9x9 convs on 16-row big-patch tiles
         ("V4_ch_det", (1, 3, 160, 224)),                                        # ragged tile edges in both directions
         ("V4_ch_det", (2, 3, 96, 224)),                                         # head kernel: half-empty last column tile
         ("V4_ch_det_fast", (2, 3, 96, 160)), ("V3_ch_det_fast", (1, 3, 96, 160)),
         ("V2_ch_det", (1, 3, 64, 96)), ("V4_ch_rec", (3, 3, 48, 200)), ("V4_ch_rec_fast", (2, 3, 48, 320)),
         ("V4_ch_rec_fast", (3, 3, 48, 200)),      # maps 100 / 50 px wide: the column-walk depthwise kernel's partial quads, 12 / 6 / 3-row maps
         ("V4_en_rec_fast", (6, 3, 48, 352)), ("V3_ch_rec_fast", (2, 3, 48, 160)), ("V3_latin_rec_fast", (2, 3, 48, 160)),
         ("V2_ch_rec", (2, 3, 32, 128))]


def run_hip(ctx, desc, w, x, want_probs=True, with_net=False):
    import torch
    from vse_amd import engine
    net = engine.Net(ctx, desc, w, want_probs=want_probs)
    xt = torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda()
    outs = [o.cpu().numpy() for o in net.run(xt)]
    return (outs, net) if with_net else outs


@pytest.mark.parametrize("mid,shape", CASES)
def test_net_matches_oracle(ctx, mid, shape):
    desc, w = net_ref.get_weights(mid)
    if mid == "V3_ch_det_fast":
        # real weights: feed a text-like image so the map is not trivially zero
        from vse_amd import synth
        from oracle import pipeline_ref
        fr = synth.make_frames(1, 270, 480, seed=5)[0]
        x, _ = pipeline_ref.det_preprocess(fr)
        shape = x.shape
    else:
        x = np.random.default_rng(0).uniform(-1, 1, shape).astype(np.float32)
    x = x.astype(np.float16).astype(np.float32)
    ref = net_ref.run_graph(desc, w, x)[0].numpy()
    outs, net = run_hip(ctx, desc, w, x, with_net=True)
    if "_det" in mid:
        got = outs[0][..., 0]
        r = ref[:, 0]
        # fp16 activations through ~50-100 layers; tolerance on the sigmoid probability map.  The real-weight
        # MobileNetV3 detector carries activations of several hundred (fp16 ulp 0.25-0.5), so its map is held to a
        # looser max error; what matters downstream is the bitmap (prob > 0.3) and the boxes (test_gpu_pipeline).
        real = mid == "V3_ch_det_fast"
        assert np.abs(got - r).max() < (1e-1 if real else 2e-2), np.abs(got - r).max()
        assert np.abs(got - r).mean() < 2e-3
        clear = np.abs(r - 0.3) > (0.1 if real else 0.02)
        assert np.array_equal((got > 0.3)[clear], (r > 0.3)[clear])
        assert ((got > 0.3) != (r > 0.3)).mean() < 1e-3
    else:
        # north_star: "recogniser logits within 1e-3 fp16" — held in LOG-probabilities of every class, the per-step max probability
        # and the arg-max outside near-ties (tests/parity.py: bounds 2-2.5 x the figures measured on MI355X; the V3 stand-ins, whose
        # fp16 WEIGHTS alone move their ill-conditioned logits, have their own row)
        from parity import check_rec_probs
        probs = outs[0][:, 0]
        idx = outs[-1].view(np.int32)[:, 0, :, 0]
        maxp = outs[-1][:, 0, :, 1]
        stats = check_rec_probs(mid, probs, ref, idx=idx, maxp=maxp)
        print(f"{mid} {shape}: " + ", ".join(f"{k} {v:.3g}" for k, v in stats.items()))
        # the kernel legs: the SAME program (the engine's folded fp16 weights) on the CPU emulator — fp32 activations, then fp16 storage
        # where the kernels store fp16: what the kernels themselves add, without the weight rounding that dominates the bound above
        prog = net.program(x.shape[0], x.shape[2], x.shape[3])
        for leg, rnd in (("fp16w", False), ("stored", True)):
            emu = ir_emul.Emulator(prog, round_f16=rnd).run(ir_emul.to_nhwc8(x).astype(np.float16 if rnd else np.float32))[0][:, 0]
            st = check_rec_probs(mid, probs, emu, idx=idx, maxp=maxp, leg=leg)
            print(f"{mid} {shape} kernel leg {leg}: " + ", ".join(f"{k} {v:.3g}" for k, v in st.items()))


KERNEL_LEG_CONV = 1.2e-3    # of the output range; measured <= 4.7e-4 over the 49 cases (MI355X, round 6): about one fp16 ulp

CONVS = [  # cin, cout, k, stride, pad, h, w, n
    (3, 16, (3, 3), (2, 2), (1, 1), 33, 47, 2), (16, 40, (1, 1), (1, 1), (0, 0), 9, 13, 3),
    (64, 64, (9, 9), (1, 1), (4, 4), 17, 30, 1), (32, 32, (7, 1), (1, 1), (3, 0), 12, 20, 2),
    (32, 32, (1, 7), (1, 1), (0, 3), 12, 20, 2), (128, 160, (3, 3), (1, 1), (1, 1), 20, 24, 1),
    (72, 224, (3, 3), (2, 1), (1, 1), 11, 19, 2), (24, 8, (5, 5), (1, 2), (2, 2), 15, 15, 1),
    (256, 1000, (1, 1), (1, 1), (0, 0), 1, 40, 2), (8, 136, (1, 3), (1, 1), (0, 1), 1, 50, 4),
    # LDS-resident-patch kernel (stride 1, long K, maps that tile into 8x32 / 16x32 patches)
    (64, 64, (9, 9), (1, 1), (4, 4), 32, 64, 2),       # conv_patch_kernel<16,64,true>  (960-pixel patch)
    (64, 64, (9, 9), (1, 1), (4, 4), 40, 70, 1),       # <8,64>: 40 rows pad to 48 (> 20 %) and a ragged right edge
    (128, 160, (3, 3), (1, 1), (1, 1), 16, 64, 2),     # 160 couts -> three 64-cout tiles, one half empty
    (128, 128, (3, 3), (1, 1), (1, 1), 24, 96, 1),     # <8,128>
    (128, 128, (3, 3), (1, 1), (1, 1), 32, 96, 2),     # LIGHT <8,128,2>, map of whole 16-row tiles
    (96, 80, (3, 3), (1, 1), (1, 1), 48, 40, 1),       # LIGHT <8,128,2> with a cout tail (80 of 128) and a column tail (40 of 64)
    (72, 64, (3, 3), (1, 1), (1, 1), 32, 64, 1),       # channel tail (72 = 2*32 + 8), <16,64,false>
    (32, 32, (7, 7), (1, 1), (3, 3), 16, 96, 2),       # 49 taps (odd): last step has a single tap; <16,32,1>
    (64, 24, (5, 5), (1, 1), (2, 2), 32, 64, 1),       # <16,32,1> with a cout tail (24 of 32), two channel chunks
    (96, 40, (7, 1), (1, 1), (3, 0), 16, 64, 1), (96, 40, (1, 7), (1, 1), (0, 3), 16, 64, 1),
    # conv_col_kernel (one filter column per step, 16-channel chunks; cin % 16 == 0, kh in 5/7/9, <= 64 couts) — the 9x9 / 7x7 /
    # 5x5 cases above run on it as well
    (256, 64, (9, 9), (1, 1), (4, 4), 24, 40, 1),      # 16 chunks: patch double buffer + 4-stage ring wrap many times
    (48, 64, (9, 9), (1, 1), (4, 4), 19, 33, 2),       # odd chunk count; second tile row has 3 rows (idle waves), second column 1 px
    (64, 40, (9, 5), (1, 1), (4, 2), 16, 40, 1),       # tall filter, 5 columns; cout tail (40 of 64)
    (32, 64, (5, 9), (1, 1), (2, 4), 21, 64, 1),       # wide filter: 9 steps of 5 taps
    (16, 16, (7, 7), (1, 1), (3, 3), 33, 31, 3),       # a single chunk; 16 couts of a 32-cout tile
    # conv_c3_kernel (3x3, 16-channel chunks, two blocks per CU; tile shape per map) — the 128-cout 3x3 cases above run on it too
    (64, 128, (3, 3), (1, 1), (1, 1), 12, 384, 2),     # recogniser-like map: 4 x 128 tiles, two cout tiles
    (128, 64, (3, 3), (1, 1), (1, 1), 24, 130, 1),     # 8 x 64 tiles with a 2-pixel third column tile (idle waves)
    (48, 64, (3, 3), (1, 1), (1, 1), 19, 70, 2),       # 3 chunks; ragged in both directions
    (256, 96, (3, 3), (1, 1), (1, 1), 34, 60, 1),      # 16 chunks, cout tail (96 = 64 + 32)
    (16, 16, (3, 3), (1, 1), (1, 1), 64, 64, 1),       # K = 144 < PATCH_MIN_K: stays on the implicit GEMM
    # conv_pw_kernel (1x1, <= 64 channels in and out, fragments straight from global memory); (16, 40, 1x1) above runs on it too
    (32, 64, (1, 1), (1, 1), (0, 0), 17, 31, 3),       # M tail inside the last block (1581 pixels)
    (64, 32, (1, 1), (1, 1), (0, 0), 16, 16, 1),       # exactly one block, one cout tile
    (48, 24, (1, 1), (1, 1), (0, 0), 9, 50, 2),        # three K slices, cout tail (24 of 32)
    (64, 64, (1, 1), (1, 1), (0, 0), 33, 40, 2),
    # scalar-addressed implicit GEMM (conv_gemm_kernel: cin % 32 == 0, <= 31 taps); VSE_CONV_GEMM=0 sends the same
    # cases through conv_mfma_kernel
    (64, 128, (3, 3), (1, 1), (1, 1), 20, 36, 2),      # masked, BN=128, image seam inside a tile
    (64, 64, (3, 3), (2, 2), (1, 1), 21, 37, 2),       # stride 2, BN=64, M tail
    (96, 32, (1, 1), (1, 1), (0, 0), 13, 17, 3),       # 1x1 with a zero-padded last K tile (K=96 -> 128): masked, BN=32
    (128, 256, (1, 1), (1, 1), (0, 0), 15, 23, 2),     # unmasked 1x1, two cout tiles
    (160, 72, (5, 5), (1, 1), (2, 2), 10, 12, 1),      # 25 taps, map too small for the patch kernel
    (32, 48, (2, 2), (2, 2), (0, 0), 18, 22, 2),       # even kernel, no padding
    (64, 200, (1, 1), (2, 2), (0, 0), 19, 27, 1),      # strided 1x1, cout tail inside the second tile
    # small 1x1 problems on conv_smallm_kernel (<= 256 input channels of any multiple of 8, <= 4096 wave tiles: the SVTR necks)
    (120, 360, (1, 1), (1, 1), (0, 0), 1, 112, 40),    # 7.5 K slices: the upper half of the last slice is masked; 140 x 12 wave tiles
    (240, 136, (1, 1), (1, 1), (0, 0), 3, 50, 7),      # 15 slices = two rounds of loads; cout tail (136 = 4 x 32 + 8), pixel tail
    (72, 200, (1, 1), (1, 1), (0, 0), 5, 13, 3),       # more than 64 input channels, half slice (72 = 4.5 x 16)
    # conv_stem_kernel (3x3 over <= 4 real channels; the 1x1 in front keeps 3 channels)
    (3, 64, (3, 3), (2, 2), (1, 1), 37, 70, 2),        # stride 2, odd map, row / column tile tails
    (3, 16, (3, 3), (2, 2), (1, 1), 48, 64, 1),        # 16 couts (mobile stems): second cout tile idle
    (3, 40, (3, 3), (1, 1), (1, 1), 19, 45, 2),        # stride 1, cout tail inside the second tile
    (4, 64, (3, 3), (2, 2), (1, 1), 16, 32, 1),        # 4 real channels, exactly one tile
]


@pytest.mark.parametrize("cin,cout,k,s,p,h,w,n", CONVS)
def test_conv_shapes(ctx, cin, cout, k, s, p, h, w, n):
    """Single-conv graphs covering every kernel/stride family of SURVEY App. E, with M/N/K tails."""
    import torch
    rng = np.random.default_rng(cin * 1000 + cout)
    stem = 8                                        # graph input has 3 channels: lift to `cin` with a 1x1 conv first
    desc = {"model": "unit", "ops": [
        {"type": "feed", "in": {"X": ["feed"]}, "out": {"Out": ["x"]}, "attrs": {"col": 0}},
        {"type": "conv2d", "in": {"Input": ["x"], "Filter": ["w0"]}, "out": {"Output": ["t0"]},
         "attrs": {"strides": [1, 1], "paddings": [0, 0], "groups": 1}},
        {"type": "conv2d", "in": {"Input": ["t0"], "Filter": ["w1"]}, "out": {"Output": ["t1"]},
         "attrs": {"strides": list(s), "paddings": list(p), "groups": 1}},
        {"type": "elementwise_add", "in": {"X": ["t1"], "Y": ["b1"]}, "out": {"Out": ["t2"]}, "attrs": {"axis": 1}},
        {"type": "hard_swish", "in": {"X": ["t2"]}, "out": {"Out": ["t3"]}, "attrs": {"offset": 3.0, "scale": 6.0, "threshold": 6.0}},
        {"type": "fetch", "in": {"X": ["t3"]}, "out": {"Out": ["fetch"]}, "attrs": {"col": 0}}],
        "params": {"w0": {"dims": [cin, 3, 1, 1], "dtype": 5}, "w1": {"dims": [cout, cin, k[0], k[1]], "dtype": 5},
                   "b1": {"dims": [cout], "dtype": 5}},
        "var_shapes": {"t0": [-1, cin, -1, -1], "t1": [-1, cout, -1, -1]}}
    wts = {"w0": rng.standard_normal((cin, 3, 1, 1)).astype(np.float32),
           "w1": (rng.standard_normal((cout, cin, k[0], k[1])) / np.sqrt(cin * k[0] * k[1])).astype(np.float32),
           "b1": rng.standard_normal(cout).astype(np.float32) * 0.1}
    x = rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float16).astype(np.float32)
    ref = net_ref.run_graph(desc, wts, x)[0].numpy()
    outs, net = run_hip(ctx, desc, wts, x, with_net=True)
    got = np.transpose(outs[0], (0, 3, 1, 2))
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err < 2e-2 * max(1.0, np.abs(ref).max()), err
    # the kernel's own error, weight rounding excluded (ADVICE r5 / VERDICT r5 #5): the same two-op program on the CPU emulator with
    # the engine's fp16 weights and fp16 storage of the intermediate tensor — what is left is the fp32 summation order of ONE conv and
    # the rare 1-ulp flip of a stored fp16 value: a handful of fp16 ulps of the output, whatever the kernel family
    emu = ir_emul.Emulator(net.program(n, h, w), round_f16=True).run(ir_emul.to_nhwc8(x).astype(np.float16))[0]
    emu = np.transpose(emu, (0, 3, 1, 2))[:, :cout]
    kerr = np.abs(got - emu).max() / max(1.0, np.abs(emu).max())
    print(f"conv {cin}->{cout} k{k} s{s} {h}x{w}: vs oracle {err:.3g}, kernel leg (vs emulator, fp16 storage) {kerr:.3g} of the output range")
    assert kerr < KERNEL_LEG_CONV, kerr


@pytest.mark.parametrize("mid", ["V3_ch_det_fast", "V4_ch_det_fast", "V2_ch_det"])
def test_hilo_weights_on_gpu(ctx, mid):
    """fp16 hi + lo weight pairs (F_HILO: two K passes in conv_gemm_kernel / conv_mfma_kernel, hi + lo tables in the depthwise
    kernels) against the CPU emulator of the same program and against the fp32 interpreter."""
    import torch
    from vse_amd import compiler, engine
    desc, w = net_ref.get_weights(mid)
    if mid == "V3_ch_det_fast":
        from vse_amd import synth
        from oracle import pipeline_ref
        x = np.concatenate([pipeline_ref.det_preprocess(f)[0] for f in synth.make_frames(2, 270, 480, seed=5)])
    else:
        x = np.random.default_rng(0).uniform(-1, 1, (2, 3, 96, 160)).astype(np.float32)
    x = x.astype(np.float16).astype(np.float32)
    ref = net_ref.run_graph(desc, w, x)[0].numpy()[:, 0]
    maps = {}
    for hilo in (False, True):
        net = engine.Net(ctx, desc, w, hilo=hilo)
        maps[hilo] = net.run(torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda())[0].cpu().numpy()[..., 0]
    prog = compiler.compile_model(desc, w, x.shape[0], x.shape[2], x.shape[3], hilo=True)
    emu = ir_emul.Emulator(prog, round_f16=True).run(ir_emul.to_nhwc8(x).astype(np.float16))[0][..., 0]
    # same program with fp16 storage emulated (summation order differs; the real-weight net's head amplifies fp16 ulp flips ~15x)
    real = mid == "V3_ch_det_fast"
    assert np.abs(maps[True] - emu).max() < (3e-2 if real else 2e-3)
    e_lo, e_hi = np.abs(maps[False] - ref).max(), np.abs(maps[True] - ref).max()
    assert e_hi < (5e-2 if real else 1e-2) and e_hi <= e_lo + 1e-4, (e_lo, e_hi)
    if real:        # (stand-in nets put arbitrary pixels next to the threshold: flip counts mean nothing there)
        assert ((maps[True] > 0.3) != (ref > 0.3)).sum() <= ((maps[False] > 0.3) != (ref > 0.3)).sum()


def _up_res_graph(cin, cout, k, rng):
    """conv k x k over a nearest-x2-upsampled tensor (gathered on load: P_INSHIFT) + bias + residual add (F_RES) + relu."""
    pad = [k[0] // 2, k[1] // 2]
    desc = {"model": "unit", "ops": [
        {"type": "feed", "in": {"X": ["feed"]}, "out": {"Out": ["x"]}, "attrs": {"col": 0}},
        {"type": "conv2d", "in": {"Input": ["x"], "Filter": ["wr"]}, "out": {"Output": ["r"]},
         "attrs": {"strides": [1, 1], "paddings": [0, 0], "groups": 1}},
        {"type": "pool2d", "in": {"X": ["x"]}, "out": {"Out": ["p"]},
         "attrs": {"pooling_type": "avg", "ksize": [2, 2], "strides": [2, 2], "paddings": [0, 0], "ceil_mode": False,
                   "exclusive": True, "adaptive": False, "global_pooling": False, "padding_algorithm": "EXPLICIT"}},
        {"type": "conv2d", "in": {"Input": ["p"], "Filter": ["w0"]}, "out": {"Output": ["t0"]},
         "attrs": {"strides": [1, 1], "paddings": [0, 0], "groups": 1}},
        {"type": "nearest_interp_v2", "in": {"X": ["t0"]}, "out": {"Out": ["u"]},
         "attrs": {"scale": [2.0, 2.0], "align_corners": False, "interp_method": "nearest", "out_h": -1, "out_w": -1}},
        {"type": "conv2d", "in": {"Input": ["u"], "Filter": ["w1"]}, "out": {"Output": ["t1"]},
         "attrs": {"strides": [1, 1], "paddings": pad, "groups": 1}},
        {"type": "elementwise_add", "in": {"X": ["t1"], "Y": ["b1"]}, "out": {"Out": ["t1b"]}, "attrs": {"axis": 1}},
        {"type": "elementwise_add", "in": {"X": ["t1b"], "Y": ["r"]}, "out": {"Out": ["t2"]}, "attrs": {"axis": -1}},
        {"type": "relu", "in": {"X": ["t2"]}, "out": {"Out": ["t3"]}, "attrs": {}},
        {"type": "fetch", "in": {"X": ["t3"]}, "out": {"Out": ["fetch"]}, "attrs": {"col": 0}}],
        "params": {"w0": {"dims": [cin, 3, 1, 1], "dtype": 5}, "w1": {"dims": [cout, cin, k[0], k[1]], "dtype": 5},
                   "b1": {"dims": [cout], "dtype": 5}, "wr": {"dims": [cout, 3, 1, 1], "dtype": 5}},
        "var_shapes": {"p": [-1, 3, -1, -1], "t0": [-1, cin, -1, -1], "u": [-1, cin, -1, -1], "t1": [-1, cout, -1, -1],
                       "t1b": [-1, cout, -1, -1], "r": [-1, cout, -1, -1], "t2": [-1, cout, -1, -1]}}
    wts = {"w0": rng.standard_normal((cin, 3, 1, 1)).astype(np.float32),
           "w1": (rng.standard_normal((cout, cin, k[0], k[1])) / np.sqrt(cin * k[0] * k[1])).astype(np.float32),
           "b1": rng.standard_normal(cout).astype(np.float32) * 0.1,
           "wr": rng.standard_normal((cout, 3, 1, 1)).astype(np.float32) * 0.3}
    return desc, wts


@pytest.mark.parametrize("cin,cout,k,h,w", [(64, 64, (3, 3), 32, 64), (128, 96, (3, 3), 24, 128), (64, 64, (9, 9), 32, 64),
                                            (32, 32, (5, 5), 48, 64), (96, 200, (3, 3), 16, 64)])
def test_column_kernels_with_upsampled_input_and_residual(ctx, cin, cout, k, h, w):
    """conv_c3_kernel / conv_col_kernel with the input gathered through a nearest x2 upsample (P_INSHIFT) and a residual add in
    the epilogue (F_RES) — the FPN patterns; the last case (200 couts) takes the implicit GEMM and checks the same graph there."""
    from vse_amd import compiler, ir
    rng = np.random.default_rng(cin + cout)
    desc, wts = _up_res_graph(cin, cout, k, rng)
    prog = compiler.compile_model(desc, wts, 2, h, w)
    big = [o for o in prog.ops if int(o["kind"]) == ir.OP_CONV and int(o["p"][ir.P_KH]) == k[0]][0]
    assert int(big["p"][ir.P_INSHIFT]) == 1 and int(big["flags"]) & ir.F_RES
    assert bool(int(big["flags"]) & ir.F_COL) == (cout <= 192)
    x = rng.uniform(-1, 1, (2, 3, h, w)).astype(np.float16).astype(np.float32)
    ref = net_ref.run_graph(desc, wts, x)[0].numpy()
    got = np.transpose(run_hip(ctx, desc, wts, x)[0], (0, 3, 1, 2))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("cin,cout,k,h,w,n", [(256, 64, (9, 9), 136, 240, 4), (128, 128, (3, 3), 136, 240, 4), (160, 160, (3, 3), 68, 120, 8),
                                              (32, 32, (7, 7), 136, 240, 4)])
def test_column_kernels_race_screen(ctx, cin, cout, k, h, w, n):
    """The column kernels order their LDS-DMA ring with counted vmcnt waits + one barrier per step; a miscounted wait shows up as
    a rare stale fragment that a tolerance test can miss.  Screen: the same launch repeated under a competing stream (timing
    varies from run to run) must give bit-identical outputs every time, at full detector map sizes (many rounds of blocks)."""
    import torch
    from vse_amd import engine
    rng = np.random.default_rng(7)
    desc = {"model": "unit", "ops": [
        {"type": "feed", "in": {"X": ["feed"]}, "out": {"Out": ["x"]}, "attrs": {"col": 0}},
        {"type": "conv2d", "in": {"Input": ["x"], "Filter": ["w0"]}, "out": {"Output": ["t0"]},
         "attrs": {"strides": [1, 1], "paddings": [0, 0], "groups": 1}},
        {"type": "conv2d", "in": {"Input": ["t0"], "Filter": ["w1"]}, "out": {"Output": ["t1"]},
         "attrs": {"strides": [1, 1], "paddings": [k[0] // 2, k[1] // 2], "groups": 1}},
        {"type": "relu", "in": {"X": ["t1"]}, "out": {"Out": ["t2"]}, "attrs": {}},
        {"type": "fetch", "in": {"X": ["t2"]}, "out": {"Out": ["fetch"]}, "attrs": {"col": 0}}],
        "params": {"w0": {"dims": [cin, 3, 1, 1], "dtype": 5}, "w1": {"dims": [cout, cin, k[0], k[1]], "dtype": 5}},
        "var_shapes": {"t0": [-1, cin, -1, -1], "t1": [-1, cout, -1, -1]}}
    wts = {"w0": rng.standard_normal((cin, 3, 1, 1)).astype(np.float32),
           "w1": (rng.standard_normal((cout, cin, k[0], k[1])) / np.sqrt(cin * k[0] * k[1])).astype(np.float32)}
    net = engine.Net(ctx, desc, wts)
    x = (torch.rand((n, h, w, 8), device="cuda") * 2 - 1).half()
    x[..., 3:] = 0
    first = net.run(x)[0].clone()
    side = torch.cuda.Stream()
    junk = torch.rand((4096, 4096), device="cuda")
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                junk = junk @ junk * 1e-4           # competing load on another stream
        out = net.run(x)[0]
        assert torch.equal(out, first), rep
    torch.cuda.synchronize()


def test_conv_fuzz_sample(ctx):
    """A fixed-seed sample of tools/fuzz_conv.py (random channels / filter / stride / map / batch around every kernel family's
    tile, chunk and cout-tile boundaries) against the fp32 oracle; the tool itself was run over 2800 cases with no mismatch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_conv", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_conv.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2026)
    seen = set()
    for _ in range(120):
        fam, c = fz.draw(rng)
        seen.add(str(fam))
        test_conv_shapes(ctx, *c)
    assert {"c3", "col", "pw", "gemm", "gemm1x1", "stem"} <= seen


def test_graph_fuzz_sample(ctx):
    """A fixed-seed sample of tools/fuzz_graph.py on the engine: random graphs over the reference models' operator set match
    the fp32 interpreter or are refused by the compiler (UnsupportedGraph); 600 graphs ran clean with the tool."""
    import importlib.util
    from vse_amd import compiler
    spec = importlib.util.spec_from_file_location("fuzz_graph", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_graph.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(99)
    ok = 0
    for _ in range(60):
        h, w = int(rng.integers(2, 9)) * 16, int(rng.integers(2, 11)) * 16
        n = int(rng.integers(1, 4))
        desc, wts, cout = fz.random_graph(rng, h, w)
        x = rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float16).astype(np.float32)
        ref = net_ref.run_graph(desc, wts, x)[0].numpy()
        try:
            got = run_hip(ctx, desc, wts, x)[0]
        except compiler.UnsupportedGraph:
            continue
        got = np.transpose(got[..., :cout], (0, 3, 1, 2))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())
        ok += 1
    assert ok >= 45


def test_workspace_budget_evicts_least_recently_used(ctx):
    """A long video meets many recogniser shapes, each with its own zero-initialised workspace: beyond the budget the least
    recently used ones are dropped — or, when one of this stream is large enough, re-zeroed and reused in place — and re-created on demand;
    results do not change."""
    import torch
    from vse_amd import engine
    desc, w = net_ref.get_weights("V4_en_rec_fast")
    net = engine.Net(ctx, desc, w, want_probs=True)
    rng = np.random.default_rng(0)
    xs = [torch.from_numpy(ir_emul.to_nhwc8(rng.uniform(-1, 1, (2, 3, 48, wd)).astype(np.float16).astype(np.float32))
                           .astype(np.float16)).cuda() for wd in (96, 160, 224, 96)]
    base = [net.run(x)[0].cpu().numpy() for x in xs]
    sizes = sorted(int(v[0].numel()) for v in net.ws.values())
    assert len(sizes) == 3
    small = engine.Net(ctx, desc, w, want_probs=True)
    small.ws_budget = sizes[-1] + sizes[0] // 2          # room for the largest workspace and a bit: one plan resident at a time
    again = [small.run(x)[0].cpu().numpy() for x in xs]
    assert len(small.ws) == 1 and list(small.ws)[0][0] == (2, 48, 96)
    for a, b in zip(base, again):
        assert np.array_equal(a, b)


_WIDE_SNIPPET = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, {root!r})
from vse_amd import engine, modelzoo
from oracle import ir_emul
ctx = engine.Context(0)
for mid, shape in (("V4_ch_det", (3, 3, 160, 288)), ("V4_ch_rec", (5, 3, 48, 320))):
    desc, w = modelzoo.get_model(mid)
    net = engine.Net(ctx, desc, w, want_probs=False)
    x = np.random.default_rng(3).uniform(-1, 1, shape).astype(np.float32)
    xt = torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda()
    h = hashlib.sha256()
    for o in net.run(xt):
        h.update(np.ascontiguousarray(o.cpu().numpy()).tobytes())
    print("DIGEST", mid, h.hexdigest())
"""


@pytest.mark.skipif(os.environ.get("VSE_DEV_BUILD", "0") != "1", reason="conv_c3w.hip is compiled into development builds only (VSE_DEV_BUILD=1)")
def test_wide_3x3_route_gives_identical_bits():
    """conv_c3w_kernel (the experimental persistent one-block-per-CU 3x3 kernel, VSE_C3_WIDE=1) accumulates in conv_c3_kernel's
    order: whole-network outputs must be identical bit for bit (the launcher reads the switch once per process, hence two
    child processes)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for wide in ("0", "1"):
        env = dict(os.environ, VSE_C3_WIDE=wide)
        r = subprocess.run([sys.executable, "-c", _WIDE_SNIPPET.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[wide] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")]
        assert len(out[wide]) == 2
    assert out["0"] == out["1"]


@pytest.mark.parametrize("shape", [(2, 96, 160), (1, 160, 224), (3, 96, 224), (2, 544, 960)])
def test_head_tail_fusion_gives_identical_bits(ctx, shape):
    """F_TAIL2 (conv_pw_tail_kernel): the server detector's second head deconv (64 -> 1, the base map) inside the launch of the first,
    the map stored densely (ld 1) and read by conv_head_up2r_kernel at pixel stride 1 — every output bit equals the two separate
    conv_pw launches (tail2=False), odd tile edges and the full detector input size included."""
    import torch
    from vse_amd import engine, ir
    desc, w = net_ref.get_weights("V4_ch_det")
    n, h, wd = shape
    x = np.random.default_rng(7).uniform(-1, 1, (n, 3, h, wd)).astype(np.float32)
    xt = torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda()
    outs = {}
    for t2 in (None, False):
        net = engine.Net(ctx, desc, w, tail2=t2)
        fused = [int(r["flags"]) & ir.F_TAIL2 for r in net.program(n, h, wd).ops]
        assert any(fused) == (t2 is None)
        outs[t2] = [o.cpu().numpy() for o in net.run(xt)]
    assert len(outs[None]) == len(outs[False])
    for a, b in zip(outs[None], outs[False]):
        assert a.tobytes() == b.tobytes()
    assert outs[None][0].std() > 0


def test_detector_head_forms_are_identical_and_race_free():
    """The server detector's last conv has two kernels — the streaming one and the persistent resident-weight one (the default):
    same bits on five shapes, and the same bits every time beside unrelated work on another stream (tools/race_screen_det.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "race_screen_det.py"), "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "clean, both forms identical" in r.stdout
