"""Ragged recogniser batches on the HIP engine (through the C ABI): every sample of a mixed-width batch gets
  (a) the oracle's result for a batch of exactly its width (the reference pads a chunk of <= 6 crops of one frame to the
      chunk's widest crop: backend/tools/ocr.py:99, backend/config.py:58; paddleocr grouping restated in
      oracle/pipeline_ref.py rec_batches), within the bounds of tests/parity.py (log-probabilities of every class), and
  (b) BIT-IDENTICAL outputs (arg-max index and max probability of every time step) whatever batch it rides in."""
import numpy as np
import pytest

from oracle import ir_emul, net_ref

pytestmark = pytest.mark.gpu

CASES = [("V4_ch_rec", 48, (320, 333, 481, 500, 777, 322, 640, 1001)), ("V4_en_rec_fast", 48, (320, 401, 555, 339, 896)),
         ("V3_ch_rec_fast", 48, (320, 397, 640)), ("V2_ch_rec", 32, (320, 333, 470, 801, 100))]


def make_batch(widths, h, wmax, seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((len(widths), 3, h, wmax), np.float32)
    for n, wn in enumerate(widths):
        x[n, :, :, :wn] = rng.uniform(-1, 1, (3, h, wn))
    return x.astype(np.float16).astype(np.float32)


def run_ragged(net, x, widths):
    import torch
    xt = torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda()
    outs = [o.cpu().numpy() for o in net.run(xt, widths=np.asarray(widths, np.int32))]
    tl = net.last_tlen.cpu().numpy()
    return outs, tl


@pytest.mark.parametrize("mid,h,widths", CASES)
def test_ragged_batch_matches_oracle_per_sample(ctx, mid, h, widths):
    from parity import check_rec_probs
    from vse_amd import engine
    desc, w = net_ref.get_weights(mid)
    wmax = (max(widths) + 63) // 64 * 64
    x = make_batch(widths, h, wmax, seed=11)
    net = engine.Net(ctx, desc, w, want_probs=True, ragged=True)
    outs, tl = run_ragged(net, x, widths)
    probs, idx = outs[0][:, 0], outs[-1].view(np.int32)[:, 0, :, 0]
    for n, wn in enumerate(widths):
        ref = net_ref.run_graph(desc, w, x[n:n + 1, :, :, :wn])[0].numpy()[0]
        tn = int(tl[n])
        assert ref.shape[0] == tn, (ref.shape, tn)
        check_rec_probs(mid, probs[n, :tn], ref, idx=idx[n, :tn])     # log-probabilities of every class, max probability, arg-max outside near-ties
    # the kernel leg (tests/parity.py TOL_KERNEL): the same ragged program on the CPU emulator with fp16 storage where the kernels store
    # fp16 — the weight rounding that dominates the bound above is on both sides, what is left is the kernels' own arithmetic
    prog = net.program(len(widths), h, wmax)
    emu = ir_emul.Emulator(prog, round_f16=True).run(ir_emul.to_nhwc8(x).astype(np.float16), widths=widths)[0][:, 0]
    worst = {}
    for n, wn in enumerate(widths):
        tn = int(tl[n])
        st = check_rec_probs(mid, probs[n, :tn], emu[n, :tn], idx=idx[n, :tn], leg="stored")
        worst = {k: max(v, worst.get(k, 0.0)) for k, v in st.items()}
    print(f"{mid} ragged {widths} kernel leg stored: " + ", ".join(f"{k} {v:.3g}" for k, v in worst.items()))


@pytest.mark.parametrize("mid,h,widths", CASES)
def test_sample_outputs_do_not_depend_on_the_batch(ctx, mid, h, widths):
    """Each sample alone in a tensor of exactly its width (what the reference's chunk gives it when it is the widest crop)
    vs the same sample inside wide mixed batches, shuffled, with different batch sizes: identical bits."""
    from vse_amd import engine
    desc, w = net_ref.get_weights(mid)
    net = engine.Net(ctx, desc, w, want_probs=False, ragged=True)
    wmax = (max(widths) + 255) // 256 * 256
    x = make_batch(widths, h, wmax, seed=5)
    alone = []
    for n, wn in enumerate(widths):
        outs, tl = run_ragged(net, x[n:n + 1, :, :, :wn], [wn])
        alone.append(outs[-1][0, 0, :int(tl[0])].copy())           # [T_n, 2] raw (idx bits, maxp)
    rng = np.random.default_rng(0)
    for trial in range(3):
        order = rng.permutation(len(widths))
        if trial == 1:
            order = np.concatenate([order, order[:3]])             # another batch size, repeated samples
        if trial == 2:
            order = order[:max(2, len(order) // 2)]
        ws = [widths[i] for i in order]
        wt = wmax if trial != 2 else (max(ws) + 31) // 32 * 32
        outs, tl = run_ragged(net, x[order][:, :, :, :wt], ws)
        for k, i in enumerate(order):
            got = outs[-1][k, 0, :int(tl[k])]
            assert got.shape == alone[i].shape
            assert np.array_equal(got.view(np.int32), alone[i].view(np.int32)), (mid, trial, widths[i])


def test_plain_run_refuses_a_ragged_plan(ctx):
    import ctypes as C
    import torch
    from vse_amd import engine
    desc, w = net_ref.get_weights("V4_en_rec_fast")
    net = engine.Net(ctx, desc, w, want_probs=False, ragged=True)
    x = torch.zeros((1, 48, 320, 8), dtype=torch.float16, device="cuda")
    net.run(x, widths=[320])
    prog, handle = net._ensure((1, 48, 320))
    assert ctx.lib.vse_plan_width_levels(handle) == len(prog.wlevels)
    outs, ptrs = net._ext(prog, x)
    ws = net._workspace((1, 48, 320), prog, 0)
    rc = ctx.lib.vse_plan_run(handle, C.c_void_p(ws.data_ptr()), ptrs, len(ptrs), ctx.stream())
    assert rc < 0 and b"ragged" in ctx.lib.vse_last_error()
