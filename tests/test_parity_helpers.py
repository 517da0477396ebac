"""The string criterion of tests/parity.py (reachable(): is a string producible from the oracle's per-step distribution through
near-ties only?) on hand-made distributions — CPU, no engine."""
import numpy as np

from parity import check_rec_probs, reachable

CS = ["blank", "a", "b", "c", " ", " "]


def _probs(rows):
    p = np.asarray(rows, np.float64)
    return p / p.sum(1, keepdims=True)


def test_reachable_follows_ctc_collapse_and_near_ties():
    clear = _probs([[.9, .02, .02, .02, .02, .02], [.02, .9, .02, .02, .02, .02], [.02, .9, .02, .02, .02, .02], [.9, .02, .02, .02, .02, .02],
                    [.02, .9, .02, .02, .02, .02], [.02, .02, .02, .9, .02, .02]])
    assert reachable("aac", clear, CS, 2e-2)                 # a a(repeat) blank a c
    assert not reachable("ac", clear, CS, 2e-2) and not reachable("aab", clear, CS, 2e-2) and not reachable("aacc", clear, CS, 2e-2)
    # a near-tie between 'b' and 'c' at the last step (log-margin 1e-2): both strings are reachable, a third one is not
    tie = clear.copy()
    tie[5] = _probs([[.02, .02, .45 * np.exp(-1e-2), .45, .02, .02]])[0]
    assert reachable("aac", tie, CS, 2e-2) and reachable("aab", tie, CS, 2e-2) and not reachable("aaa", tie, CS, 2e-2)
    assert not reachable("aab", tie, CS, 5e-3)                # the same margin is NOT a tie under a tighter bound
    # a tie with the blank removes a character; a tie with the previous class merges a repeat
    tb = clear.copy()
    tb[4] = _probs([[.45, .45 * np.exp(-5e-3), .02, .02, .02, .02]])[0]
    assert reachable("aac", tb, CS, 2e-2) and reachable("ac", tb, CS, 2e-2)
    # two classes with the same character (the two ' ' entries of the en table) collapse by CLASS id, not by character
    sp = _probs([[.02, .02, .02, .02, .9, .02], [.02, .02, .02, .02, .02, .9]])
    assert reachable("  ", sp, CS, 2e-2) and not reachable(" ", sp, CS, 2e-2)


def test_check_rec_probs_bounds_have_teeth():
    rng = np.random.default_rng(0)
    z = rng.standard_normal((2, 40, 6625))
    ref = np.exp(z) / np.exp(z).sum(-1, keepdims=True)
    ok = np.exp(z + 1e-2 * rng.standard_normal(z.shape))             # the engine's measured level: median |delta log p| ~1e-2
    ok /= ok.sum(-1, keepdims=True)
    st = check_rec_probs("V4_ch_rec", ok, ref)
    assert st["dlog_max"] < 0.1 and st["dlog_median"] < 1.5e-2
    bad = ref.copy()                                                 # every probability moved by 10 % (up for even classes, down for odd
    bad[..., 0::2] *= 1.1                                            # ones): an absolute 1e-3 bound on a flat softmax (max-p ~0.004) passes
    bad[..., 1::2] *= 0.9                                            # this; the median log bound does not
    try:
        check_rec_probs("V4_ch_rec", bad, ref)
    except AssertionError:
        pass
    else:
        raise AssertionError("a 10 % error passed")
    try:
        check_rec_probs("V2_ch_rec", ok, ref)                        # the BiLSTM CRNN is held to its own (40 x tighter) figures
    except AssertionError:
        pass
    else:
        raise AssertionError("the V2 bounds accepted the SVTR models' error level")
    idx = ref.argmax(-1).copy()
    srt = np.sort(ref, -1)
    t = np.unravel_index(np.argmax(srt[..., -1] / srt[..., -2]), idx.shape)      # the clearest step: flip its arg-max
    idx[t] = (idx[t] + 1) % 6625
    try:
        check_rec_probs("V4_ch_rec", ref, ref, idx=idx)
    except AssertionError:
        return
    raise AssertionError("an arg-max flip at a clear step passed")
