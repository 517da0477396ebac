"""engine.Context.side_streams: the streaming form's side streams are verified to run beside each other and beside the main stream
(torch's pool streams may share a hardware queue and then execute in order: DESIGN §3.2b); where nothing runs side by side — a profiler that
serialises dispatches — the call degrades to unverified streams instead of failing."""
import warnings

import pytest

pytestmark = pytest.mark.gpu


def test_side_streams_are_concurrent_and_shared(ctx):
    import torch
    main = torch.cuda.current_stream(ctx.tdev)
    ss = ctx.side_streams(2)
    assert len(ss) == 2 and ss[0].cuda_stream != ss[1].cuda_stream
    assert ctx.streams_concurrent(main, ss[0]) and ctx.streams_concurrent(main, ss[1]) and ctx.streams_concurrent(ss[0], ss[1])
    assert [s.cuda_stream for s in ctx.side_streams(2)] == [s.cuda_stream for s in ss]           # cached: every pipeline gets the same ones
    hi = ctx.side_streams(2, priority=-1)
    assert len(hi) == 2 and all(ctx.streams_concurrent(main, s) for s in hi)
    assert not ctx.streams_concurrent(ss[0], ss[0])                                               # one stream is in order with itself


def test_side_streams_degrade_when_nothing_runs_concurrently(ctx, monkeypatch):
    from vse_amd import engine
    c2 = engine.Context(0)
    monkeypatch.setattr(c2, "streams_concurrent", lambda a, b, spin_cycles=0: False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ss = c2.side_streams(3)
    assert len(ss) == 3 and c2.side_streams_verified is False and any("verified" in str(x.message) for x in w)
    c2.close()
