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


def test_roles_never_share_stream_objects_even_at_equal_priority(ctx):
    """with one cached list per PRIORITY, a recogniser at the detector's priority (bench.py --det-priority -1, or
    rec_stream_priority = 0) received the detector's own stream objects and the det / rec overlap serialised silently."""
    from vse_amd import engine
    c2 = engine.Context(0)
    det = c2.side_streams(2, priority=0, role="det")
    rec = c2.side_streams(2, priority=0, role="rec")
    handles = [s.cuda_stream for s in det + rec]
    assert len(set(handles)) == 4, handles
    assert [s.cuda_stream for s in c2.side_streams(2, priority=0, role="det")] == handles[:2]       # cached per (priority, role)
    assert all(c2.streams_concurrent(a, b) for a in det for b in det if a is not b)
    assert all(c2.streams_concurrent(a, b) for a in rec for b in rec if a is not b)
    print("cross-verified against the other role:", getattr(c2, "side_streams_cross_verified", True),
          [c2.streams_concurrent(a, b) for a in det for b in rec])
    c2.close()


def test_side_streams_degrade_when_nothing_runs_concurrently(ctx, monkeypatch):
    from vse_amd import engine
    c2 = engine.Context(0)
    monkeypatch.setattr(c2, "streams_concurrent", lambda a, b, spin_cycles=0: False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ss = c2.side_streams(3)
    assert len(ss) == 3 and c2.side_streams_verified is False and any("verified" in str(x.message) for x in w)
    c2.close()
