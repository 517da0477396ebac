#!/usr/bin/env python3
"""The detector head's two kernels (streaming / persistent resident-weight) on 14 random (n, H, W): identical probability maps bit for bit."""
import os, subprocess, sys
sys.path.insert(0, "/root/repo")
SN = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
from vse_amd import engine, modelzoo
torch.manual_seed(1)
ctx = engine.Context(0)
desc, w = modelzoo.get_model("V4_ch_det")
net = engine.Net(ctx, desc, w, fetch_cols=(0,))
rng = np.random.default_rng(5)
for k in range(14):
    n = int(rng.integers(1, 6)); h = 32 * int(rng.integers(2, 12)); wd = 32 * int(rng.integers(2, 14))
    x = (torch.rand((n, h, wd, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
    o = net.run(x)[0]
    print("D", n, h, wd, hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
"""
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
outs = {}
for v in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", SN % root], env=dict(os.environ, VSE_HEAD_RESIDENT=v), capture_output=True, text=True)
    outs[v] = [l for l in r.stdout.splitlines() if l.startswith("D")]
    if r.returncode: print(r.stderr[-800:])
ok = outs["1"] == outs["0"] and len(outs["1"]) == 14
for a, b in zip(outs["1"], outs["0"]): print(a, "==" if a == b else "!=", b.split()[-1])
print("head forms identical on random shapes:", ok)
