#!/usr/bin/env python3
"""Stand-in weights of oracle/net_ref.py (synth_weights + calibrate): are the nets ALIVE (two inputs -> different outputs) and WELL-CONDITIONED
(fp16 rounding of the weights alone -> small output change)?  CPU only.  usage: [VSE_CALIB_SHIFT=.. VSE_CALIB_CENTER=..] python tools/standin_study.py MODEL ..."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import net_ref
for mid in sys.argv[1:]:
    desc,w=net_ref.get_weights(mid)
    det="_det" in mid
    h=32 if mid.startswith("V2_") and not det else 48
    shape=(2,3,64,96) if det else (2,3,h,160)
    rng=np.random.default_rng(0)
    xa=rng.uniform(-1,1,shape).astype(np.float16).astype(np.float32); xb=rng.uniform(-1,1,shape).astype(np.float16).astype(np.float32)
    w16={k:(v.astype(np.float16).astype(np.float32) if v.ndim>=2 else v) for k,v in w.items()}
    pa=net_ref.run_graph(desc,w,xa)[0].numpy().astype(np.float64); pb=net_ref.run_graph(desc,w,xb)[0].numpy().astype(np.float64); p16=net_ref.run_graph(desc,w16,xa)[0].numpy().astype(np.float64)
    if det:
        print(f"{mid:18s} center {net_ref._CALIB_CENTER}: map range {pa.min():.3f}..{pa.max():.3f} frac>0.3 {float((pa>0.3).mean()):.3f}; |a-b| mean {np.abs(pa-pb).mean():.3e}; fp16-weight err max {np.abs(pa-p16).max():.2e}")
    else:
        dl=np.abs(np.log(np.maximum(p16,1e-300))-np.log(np.maximum(pa,1e-300)))
        dab=np.abs(np.log(np.maximum(pb,1e-300))-np.log(np.maximum(pa,1e-300)))
        print(f"{mid:18s} center {net_ref._CALIB_CENTER}: maxp median {np.median(pa.max(-1)):.4f}; distinct argmax {len(set(pa.argmax(-1).ravel().tolist()))}/{pa.shape[0]*pa.shape[1]}; |dlogp| between inputs median {np.median(dab):.2e}; fp16-weight err: max {dl.max():.2e} median {np.median(dl):.2e}")
