#!/bin/bash
# all randomised differential testers once, on the GPU box (a few minutes); exit code != 0 if any of them found a mismatch
S=${1:-100}
rc=0
python tools/fuzz_conv.py --cases 500 --seed $S | tail -3 || rc=1
python tools/fuzz_graph.py --gpu --cases 200 --seed $S | tail -3 || rc=1
python tools/fuzz_graph.py --gpu --hilo --cases 100 --seed $((S + 1)) | tail -3 || rc=1
python tools/fuzz_prepost.py --cases 400 --seed $S | tail -3 || rc=1
python tools/fuzz_db.py --cases 400 --seed $S | tail -3 || rc=1
python tools/fuzz_shapes.py --gpu --cases 60 --seed $S | tail -3 || rc=1
python tools/fuzz_ragged.py --cases 60 --seed $S | tail -3 || rc=1
exit $rc
