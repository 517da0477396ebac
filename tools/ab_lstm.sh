python -m pytest tests/test_gpu_nets.py tests/test_gpu_ragged.py -m gpu -x -q -k "V2" 2>&1 | tail -2
for W in 8 16 8 16; do echo "VSE_LSTM_WAVES=$W"; VSE_LSTM_WAVES=$W python tools/gpu_profile_net.py V2_ch_rec 32 32 768 --ragged --wmin 520 --top 2 2>&1 | grep "ops, total\|kind=12"; done
for W in 8 16; do VSE_LSTM_WAVES=$W python tools/gpu_profile_net.py V2_ch_rec 64 32 1024 --ragged --wmin 520 --top 2 2>&1 | grep "ops, total\|kind=12"; done
