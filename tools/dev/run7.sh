cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
python tools/online_probe.py 1000 2>&1 | grep spin
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "roctx or graph_equals_eager or abi_contract or g6_rip or r11 or replay_cached or packed" 2>&1 | tail -3
