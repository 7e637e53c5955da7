cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "g6 or teacher or bench_configuration or config4 or candidates_vs_oracle or model_parallel" 2>&1 | tail -2
for b in 4 8 16 32 64 128 256; do for k in 1 3; do python tools/stage_times.py --obs-batch $b --iters 20 --enc bf16 --search-kernel $k 2>&1 | grep "B=" | sed "s/^/kernel=$k /"; done; done
