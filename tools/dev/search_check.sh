cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "g6 or teacher or bench_configuration or config4 or candidates_vs_oracle or model_parallel" 2>&1 | tail -3
for b in 512 128; do python tools/stage_times.py --obs-batch $b --iters 30 --enc bf16 2>&1 | grep "B="; done
