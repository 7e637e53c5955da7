cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16" -s 2>&1 | tail -25
for f in 0 7 17; do for b in 512 128 32 8; do python tools/stage_times.py --obs-batch $b --iters 30 --enc bf16 --fused $f 2>&1 | grep "B=" | sed "s/^/fused=$f /"; done; done
