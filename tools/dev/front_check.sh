cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16 or fused or encoder" -s 2>&1 | grep -v "^$" | tail -14
bash tools/dev/tile_prof.sh 512 | grep -v "pw_bf16\|dw_bf16" | tail -26
for b in 512 256; do python tools/stage_times.py --obs-batch $b --iters 30 --enc bf16 2>&1 | grep "B="; done
