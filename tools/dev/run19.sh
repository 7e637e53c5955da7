cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests -m gpu -q -x -s -k "mega or g5_params or layerwise_encoder or g6_rip" 2>&1 | tail -6
python tools/stage_times.py --obs-batch 1 --iters 200 --mega 1 2>&1 | grep "B="
python tools/stage_times.py --obs-batch 1 --iters 200 --mega 0 2>&1 | grep "B="
python tools/stage_times.py --obs-batch 2 --iters 200 --mega 1 2>&1 | grep "B="
RIP_MEGA_TICKS=1 python tools/stage_times.py --obs-batch 1 --iters 20 --mega 1 2>&1 | grep -E "mega|B=" > gpurun_out/r3k/ticks3.log
tail -3 gpurun_out/r3k/ticks3.log
python tools/online_probe.py 1000 2>&1 | grep "graph="
