cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
RIP_MEGA_TICKS=1 python tools/stage_times.py --obs-batch 1 --iters 50 --mega 1 2>&1 | grep -E "mega|B=" > gpurun_out/r3k/ticks4.log
tail -2 gpurun_out/r3k/ticks4.log
