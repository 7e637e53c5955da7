# GPU box: per-phase cycle counts of the irb2 row loop (development build with -DRIP_IRB2_TICKS).  bash tools/dev/ticks_job.sh [variant]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ticks
export RIP_IRB2_VARIANT=${1:-0}
RIP_EXTRA_HIPCC_FLAGS=-DRIP_IRB2_TICKS python __graft_entry__.py --force > gpurun_out/ticks/build.log 2>&1
python tools/stage_times.py --obs-batch 512 --iters 1 --enc bf16 2>&1 | grep "irb2<" | tail -6
