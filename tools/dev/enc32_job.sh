# GPU box: fp32 encoder timeline at B=512.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timing32; rm -rf $O; mkdir -p $O; cd $R
timeout 300 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch ${BATCH:-512} --iters 6 --enc fp32 > $O/log.txt 2>&1
python tools/trace_timeline.py $O/t > $O/timeline.txt 2>&1
tail -1 $O/log.txt; cat $O/timeline.txt | head -90
