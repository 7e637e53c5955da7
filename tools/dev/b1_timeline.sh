cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b1; rm -rf $O; mkdir -p $O; cd $R
for enc in bf16 fp32; do
rocprofv3 --kernel-trace -d $O/$enc --output-format csv -- python tools/stage_times.py --obs-batch 1 --iters 20 --enc $enc > $O/$enc.log 2>&1
python tools/trace_timeline.py $O/$enc > $O/timeline_$enc.txt 2>&1
done
tail -3 $O/bf16.log
