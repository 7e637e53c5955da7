# GPU box: correctness of the bf16 encoder paths against the bf16 oracle + per-kernel timing of the encoder stage for the
# block-kernel variants.  bash tools/dev/enc_job.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16 or mega or native" -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -25 $O/tests.log
for v in old 0 1; do
  if [ $v = old ]; then export RIP_IRB_OLD=1; else unset RIP_IRB_OLD; export RIP_IRB2_VARIANT=$v; fi
  timeout 300 rocprofv3 --kernel-trace -d $O/t_$v --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 8 --enc bf16 > $O/log_$v.txt 2>&1
  python tools/trace_timeline.py $O/t_$v > $O/timeline_$v.txt 2>&1
  echo "== variant $v"; tail -2 $O/log_$v.txt; grep "irb\|front" $O/timeline_$v.txt | tail -12
done
