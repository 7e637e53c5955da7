# GPU box: bf16 oracle tests of the encoder + per-kernel timing for the block-kernel variants given as arguments.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "bf16" -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^$" $O/tests.log | tail -${TAILN:-14}
for v in "$@"; do
  if [ $v = old ]; then export RIP_IRB_OLD=1; else unset RIP_IRB_OLD; export RIP_IRB2_VARIANT=$v; fi
  timeout 300 rocprofv3 --kernel-trace -d $O/t_$v --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 8 --enc bf16 > $O/log_$v.txt 2>&1
  python tools/trace_timeline.py $O/t_$v > $O/timeline_$v.txt 2>&1
  echo "== variant $v"; tail -1 $O/log_$v.txt; grep "irb2\|irb_rows\|^sum" $O/timeline_$v.txt
done
