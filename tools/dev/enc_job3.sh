# GPU box: teacher-forced block tests + encoder timeline.  bash tools/dev/enc_job3.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "teacher_forced_vs_bf16_oracle or end_to_end_vs_bf16" -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -v "^$" $O/tests.log | grep "bf16\|passed\|failed\|FAILED\|Error\|rc=" | cut -c1-260 | tail -16
timeout 300 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 8 --enc bf16 > $O/log.txt 2>&1
python tools/trace_timeline.py $O/t > $O/timeline.txt 2>&1
tail -1 $O/log.txt; grep -v "search\|select\|prefix\|fillBuffer" $O/timeline.txt
