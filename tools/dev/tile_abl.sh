cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for abl in 0; do
O=$R/gpurun_out/tileabl$abl; rm -rf $O; mkdir -p $O
RIP_TILE_ABL=$abl rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 3 --enc bf16 --fused 17 > $O/log.txt 2>&1
echo "abl=$abl"; python tools/trace_timeline.py $O/t | grep irb_tile | awk '{print $1,$2,$3,$4,$5, $NF}' | tr '\n' ';'; echo
done
