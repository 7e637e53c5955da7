# GPU box, development: ablations of the fused front kernel (wrong results by construction); restores the build.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for abl in ${ABLS:-0 1 2 3 4 7}; do
O=$R/gpurun_out/frontabl$abl; rm -rf $O; mkdir -p $O
RIP_EXTRA_HIPCC_FLAGS="-DRIP_FRONT_ABL=$abl" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 3 --enc bf16 > $O/log.txt 2>&1
echo "abl=$abl $(python tools/trace_timeline.py $O/t | grep front_bf16 | awk '{print $NF}')"
done
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
