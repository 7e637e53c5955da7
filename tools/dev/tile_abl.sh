# GPU box, development: ablations of the fused tile kernel (encoder_bf16_tile.hip) — rebuilds the library with
# -DRIP_TILE_ABL=<bits> (WRONG results by construction), prints the irb_tile kernel times, restores the normal build.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for abl in ${ABLS:-0 1 2 3 7 23}; do
O=$R/gpurun_out/tileabl$abl; rm -rf $O; mkdir -p $O
RIP_EXTRA_HIPCC_FLAGS="-DRIP_TILE_ABL=$abl" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 3 --enc bf16 --fused 17 > $O/log.txt 2>&1
echo "abl=$abl"; python tools/trace_timeline.py $O/t | grep irb_tile | awk '{print $1,$2,$3,$4,$5, $NF}' | tr '\n' ';'; echo
done
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
