cd $GRAFT_REPO_ROOT
export RIP_EXTRA_HIPCC_FLAGS="-DRIP_PROFILE_TICKS"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
for w in 4 8; do RIP_SPLIT_WPB=$w python tools/stage_times.py --obs-batch 512 --iters 1 --enc bf16 --search-kernel 4 2>&1 | grep "ticks" | sed "s/^/wpb=$w /" | head -8; done
unset RIP_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
