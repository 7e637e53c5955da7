cd $GRAFT_REPO_ROOT
for w in 8 4; do for b in 128 256 384 512 768 1024; do RIP_SPLIT_WPB=$w python tools/stage_times.py --obs-batch $b --iters 10 --enc bf16 --search-kernel 4 2>&1 | grep "B=" | sed "s/^/wpb=$w /"; done; done
