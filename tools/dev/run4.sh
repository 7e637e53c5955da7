# round-3 GPU call 4: split kernel variants (workgroup shape, tile pipelining, register tape), phase ticks, cache tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "packed_cache or r11 or teacher_forced or search_candidates_vs_oracle" > gpurun_out/r3d/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3d/tests.log
tail -3 gpurun_out/r3d/tests.log
timeout 120 tools/micro/split_f16 2>&1 | grep -v numerics | tee gpurun_out/r3d/split_f16.log
T="python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel 4"
for w in 4 8; do RIP_SPLIT_WPB=$w $T 2>&1 | grep "B=" | sed "s/^/HEAD wpb=$w /"; done
for fl in "-DRIP_SPLIT_PIPE=0" "-DRIP_REGTAPE=0" "-DRIP_PIPE_VALU=3"; do
  RIP_EXTRA_HIPCC_FLAGS="$fl" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
  RIP_EXTRA_HIPCC_FLAGS="$fl" RIP_SPLIT_WPB=4 $T 2>&1 | grep "B=" | sed "s/^/$fl wpb=4 /"
done
export RIP_EXTRA_HIPCC_FLAGS="-DRIP_PROFILE_TICKS"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
for w in 4 8; do RIP_SPLIT_WPB=$w python tools/stage_times.py --obs-batch 512 --iters 1 --enc bf16 --search-kernel 4 2>&1 | grep "ticks" | sed "s/^/wpb=$w /" | head -8; done
unset RIP_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
