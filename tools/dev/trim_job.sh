# GPU box, round 6: the forward step without fp32 MFMAs — micro-benchmark numerics (float64 host reference) and cycles,
# then the split / search parity gates and the launch time.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/trim
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Ioatomobile_amd/csrc tools/micro/split_f16.hip -o /tmp/split_f16 2> gpurun_out/trim/micro_build.log && timeout 300 /tmp/split_f16 > gpurun_out/trim/micro.log 2>&1
grep -v "^$" gpurun_out/trim/micro.log | head -30
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "split or search or g6 or teacher or bench_configuration or full_size or config4" > gpurun_out/trim/tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed\|Error\|error\|teacher-forced\|geometric" gpurun_out/trim/tests.log | cut -c1-230 | tail -40
for b in 512 2048; do timeout 300 python tools/stage_times.py --obs-batch $b --iters 10 --enc bf16 2>&1 | tail -1; done
