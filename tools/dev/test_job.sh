# GPU box: a subset of the GPU tests.  bash tools/dev/test_job.sh "<-k expression>"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tests
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$1" -s > gpurun_out/tests/log.txt 2>&1; echo "rc=$?" >> gpurun_out/tests/log.txt
grep -v "^$" gpurun_out/tests/log.txt | grep -v "amdgpu.ids" | tail -${TAILN:-40} | cut -c1-300
