# round-3 GPU call: the one-launch encoder (encoder_mega_kernel): parity, then stage times at 1 / 2 / 4 observations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests -m gpu -q -x -s -k "mega or g5_params or layerwise_encoder" 2>&1 | tail -15
for wgs in 16 32 64; do
  echo "== RIP_MEGA_WGS=$wgs"
  RIP_MEGA_WGS=$wgs python tools/stage_times.py --obs-batch 1 --iters 200 --mega 1 2>&1 | grep "B="
done
python tools/stage_times.py --obs-batch 1 --iters 200 --mega 0 2>&1 | grep "B="
for b in 2 4; do
  python tools/stage_times.py --obs-batch $b --iters 200 --mega 1 2>&1 | grep "B="
  python tools/stage_times.py --obs-batch $b --iters 200 --mega 0 2>&1 | grep "B="
done
RIP_MEGA_TICKS=1 python tools/stage_times.py --obs-batch 1 --iters 20 --mega 1 2>&1 | grep -E "mega|B=" > gpurun_out/r3k/ticks.log
tail -5 gpurun_out/r3k/ticks.log
python tools/online_probe.py 1000 2>&1 | grep "graph="
