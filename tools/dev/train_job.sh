cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/train; rm -rf $O; mkdir -p $O; cd $R
cat > $O/run.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bench import synth_batch
from oatomobile_amd import DIMTrainer, ImitativeModel, transform_visual
dev = torch.device("cuda", 0); B = 128
model = ImitativeModel.synthetic(7, max_batch=1).to(dev)
tr = DIMTrainer(model, lr=1e-3, max_batch=B, device=dev)
lidar, vec, goal = synth_batch(np.random.default_rng(77), B, 2)
batch = dict(visual_features=transform_visual(torch.from_numpy(lidar).to(dev), channels_last=True), velocity=torch.from_numpy(vec[:, :3].copy()).to(dev),
             is_at_traffic_light=torch.from_numpy(vec[:, 3:4].copy()).to(dev), traffic_light_state=torch.from_numpy(vec[:, 4:5].copy()).to(dev),
             player_future=torch.from_numpy(np.cumsum(np.abs(np.random.default_rng(78).normal(size=(B, 4, 2))), axis=1).astype(np.float32)).to(dev))
for _ in range(6): tr.train_step(batch)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python $O/run.py > $O/log.txt 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.2f ms over 6 steps" % (tot / 6e6))
for r in rows[:int(__import__("os").environ.get("TOPN", "28"))]:
  n = r["Name"].replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "")[:70]
  print("%-72s calls/step %6.1f  avg %8.1f us  per step %8.1f us  %5.1f%%" % (n, int(r["Calls"]) / 6, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e3, float(r["Percentage"])))
PY
[ -n "$TIMELINE" ] && python tools/dev/train_timeline.py $O/t "$TIMELINE"
