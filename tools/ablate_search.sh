# GPU box: rebuild librip_hip.so with -DRIP_ABL=n and time the search launch (development tool)
for abl in "$@"; do
  RIP_EXTRA_HIPCC_FLAGS="$abl" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i "error" | head -3
  RIP_EXTRA_HIPCC_FLAGS="$abl" python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$abl', 'search ms', round(d['roofline']['ms_per_launch'],3), 'calls/s', round(d['value']))"
done
