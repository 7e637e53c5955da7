# soak: the GPU suite three times on one box (intermittent failures), then the bench twice
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
