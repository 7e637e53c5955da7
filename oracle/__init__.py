"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's deep-imitative-model
inference path (see oracle/reference_cpu.py).  Nothing under `oatomobile_amd/`
may import this package; only `tests/`, `__graft_entry__.smoke()`, `tools/`
and `bench.py`'s `cpu_baseline` leg do."""
