# GPU box: SQ / LDS counters of the encoder block kernels.  bash tools/dev/pmc_job.sh [RIP_OPT_ENCODER_VARIANT bit mask] [kernel-regex]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O; cd $R
V=${1:-0}; RX=${2:-irb2_bf16|irb_rows}
CMD="python tools/stage_times.py --obs-batch 512 --iters 3 --enc bf16 --variant $V"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "$RX" -d $O/sq --output-format csv -- $CMD > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "$RX" -d $O/lds --output-format csv -- $CMD > $O/lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-include-regex "$RX" -d $O/vm --output-format csv -- $CMD > $O/vm.log 2>&1
python - "$O" <<'PY'
import csv, sys, glob, collections, os
O = sys.argv[1]
for sub in ("sq", "lds", "vm"):
  f = glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True)
  if not f:
    print(sub, "no counter file;", open(os.path.join(O, sub + ".log")).read()[-600:]); continue
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"].replace("void rip::(anonymous namespace)::", "").split("(")[0][:60]
    agg[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
  for (k, c), v in sorted(agg.items()):
    print("%-62s %-26s %14.6g  x%d" % (k, c, sum(v) / len(v), len(v)))
PY
