# PMC passes over the YOLOv5 stem kernel (GPU box):  bash tools/pmc_stem6.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VT_BENCH_BATCH=64 VT_BENCH_AFFINE=1
python3 $R/tools/bench_conv.py fwd 8,80,6,2,640 2>&1 | tail -1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_out -- python3 $R/tools/bench_conv.py fwd 8,80,6,2,640 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stem6' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
done
