# PMC passes over the filter-gradient kernels on the dominant 3x3 shapes (GPU box):  bash tools/pmc_wgrad.sh [layers...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LAYERS=${@:-"128,128,3,1,28 256,256,3,1,14 512,512,3,1,7"}
python3 $R/tools/bench_conv.py wgrad $LAYERS
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_out -- python3 $R/tools/bench_conv.py wgrad $LAYERS > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'wgrad' in k:
            key = (k.replace('(anonymous namespace)::', '')[:44], r['Grid_Size'], r.get('LDS_Block_Size', ''))
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in agg.items():
    print(key, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
done
