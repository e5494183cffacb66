cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/bench_pw.py bwd 64 64 3211264
python3 $R/tools/bench_pw.py bwd 64 64 802816
python3 $R/tools/bench_pw.py bwd 32 32 3211264
python3 $R/tools/bench_pw.py reduce 64 64 3211264
python3 $R/tools/bench_pw.py apply 64 64 3211264
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_out -- python3 $R/tools/bench_pw.py bwd 64 64 3211264 5 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pw_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
done
