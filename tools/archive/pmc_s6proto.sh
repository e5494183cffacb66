# SQ / LDS counters of span6_kernel on 128 -> 128 3x3 @28x28 (batch 256), shipped kernel against the consumer-side-normalise
# prototype (tools/diag/libvt_s6proto.so: -DVT_SPAN6_PROTO_NORM).  GPU box:  bash tools/pmc_s6proto.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$R/vision-toolbox_amd/csrc/libvt_amd.so" "$R/tools/diag/libvt_s6proto.so"; do
  echo "== $lib"
  for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_out
    VT_AMD_LIB=$lib rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_out -- python3 $R/tools/bench_conv.py fwd 128,128,3,1,28 > /dev/null 2>&1
    python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
dur = []
for f in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'span6_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('/tmp/pmc_out/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'span6_kernel' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print({k: round(sum(v) / len(v)) for k, v in agg.items()}, 'avg us', round(sum(dur[5:]) / max(len(dur[5:]), 1), 1))
PY
  done
done
