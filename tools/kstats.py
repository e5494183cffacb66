"""Dev: print the top rows of a rocprofv3 --stats kernel_stats.csv (names truncated).  python tools/kstats.py <dir> [rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    print(name[:84].ljust(84), r["Calls"].rjust(6), "%9.1f us" % (float(r["AverageNs"]) / 1e3), "%6.2f %%" % float(r["Percentage"]))
