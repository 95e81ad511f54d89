#!/bin/bash
# rocprofv3 kernel stats of the block at small batches (what a rank of a strong-scaling run sees):
#   tools/kstats_batch.sh TAG 32 64 ...
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for b in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/b$b -- python3 bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/$TAG/b$b.json 2> gpurun_out/$TAG/b$b.err
  python3 - $TAG $b <<'PY'
import csv, glob, sys
tag, b = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/{tag}/b{b}/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("B =", b, " sum of kernel averages (one call each per step; conv twice): ")
tot = 0.0
for r in rows[:9]:
    n = 2 if "k_conv" in r["Name"] else 1
    tot += n * float(r["AverageNs"]) / 1e3
    print("  %-62s calls %5s avg %8.1f us  %5s %%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print("  ~ %.1f us of kernels per step" % tot)
PY
done
