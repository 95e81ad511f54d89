#!/bin/bash
# Per-kernel time of the whole TimesNet.forward mirror (eager launches + graph replays, 10 + 2 + 10 + 2 forwards):
#   tools/model_kernels.sh TAG [B L N d_model]
TAG=${1:-model}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$TAG -- python3 $REPO/tools/model_profile.py 10 "$@" > $REPO/gpurun_out/$TAG/out.txt 2>&1
tail -1 $REPO/gpurun_out/$TAG/out.txt
python3 - $REPO/gpurun_out/$TAG <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
nfwd = 26.0     # lazy build (1, B = 2) + eager 2 + 10 + capture ~2 + graph 2 + 10
for r in rows[:24]:
    print("  %-70s calls %5s avg %8.1f us total/fwd %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / nfwd))
PY
