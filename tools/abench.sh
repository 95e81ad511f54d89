#!/bin/bash
# A/B of environment switches on one box, interleaved rounds in fresh processes:
#   tools/abench.sh OUTTAG ROUNDS "ENV=1" "ENV=2 OTHER=3" ...      ("-" = no switch)
# prints ms/step and the stage times of every run; JSON lines land in gpurun_out/OUTTAG/.
TAG=$1; ROUNDS=$2; shift 2
mkdir -p gpurun_out/$TAG
for r in $(seq 1 $ROUNDS); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    [ "$v" = "-" ] && vv="" || vv="$v"
    env $vv python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-200} ${BENCH_ARGS:-} > gpurun_out/$TAG/v${i}_r$r.json 2> gpurun_out/$TAG/v${i}_r$r.err \
      || { echo "FAILED: $v"; tail -5 gpurun_out/$TAG/v${i}_r$r.err; continue; }
    python3 - "$v" gpurun_out/$TAG/v${i}_r$r.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms"]
print(f"{sys.argv[1]:40s} {d['ms_per_step']:.4f} ms  " + " ".join(f"{k.split('(')[0]}={v*1000:.1f}" for k, v in st.items() if k != "-"), flush=True)
PY
  done
done
