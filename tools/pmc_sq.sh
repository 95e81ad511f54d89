#!/bin/bash
# SQ counter passes only (no HBM counters), for A/B of one kernel under environment switches:
#   tools/pmc_sq.sh TAG   (environment switches are inherited)
set -u
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 $ARGS > $OUT/lds.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/table.txt 2>&1
python3 - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/summary.json"))
for k, e in d.items():
    wc = e.get("SQ_WAVE_CYCLES")
    if not wc or "mlp" not in k: continue
    print(k[:60], "us=%.1f" % e["avg_us"], "mfma_util=%.3f" % e.get("mfma_util", 0), "clk=%.2f" % e.get("clock_ghz_profiled", 0),
          "issue=%.3f wait_inst=%.3f wait_any=%.3f" % (e["SQ_ACTIVE_INST_ANY"] / wc, e["SQ_WAIT_INST_ANY"] / wc, e["SQ_WAIT_ANY"] / wc),
          "valu=%.1fM mfma=%.2fM lds=%.2fM wave_cyc=%.1fM valu_active=%.1fM" % (e["SQ_INSTS_VALU"] / 1e6, e["SQ_INSTS_MFMA"] / 1e6, e["SQ_INSTS_LDS"] / 1e6, wc / 1e6, e["SQ_ACTIVE_INST_VALU"] / 1e6))
PY
