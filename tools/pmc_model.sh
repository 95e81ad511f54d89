#!/bin/bash
# PMC counters of the whole-model forward (separate passes, kernel-trace only): tools/pmc_model.sh TAG [B L N d_model]
TAG=${1:-pmcm}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $REPO/tools/model_profile.py 3 "$@" > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 $REPO/tools/model_profile.py 3 "$@" > $OUT/lds.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "lds"):
    f = glob.glob(f"{out}/{sub}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in acc.items():
        if any(t in k for t in ("k_conv_bf", "k_mlp_bf", "k_out_h", "k_embed", "k_head")):
            tot = {c: x for c, x in v.items()}
            print(sub, k, {c: f"{x:.3g}" for c, x in tot.items()})
PY
