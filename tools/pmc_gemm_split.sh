#!/bin/bash
# SQ counters of the three split-bf16 GEMMs of the HashProbDistribution's last layer (tools/perf_gemm_split.py), three --pmc passes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_gs_a -o p -- python3 $ROOT/tools/perf_gemm_split.py > $OUT/pmc_gs_a.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_gs_b -o p -- python3 $ROOT/tools/perf_gemm_split.py > $OUT/pmc_gs_b.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_gs_c -o p -- python3 $ROOT/tools/perf_gemm_split.py > $OUT/pmc_gs_c.log 2>&1 &&
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_gs_a $OUT/pmc_gs_b $OUT/pmc_gs_c > $OUT/pmc_gs.json
rm -rf $OUT/pmc_gs_a $OUT/pmc_gs_b $OUT/pmc_gs_c
python3 - <<PY
import json
d = json.load(open("$OUT/pmc_gs.json"))
for k, v in d.items():
    if "gemm128" in k: print(k[:60], v)
PY
tail -5 $OUT/pmc_gs_a.log
