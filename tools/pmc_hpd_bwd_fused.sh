#!/bin/bash
# SQ counters of the fused HPD backward kernels at full size (tools/perf_hpd_bwd_fused.py), two --pmc passes after a kernel-trace pass
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
$ROOT/tools/prof_hpd_bwd_fused.sh | grep -E "hpd_d|fault" || exit 1
if grep -q "Memory access fault" $OUT/prof_hbf.log; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_hbf_a -o p -- python3 $ROOT/tools/perf_hpd_bwd_fused.py > $OUT/pmc_hbf_a.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_hbf_b -o p -- python3 $ROOT/tools/perf_hpd_bwd_fused.py > $OUT/pmc_hbf_b.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_hbf_c -o p -- python3 $ROOT/tools/perf_hpd_bwd_fused.py > $OUT/pmc_hbf_c.log 2>&1 &&
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_hbf_a $OUT/pmc_hbf_b $OUT/pmc_hbf_c > $OUT/pmc_hbf.json
rm -rf $OUT/pmc_hbf_a $OUT/pmc_hbf_b $OUT/pmc_hbf_c
python3 - <<PY
import json
d = json.load(open("$OUT/pmc_hbf.json"))
for k, v in d.items():
    if "hpd_d" in k or "planes_kernel<true, false, 2>" in k or "planes_kernel<false, false, 2>" in k:
        print(k[:48], {a: round(b / 1e6, 1) for a, b in v.items()})
PY
