#!/bin/bash
# gpurun_out/prof_<tag>_* (tools/profile_round.sh) -> profiles/<tag>_*  (summaries only; run in the build container after the gpurun call)
set -eu
TAG=${1:-r05}
O=gpurun_out
for pair in frozen:gngf_frozen hash:hash learning:gngf_learning cfg4:cfg4_hash cfg5:cfg5_hash_fp16; do
  d=${pair%%:*}; m=${pair##*:}
  python tools/prof_summary.py $O/prof_${TAG}_$d 30 > profiles/${TAG}_kernel_stats_$m.txt
  cp $(ls $O/prof_${TAG}_$d/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats_$m.csv
done
python tools/pmc_summary.py $O/prof_${TAG}_pmc_fetch $O/prof_${TAG}_pmc_write > profiles/${TAG}_pmc_fetch_write.json
python tools/pmc_summary.py $O/prof_${TAG}_pmc_fetch_cfg4 $O/prof_${TAG}_pmc_write_cfg4 > profiles/${TAG}_pmc_fetch_write_cfg4.json
if [ -d $O/prof_${TAG}_pmc_fetch_cfg5 ]; then python tools/pmc_summary.py $O/prof_${TAG}_pmc_fetch_cfg5 $O/prof_${TAG}_pmc_write_cfg5 > profiles/${TAG}_pmc_fetch_write_cfg5.json; fi
python tools/pmc_summary.py $O/prof_${TAG}_pmc_sq1 $O/prof_${TAG}_pmc_sq2 > profiles/${TAG}_pmc_sq_counters.json
GNGF_SQ_COUNTERS=profiles/${TAG}_pmc_sq_counters.json python tools/make_traffic.py $O/prof_${TAG}_pmc_fetch $O/prof_${TAG}_pmc_write > /dev/null
ls -la profiles | head -30
