set -u
cd $GRAFT_REPO_ROOT
for mode in fast balanced strict; do
  export MSS_WINO_ACCURACY=$mode
  python tools/attribute_wino_error.py --totals-only --tag _$mode > gpurun_out/attr_$mode.log 2>&1
  python bench.py --steps 5 --no-cpu-baseline --no-ood --no-split > gpurun_out/bench_$mode.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$mode.json")); a=json.load(open("gpurun_out/wino_attribution_$mode.json"))
print("$mode", d["ms_per_step"], d["value"], d["roofline"]["frac"], a["policy"]["vs_reference"], a["policy"]["vs_direct"])
PY
done
export MSS_WINO_ACCURACY=balanced MSS_WINO_F4_MIN_CHANNELS=64
python tools/attribute_wino_error.py --totals-only --tag _balanced_f4c64 > gpurun_out/attr_b64.log 2>&1
python bench.py --steps 5 --no-cpu-baseline --no-ood --no-split > gpurun_out/bench_balanced_f4c64.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/bench_balanced_f4c64.json")); a=json.load(open("gpurun_out/wino_attribution_balanced_f4c64.json"))
print("balanced_f4c64", d["ms_per_step"], d["value"], a["policy"]["vs_reference"], a["policy"]["vs_direct"])
PY
unset MSS_WINO_ACCURACY MSS_WINO_F4_MIN_CHANNELS
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/msda_bwd_stats -- python3 $GRAFT_REPO_ROOT/tools/prof_msda_bwd.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/msda_bwd_stats -name "*kernel_trace.csv" -delete
python tools/kstats.py "gpurun_out/msda_bwd_stats/**/*kernel_stats.csv" 14
