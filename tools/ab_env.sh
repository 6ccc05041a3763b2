#!/bin/bash
# same-box A/B of an environment switch: bench.py (fresh binning; MODE=c: --record-cache on) with VAR=a and VAR=b, interleaved,
# the median of a dozen timed regions each
#   usage (on the GPU box): bash tools/ab_env.sh VAR a b OUTDIR [reps] [MODE]
R=${GRAFT_REPO_ROOT:-/root/repo}; VAR=$1; O=$R/gpurun_out/${4:-ab_env}; mkdir -p $O; cd $R
A="--no-cpu-baseline --engine-steps 0 --no-side-workloads --steps 40 --warmup 4 --min-seconds 6"
[ "${6:-f}" = c ] && A="$A --record-cache on"
for rep in $(seq 1 ${5:-2}); do for v in $2 $3; do
  env $VAR=$v python bench.py $A > $O/${6:-f}_${v}_$rep.json 2>> $O/err.txt
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/[fc]_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]["scan_stage"]["avg_ms_by_kernel"]; c=j["config"]
    cm=c.get("config_measured"); cp=c.get("cached_pass")
    print(f.split("/")[-1], round(j["ms_per_step"],3), [round(x,3) for x in j["ms_per_step_minmax"]], j["regions"], [round(x,2) for x in r], "config", round(cm["ms"],2) if isinstance(cm,dict) and "ms" in cm else None, "cached", round(cp["ms_per_step"],2) if isinstance(cp,dict) and "ms_per_step" in cp else None)
PY
