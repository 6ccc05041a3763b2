R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-wr}; mkdir -p $O; cd $R
A="--no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 --steps 40 --warmup 4"
for rep in 1 2; do for w in 256 240 224 192; do
  GBN_BIN_WRITERS=$w python bench.py $A > $O/f_${w}_$rep.json 2>> $O/err.txt
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/f_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]["scan_stage"]["avg_ms_by_kernel"]; c=j["config"]
    cm=c.get("config_measured"); cp=c.get("cached_pass")
    print(f.split("/")[-1], round(j["ms_per_step"],3), [round(x,2) for x in r], "config", cm.get("ms") if isinstance(cm,dict) else cm, "cached", cp.get("ms_per_step") if isinstance(cp,dict) else cp)
PY
