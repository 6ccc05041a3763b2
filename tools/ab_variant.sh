R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${2:-ab}; mkdir -p $O; cd $R
A="--no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 --steps 32 --warmup 4"
for rep in 1 2 3; do
  for v in $1 main; do
    L=""; [ $v != main ] && L="variants/libgblastn_amd_$v.so"
    GBN_AMD_LIB=$L python bench.py $A --record-cache on > $O/c_${v}_$rep.json 2>> $O/err.txt
    GBN_AMD_LIB=$L python bench.py $A > $O/f_${v}_$rep.json 2>> $O/err.txt
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]["scan_stage"]["avg_ms_by_kernel"]
    print(f.split("/")[-1], round(j["ms_per_step"],3), [round(x,2) for x in r], j["config"].get("config_measured",{}).get("ms") if isinstance(j["config"].get("config_measured"),dict) else None)
PY
