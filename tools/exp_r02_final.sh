#!/bin/bash
# end of round 2: the GPU suite twice, smoke, the three bench lines of the final code
mkdir -p gpurun_out/final
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_$i.log 2>&1; echo "gpu suite $i: $(tail -n 1 gpurun_out/final/gpu_$i.log)"; grep "^FAILED\|Warning" gpurun_out/final/gpu_$i.log | head -5; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 400 python bench.py > gpurun_out/final/c2.json 2> gpurun_out/final/c2.err; timeout 300 python bench.py --workload C3 --no-cpu-baseline > gpurun_out/final/c3.json 2>/dev/null; timeout 300 python bench.py --workload C4 --no-cpu-baseline > gpurun_out/final/c4.json 2>/dev/null
python - <<'PY'
import json
for n in ("c2","c3","c4"):
    d=json.load(open("gpurun_out/final/%s.json"%n)); r=d.get("roofline") or {}
    print(n, round(d["ms_per_step"],2), round(d["value"],1), r.get("kernel"), round(r.get("frac",0),4), r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
PY
