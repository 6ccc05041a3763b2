#!/bin/bash
# runs a command and prints what the cgroup's CPU controller did to it: periods, periods in which the job was stopped, time stopped
# (cgroup v2 cpu.stat; the GPU boxes grant 16 CPUs of quota on 256 hardware threads)
s() { awk '/nr_periods|nr_throttled|throttled_usec|usage_usec/ {printf "%s=%s ", $1, $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null; }
a=$(s); "$@"; b=$(s)
python3 - "$a" "$b" <<'PY'
import sys
def p(x): return {k: int(v) for k, v in (t.split("=") for t in x.split())}
a, b = p(sys.argv[1]), p(sys.argv[2])
d = {k: b[k] - a[k] for k in b}
print("[cpu quota %s] periods %d, throttled in %d, stopped %.2f s (summed over the cgroup), CPU used %.1f s" % (open("/sys/fs/cgroup/cpu.max").read().strip(), d.get("nr_periods", 0), d.get("nr_throttled", 0), d.get("throttled_usec", 0) / 1e6, d.get("usage_usec", 0) / 1e6), file=sys.stderr)
PY
