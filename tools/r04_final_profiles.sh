#!/bin/bash
# round-4 evidence for profiles/: driver-style bench line, C2 and C3 kernel traces + counters
cd /root/repo
export TMPDIR=/tmp GRAFT_REPO_ROOT=/root/repo
mkdir -p gpurun_out/r04k
timeout 600 python bench.py > gpurun_out/r04k/bench_default.json 2> gpurun_out/r04k/bench_default.err
tail -c 600 gpurun_out/r04k/bench_default.json
timeout 1500 bash tools/profile_round.sh r04k > gpurun_out/r04k/profile_round.log 2>&1
timeout 1500 bash tools/c3_profile.sh r04k > gpurun_out/r04k/c3_profile.log 2>&1
timeout 1500 bash tools/c3_sq.sh r04k > gpurun_out/r04k/c3_sq.log 2>&1
tail -12 gpurun_out/r04k/c3_sq.log
