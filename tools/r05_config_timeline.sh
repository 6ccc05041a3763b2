cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05u}; mkdir -p $O/kt; cd $R
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python tools/config_timeline.py > $O/out.txt 2> $O/err.txt
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) 400 > $O/config_timeline_full.txt
rm -rf $O/kt; tail -3 $O/err.txt
