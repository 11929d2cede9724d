cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
D=gpurun_out/r02p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace_aql -- python3 bench.py --steps 300 --warmup 50 --no-extras > $D/trace_aql.log 2>&1; echo "rc=$?"
F=$(find $D/trace_aql -name "*kernel_stats.csv" | head -1); echo "stats: $F"; [ -n "$F" ] && head -8 "$F"
tail -3 $D/trace_aql.log
