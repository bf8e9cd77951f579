export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for d in 2 0; do
CX_REF_CLUSTER_DRY=$d CX_REF_CLUSTER_HELP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cluster_$d -- python3 $R/tools/bench_configs.py reference:1415:nofp > $R/gpurun_out/prof_cluster_$d.log 2>&1
f=$(find $R/gpurun_out/prof_cluster_$d -name "*kernel_stats.csv" | head -1)
echo "dry $d"; head -4 $f | cut -c1-200
done
