#!/bin/bash
# tools/profile_tree.sh TAG — HBM traffic of ONE exact sweep of the tree schedule on the forests of bench.py's `tree` / `tree-deep` rows:
# FETCH_SIZE and WRITE_SIZE (separate --pmc passes) over the whole process with 4 and with 24 sweeps; (difference) / 20 = one sweep.
# Summary -> gpurun_out/profiles_TAG/TAG_tree_traffic.{md,json} (copy into profiles/).
set -o pipefail
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_tree_$TAG
mkdir -p $O $R/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
for shape in random deep; do
  nf=30000; [ $shape = deep ] && nf=20000      # (bench.py's rows; tools/bench_configs.py tree: 200,000)
  for n in 4 24; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d $O/${shape}_${n}_$c -- python3 $R/tools/tree_sweeps.py $shape $nf $n > $O/${shape}_${n}_$c.log 2>&1 || { tail -5 $O/${shape}_${n}_$c.log; exit 1; }
    done
  done
done
cd $R
python3 tools/summarize_tree.py --tag $TAG --dir $O --out $R/gpurun_out/profiles_$TAG
