#!/bin/bash
# A/B of the cluster's helpers on the C4 reference-order plan (tools/bench_configs.py reference:1415:nofp): ms per replayed call.
# columns: CX_REF_CLUSTER_HELP (2: a record at a time; 3: four records in flight, rule constants too), CX_REF_CLUSTER_MEMBERS (0 = half),
# CX_REF_CLUSTER_AHEAD (0 = default)
echo "# help members ahead   ms per call"
for cfg in "$@"; do set -- $cfg
  CX_REF_CLUSTER_HELP=$1 CX_REF_CLUSTER_MEMBERS=$2 CX_REF_CLUSTER_AHEAD=$3 timeout -k 10 200 python tools/bench_configs.py reference:1415:nofp > gpurun_out/ab2_$1_$2_$3.json 2>gpurun_out/ab2_$1_$2_$3.err && python -c "
import json; r=json.load(open('gpurun_out/ab2_$1_$2_$3.json')); print('   $1     %3d     %3d   %8.2f   parity %s' % ($2, $3, r['ms_per_call'], r.get('parity')))" || exit 1
done
