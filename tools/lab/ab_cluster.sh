#!/bin/bash
# A/B of the XCD-resident cluster (cx_batch.hip: k_ref_cluster) on the C4 reference-order plan (1415 x 1415 grid: 5,659 stages, 18.0 M items):
# ms per replayed call (wall clock, device synchronised).  CX_REF_CLUSTER=0: plain launches (a HIP graph of k_batch / k_batch_run);
# CX_REF_CLUSTER_HELP: 0 every workgroup of the XCD a member, 1 half of them helpers that load the plan's records ahead of the members,
# 2 records and the lines of the items' sources, 3 (default) those and the rules' constants with four records in flight per thread;
# CX_REF_CLUSTER_DRY: 1 records and barriers but no item, 2 the bare barriers.  Last: member 0's clock (CX_REF_CLUSTER_TIME=1), us per stage.
echo "# cluster dry help   ms per call"
for cfg in "0 0 3" "1 0 0" "1 0 1" "1 0 2" "1 0 3" "1 1 0" "1 2 0"; do set -- $cfg
  CX_REF_CLUSTER=$1 CX_REF_CLUSTER_DRY=$2 CX_REF_CLUSTER_HELP=$3 timeout -k 10 200 python tools/bench_configs.py reference:1415:nofp > gpurun_out/ab_$1_$2_$3.json 2>/dev/null && python -c "
import json; r=json.load(open('gpurun_out/ab_$1_$2_$3.json')); print('   $1      $2    $3     %8.2f   (%d launches)' % (r['ms_per_call'], r['plan']['launches']))"
done
CX_REF_CLUSTER_TIME=1 timeout -k 10 200 python tools/bench_configs.py reference:1415:nofp 2>&1 >/dev/null | grep -A1 "^.cluster" | tail -2
