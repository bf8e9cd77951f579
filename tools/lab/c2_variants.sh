#!/bin/bash
# lab: the one-launch chain scan at C2: time per sweep, stamps of one launch
for v in 1 0; do
  echo "== CX_CHAIN_NT=$v"
  CX_CHAIN_NT=$v python3 tools/lab/c2_onepass.py 2>&1 | grep "one launch\|largest" | tail -3
done
CX_CHAIN_ONEPASS_STAMPS=1 python3 tools/lab/c2_onepass.py 2>&1 | grep "onepass" | tail -8
