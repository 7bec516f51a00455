#!/bin/bash
# Same-box A/B of two library builds on all three mechanisms: tools/ab_all.sh libA.so libB.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in gas aer tot; do bash tools/ab_bench.sh $1 $2 $m $([ $m = gas ] && echo 400000 || ([ $m = aer ] && echo 51200 || echo 25600)); done
