#!/bin/bash
# N fresh processes of tools/debug/slots_stress.py, each under its own timeout; prints one line per run
cd $GRAFT_REPO_ROOT
N=${1:-20}
hung=0
for i in $(seq 1 $N); do
  if timeout -k 5 45 python tools/debug/slots_stress.py > /tmp/ss.log 2>&1; then echo "run $i ok"; else echo "run $i FAILED rc=$? $(tail -1 /tmp/ss.log | cut -c1-120)"; hung=$((hung+1)); fi
done
echo "failures: $hung of $N"
