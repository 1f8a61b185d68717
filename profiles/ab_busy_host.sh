#!/bin/bash
# A/B of two library builds with the host's cores kept busy by spinning processes (run through gpurun from the repo root):
#   BURNERS=16 VARIANTS="label|ENV=V|streams ..." bash profiles/ab_busy_host.sh outdir
set -o pipefail
R=$PWD
N=${BURNERS:-16}
pids=""
for i in $(seq $N); do
  timeout 280 sh -c 'while :; do :; done' &
  pids="$pids $!"
done
sleep 1
STEPS=${STEPS:-200} bash profiles/ab_r03.sh ${1:-busy}
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
