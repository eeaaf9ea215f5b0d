#!/bin/bash
# Runs "label|command" lines one after the other (each under its own timeout) and stops as soon as a
# step was killed by its timeout; any other exit code (a segfault at exit included) is recorded and the
# next step runs.  usage: tools/run_steps.sh <outdir> <seconds> < tools/steps/<list>.txt   (the lists of round 4 are kept there)
out=$1; lim=$2
while IFS='|' read -r label cmd; do
  [ -z "$label" ] && continue
  echo "== $label: $cmd" | tee -a "$out/steps.log"
  timeout -k 10 "$lim" bash -c "$cmd" > "$out/$label.log" 2>&1
  rc=$?
  echo "   rc=$rc" | tee -a "$out/steps.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   killed by its timeout: stopping" | tee -a "$out/steps.log"; exit 1; fi
done
exit 0
