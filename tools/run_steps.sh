#!/bin/bash
# Runs "label|command" lines one after the other (each under its own timeout) and stops as soon as a step was
# killed by its timeout, or ended with a signal that is not the ONE exit-time fault this image is known for.
#
# The known one (profiles/r04_b_coop_exit_sigsegv.txt): under rocprofv3, a process that has made a cooperative
# launch segfaults inside exit(), after the profiler has written its files, in hsa_shut_down of
# libhsa-runtime64.so.1.18.70200 — always with the same nine return addresses (low 12 bits 59e e63 31d d67 01d
# cee c5a fce 097, in that order).  rc 139 is waved through ONLY when the step's log carries exactly that
# sequence; any other segfault (a different stack, or no stack printed) stops the list: it would otherwise hide
# behind the known one.  usage: tools/run_steps.sh <outdir> <seconds> < tools/steps/<list>.txt
out=$1; lim=$2
known_exit_fault() {  # $1 = log: the stack trace's frames carry the nine suffixes in order
  python3 - "$1" <<'PY'
import re, sys
frames = [m.group(1)[-3:] for m in re.finditer(r"^\s*@\s+0x([0-9a-f]+)", open(sys.argv[1], errors="replace").read(), re.M)]
want = ["59e", "e63", "31d", "d67", "01d", "cee", "c5a", "fce", "097"]
ok = any(frames[i:i + len(want)] == want for i in range(len(frames)))
sys.exit(0 if ok else 1)
PY
}
while IFS='|' read -r label cmd; do
  [ -z "$label" ] && continue
  echo "== $label: $cmd" | tee -a "$out/steps.log"
  timeout -k 10 "$lim" bash -c "$cmd" > "$out/$label.log" 2>&1
  rc=$?
  echo "   rc=$rc" | tee -a "$out/steps.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   killed by its timeout: stopping" | tee -a "$out/steps.log"; exit 1; fi
  if [ $rc -eq 139 ] || [ $rc -eq 134 ] || [ $rc -eq 135 ] || [ $rc -eq 136 ]; then
    if [ $rc -eq 139 ] && known_exit_fault "$out/$label.log"; then
      echo "   rc 139 with the known exit-time signature (hsa_shut_down after a cooperative launch under rocprofv3): accepted" | tee -a "$out/steps.log"
    else
      echo "   died on a signal WITHOUT the known exit-time signature: stopping (read $out/$label.log)" | tee -a "$out/steps.log"; exit 1
    fi
  fi
done
exit 0
