#!/bin/bash
# tools/gpu_steps.sh: helper for gpurun calls.  `step NAME SECONDS cmd...` runs one GPU step under its own time limit with
# its output in gpurun_out/NAME.log.  A step that ends with exit status 1 (a failed assertion in pytest) lets the next one run;
# any other failure (time limit, abort, fault) ends the call there: nothing more is started on the GPU.
mkdir -p gpurun_out
step() {
  name=$1; lim=$2; shift 2
  echo "== $name: $*"
  timeout -k 10 $lim "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "== $name rc=$rc"; tail -n ${TAILN:-4} gpurun_out/$name.log
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name"; exit $rc; fi
}
