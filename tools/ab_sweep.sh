#!/bin/bash
# tools/ab_sweep.sh "name1 name2 ..." [sweep_rays.py args]: the size sweep for each build/rt_NAME.so, back to back on one box
names=$1; shift
for w in $names; do echo "== $w $*"; RT_LIB=$PWD/build/rt_$w.so timeout -k 10 300 python tools/sweep_rays.py "$@" 2>&1 >/dev/null | grep -v amdgpu.ids || exit 1; done
