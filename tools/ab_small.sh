#!/bin/bash
# tools/ab_small.sh name1 name2 ...: the small-run sweep (2^15 .. 2^19 rays per step, persistent kernel) for each build/rt_NAME.so
for w in "$@"; do echo "== $w"; RT_LIB=$PWD/build/rt_$w.so timeout -k 10 200 python tools/sweep_rays.py --lo ${AB_LO:-15} --hi ${AB_HI:-19} 2>&1 >/dev/null | grep "rays:" || exit 1; done
