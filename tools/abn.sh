#!/bin/bash
# tools/abn.sh name1 name2 ...: bench each build/rt_NAME.so back to back on the same box
for w in "$@"; do echo "== $w"; RT_LIB=$PWD/build/rt_$w.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g inter/s  %.2f ms/step  kernel %.2f ms' % (d['value'], d['ms_per_step'], d['roofline_path']['kernel_ms_per_step']))" || exit 1; done
