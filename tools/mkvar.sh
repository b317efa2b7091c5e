#!/bin/bash
# tools/mkvar.sh NAME [git-rev] [extra flags]: build build/rt_NAME.so from the working tree (or from a git revision's csrc)
set -e
name=$1; rev=$2; shift; shift || true
root=$(cd $(dirname $0)/.. && pwd)
src=$root/raytracer_amd/csrc
if [ -n "$rev" ] && [ "$rev" != "WT" ]; then
  tmp=$(mktemp -d); mkdir -p $tmp/raytracer_amd/csrc $tmp/include
  for f in $(git -C $root ls-tree --name-only $rev raytracer_amd/csrc/ include/); do git -C $root show $rev:$f > $tmp/$f; done
  src=$tmp/raytracer_amd/csrc
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -munsafe-fp-atomics -pthread -I$root/include "$@" -o $root/build/rt_$name.so $src/rt_kernels.hip
echo built build/rt_$name.so
