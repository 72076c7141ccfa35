#!/bin/bash
# A/B on the GPU box: per-kernel ms of one cfg2 frame for the shipped library ("base") and for each
# variant build named (make -C rustraytracer_amd/csrc variant NAME=x DEFS=...):  bash tools/ab.sh x y -- [bench args]
names=(base)
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${names[@]}"; do
  if [ "$v" = base ]; then unset RRT_LIB; else export RRT_LIB=$PWD/rustraytracer_amd/librrt_$v.so; fi
  echo "== $v"
  timeout -k 10 200 bash tools/quick.sh "$@" || exit 1
done
