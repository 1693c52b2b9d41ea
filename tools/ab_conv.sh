#!/bin/bash
# A/B of conv kernel variants built with tools/build_variant.py: tools/ab_conv.sh NAME... (default lib first)
mkdir -p gpurun_out
for v in default "$@"; do
  if [ "$v" = default ]; then unset PPN_LIB; else export PPN_LIB=$PWD/tools/bin/libppn_$v.so; fi
  echo "== $v" 
  timeout -k 10 120 python tools/bench_conv.py "L6 512" "L7 512" "L6.0" "L5 256" "B2 512" "conv3" || exit 1
done
