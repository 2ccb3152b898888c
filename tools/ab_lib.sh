#!/bin/bash
# A/B of library variants on the GPU box: tools/ab_lib.sh <out file> <command...>
# runs <command> with stark-verifier_amd/lib/libgl355.so (as built) and then with every stark-verifier_amd/lib/variants/libgl355_*.so swapped in, twice, alternating.
out=$1; shift
L=stark-verifier_amd/lib
cp $L/libgl355.so /tmp/libgl355_main.so
for rep in 1 2; do
  echo "== main (rep $rep)" >> $out; cp /tmp/libgl355_main.so $L/libgl355.so; "$@" >> $out 2>&1
  for v in $L/variants/libgl355_*.so; do
    echo "== $(basename $v .so | sed s/libgl355_//) (rep $rep)" >> $out; cp $v $L/libgl355.so; "$@" >> $out 2>&1
  done
done
cp /tmp/libgl355_main.so $L/libgl355.so
