#!/bin/bash
# explicit sample_split on the two mesh workloads against the automatic choice (16 and 8), product library
mkdir -p gpurun_out
L=gpurun_out/r04aw_mesh_split_sweep.log; : > $L
for S in 0 4 8 16 32; do
  echo "== sample_split $S (0 = automatic)" >> $L
  SPLIT=$S REPS=2 timeout -k 10 400 python tests/probes/gpu_r04_mesh_ab.py mesh1m mesh10m 2>&1 | grep -E "ms|rror" >> $L || exit 1
done
cut -c1-100 $L
