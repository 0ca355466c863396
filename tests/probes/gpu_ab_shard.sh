#!/bin/bash
for L in "$@"; do echo "== $L"; RT_HIP_LIB=$PWD/raytracing-rust_amd/$L python tests/probes/gpu_shard_probe.py 2>&1 | grep -E "shards 1 split  1|shards 1 split  4|shards 2 split  [124]|shards 8 split  [48]|shards 8 split 16"; done
