#!/bin/bash
# env-knob sweeps on the 64-spp bench frame (no rebuild): sample packing cap, chunks per block
for p in 6 5 4 3; do
  echo "== RRT_SAMPLE_PACK=$p"
  RRT_SAMPLE_PACK=$p timeout -k 10 120 bash tools/quick.sh || exit 1
done
for c in 4 8 32; do
  echo "== RRT_CHUNKS_PER_BLOCK=$c"
  RRT_CHUNKS_PER_BLOCK=$c timeout -k 10 120 bash tools/quick.sh || exit 1
done
