#!/bin/bash
# A/B of RRT_SAMPLE_PACK (samples of a pixel sharing a wave) on cfg5 and cfg2
for p in 0 2 4 6; do
  echo "cfg5 pack=$p"; RRT_SAMPLE_PACK=$p timeout -k 10 120 bash tools/quick.sh --config cfg5 --steps 3 --warmup 1 || exit 1
done
for p in 0 1 2 4; do
  echo "cfg2 pack=$p"; RRT_SAMPLE_PACK=$p timeout -k 10 120 bash tools/quick.sh || exit 1
done
