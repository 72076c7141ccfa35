#!/bin/bash
# rank frame (tile_world 8, 1080p/64spp) in flight: sweep of the primary kernel's chunks per persistent block
for c in 4 8 16 32 64; do
  echo "== RRT_CHUNKS_PER_BLOCK=$c"
  RRT_CHUNKS_PER_BLOCK=$c timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
done
echo "== full frame"
timeout -k 10 120 python3 tools/two_stream.py 1 64 12 || exit 1
