#!/bin/bash
# rank frame (tile_world 8, 1080p/64spp) in flight: sweep of the primary kernel's chunks per persistent block
for d in 0 1; do for c in 2 4 8 16; do
  echo "== DEAL=$d RRT_CHUNKS_PER_BLOCK=$c"
  DEAL=$d RRT_CHUNKS_PER_BLOCK=$c timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
done; done
echo "== full frame"
timeout -k 10 120 python3 tools/two_stream.py 1 64 12 || exit 1
