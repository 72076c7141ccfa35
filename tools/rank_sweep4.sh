# rank frame (1/8 of the 1080p/64spp frame, positional tile map) in flight under grid knobs
export DEAL=0
echo "== default"; timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== RRT_TRACE_GRID=1024"; RRT_TRACE_GRID=1024 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== RRT_BOUNCE_GRID=1024"; RRT_BOUNCE_GRID=1024 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== both 1024"; RRT_TRACE_GRID=1024 RRT_BOUNCE_GRID=1024 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== both 512"; RRT_TRACE_GRID=512 RRT_BOUNCE_GRID=512 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== RRT_CHUNKS_PER_BLOCK=4"; RRT_CHUNKS_PER_BLOCK=4 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
