export DEAL=0 NSTREAMS=1,4
echo "== default"; timeout -k 10 120 python3 tools/two_stream.py 8 64 60 || exit 1
echo "== T1024 B1024 RPB4096"; RRT_TRACE_GRID=1024 RRT_BOUNCE_GRID=1024 RRT_BOUNCE_RPB=4096 timeout -k 10 120 python3 tools/two_stream.py 8 64 60 || exit 1
echo "== T1024 B1024 RPB8192"; RRT_TRACE_GRID=1024 RRT_BOUNCE_GRID=1024 RRT_BOUNCE_RPB=8192 timeout -k 10 120 python3 tools/two_stream.py 8 64 60 || exit 1
echo "== T1024 B512 RPB8192"; RRT_TRACE_GRID=1024 RRT_BOUNCE_GRID=512 RRT_BOUNCE_RPB=8192 timeout -k 10 120 python3 tools/two_stream.py 8 64 60 || exit 1
echo "== spp 128"; timeout -k 10 120 python3 tools/two_stream.py 8 128 40 || exit 1
echo "== spp 256"; timeout -k 10 120 python3 tools/two_stream.py 8 256 24 || exit 1
