export DEAL=0
for g in 1024 4096; do echo "== RRT_BOUNCE_GRID=$g"; RRT_BOUNCE_GRID=$g timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1; done
echo "== RRT_SKY_FORK=0"; RRT_SKY_FORK=0 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== RRT_SKY_FORK=0 full"; RRT_SKY_FORK=0 timeout -k 10 120 python3 tools/two_stream.py 1 64 12 || exit 1
echo "== RRT_COVER=0"; RRT_COVER=0 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
echo "== RRT_TRACE_GRID=1024"; RRT_TRACE_GRID=1024 timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1
