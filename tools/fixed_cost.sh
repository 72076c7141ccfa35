# what a frame costs whatever its size: the 1080p frame at 1 and 4 spp, 4 in flight, under the knobs
export NSTREAMS=1,4
for spp in 1 4; do
echo "== spp $spp default"; timeout -k 10 120 python3 tools/two_stream.py 1 $spp 200 || exit 1
echo "== spp $spp RRT_COVER=0"; RRT_COVER=0 timeout -k 10 120 python3 tools/two_stream.py 1 $spp 200 || exit 1
echo "== spp $spp RRT_BINS=0"; RRT_BINS=0 timeout -k 10 120 python3 tools/two_stream.py 1 $spp 200 || exit 1
echo "== spp $spp RRT_SKY_FORK=0"; RRT_SKY_FORK=0 timeout -k 10 120 python3 tools/two_stream.py 1 $spp 200 || exit 1
echo "== spp $spp RRT_TRACE_GRID=512 RRT_BOUNCE_GRID=512"; RRT_TRACE_GRID=512 RRT_BOUNCE_GRID=512 timeout -k 10 120 python3 tools/two_stream.py 1 $spp 200 || exit 1
done
