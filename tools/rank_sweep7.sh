export DEAL=0
for q in 4 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q NSTREAMS=3,4,6,8 timeout -k 10 120 python3 tools/two_stream.py 8 64 48 || exit 1; GPU_MAX_HW_QUEUES=$q NSTREAMS=1,3,4 timeout -k 10 120 python3 tools/two_stream.py 1 64 16 || exit 1; done
echo "== RRT_SKY_FORK=0 q=4";  RRT_SKY_FORK=0 NSTREAMS=3,4,6,8 timeout -k 10 120 python3 tools/two_stream.py 8 64 48 || exit 1
