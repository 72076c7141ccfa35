export DEAL=0
for r in 2048 1024 4096 8192; do echo "== RRT_BOUNCE_RPB=$r"; RRT_BOUNCE_RPB=$r timeout -k 10 120 python3 tools/two_stream.py 8 64 40 || exit 1; NSTREAMS=1,4 RRT_BOUNCE_RPB=$r timeout -k 10 120 python3 tools/two_stream.py 1 64 16 || exit 1;  NSTREAMS=1,4 RRT_BOUNCE_RPB=$r timeout -k 10 120 python3 tools/two_stream.py 1 16 40 || exit 1; done
