export DEAL=0 NSTREAMS=1,4
for n in 0 8 16; do echo "== RRT_PROBE_LAUNCHES=$n"; RRT_PROBE_LAUNCHES=$n timeout -k 10 120 python3 tools/two_stream.py 8 64 60 || exit 1; RRT_PROBE_LAUNCHES=$n timeout -k 10 120 python3 tools/two_stream.py 1 1 200 || exit 1; done
