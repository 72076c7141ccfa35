export DEAL=0
:
export NSTREAMS=4,6,8
for w in 2 4 8 16; do timeout -k 10 120 python3 tools/two_stream.py $w 64 48 || exit 1; done
