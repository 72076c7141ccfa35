for g in 4096 8192 16384 32768 65536; do echo "grid $g"; RRT_TRACE_GRID=$g timeout -k 10 200 bash tools/quick.sh --config cfg5 --steps 3 --warmup 1 --in-flight 1 || exit 1; done
