# A/B of a variant library against the shipped one: bash tools/ab_lib.sh <name>   (rustraytracer_amd/librrt_<name>.so)
N=$1
for lib in rustraytracer_amd/librrt.so rustraytracer_amd/librrt_$N.so; do
  echo "== $lib"
  export RRT_LIB=$PWD/$lib
  timeout -k 10 200 bash tools/quick.sh --in-flight 1 || exit 1
  timeout -k 10 200 bash tools/quick.sh || exit 1
  timeout -k 10 200 bash tools/quick.sh --spp 16 --in-flight 1 || exit 1
done
