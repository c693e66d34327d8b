# dev: time knn.hip built with different -D flags: bash tools/dev/knn_exp3.sh "B N C LD K" "-DA=1" "-DA=2" ...
cd cloudaae_amd/csrc
SHAPE=$1; shift
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden $f -c knn.hip -o knn.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libcloudaae_hip.so || exit 1
  echo "== $f"
  (cd ../.. && python tools/bench_knn1.py $SHAPE 300)
done
