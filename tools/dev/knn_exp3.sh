# dev: time knn.hip built with different -D flags: bash tools/dev/knn_exp3.sh "-DA=1" "-DA=2" ...
cd cloudaae_amd/csrc
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden $f -c knn.hip -o knn.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libcloudaae_hip.so || exit 1
  echo "== $f"
  (cd ../.. && python tools/bench_knn1.py 32 1024 64 320 10 300 && python tools/bench_knn1.py 128 1024 64 320 10 100)
done
