# dev: the small-LDS kNN kernel (53 KB) in one process next to a process looping a 150 KB-LDS kernel (knn64_wide)
DEPTH=32 python tools/dev/knn_repro_stress.py 60000 1 32 1024 64 64 10 > /tmp/big.log 2>&1 &
HP=$!
sleep 6
DEPTH=32 python tools/dev/knn_repro_stress.py 40000 1 16 256 3 24 10 2>&1 | grep "^proc"
DEPTH=32 python tools/dev/knn_repro_stress.py 20000 1 16 256 64 64 10 2>&1 | grep "^proc"
wait $HP
grep "^proc" /tmp/big.log
