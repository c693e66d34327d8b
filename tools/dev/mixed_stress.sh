# dev: the small assemble + kNN loop in one process while another process runs whole steps on the same GPU
MODE=eval python tools/dev/fwd_repro_stress.py 3000 1 > /tmp/heavy.log 2>&1 &
HP=$!
sleep 8
python tools/dev/assemble_repro_stress.py 6000 1 16 256 2>&1 | grep "^proc"
python tools/dev/knn_repro_stress.py 6000 1 16 256 3 24 10 2>&1 | grep "^proc"
wait $HP
grep "^proc" /tmp/heavy.log | cut -c1-80
