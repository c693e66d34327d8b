# dev: whole steps in one process while another process runs large PyTorch matmuls on the same GPU
python - <<'PY' > /tmp/torchheavy.log 2>&1 &
import torch, time
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
t = time.time(); n = 0
while time.time() - t < 28:
    for _ in range(20): c = a @ b
    torch.cuda.synchronize(); n += 20
print("matmuls", n)
PY
HP=$!
sleep 6
MODE=eval python tools/dev/fwd_repro_stress.py 1500 1 2>&1 | grep "^proc" | cut -c1-200
wait $HP
cat /tmp/torchheavy.log | tail -1
