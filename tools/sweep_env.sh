run() { python bench.py --steps 300 --warmup 30 --step-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo base $(run) $(run)
for v in 256 512; do echo FWD_BLOCKS=$v $(CLOUDAAE_FC_FWD_BLOCKS=$v run); done
for v in 768 1536; do echo BWD_BLOCKS=$v $(CLOUDAAE_FC_BWD_BLOCKS=$v run); done
for v in 256 4096; do echo BWD_FINE=$v $(CLOUDAAE_FC_BWD_FINE=$v run); done
echo base $(run)
