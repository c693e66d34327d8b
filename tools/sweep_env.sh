# A/B of environment knobs through the step bench: bash tools/sweep_env.sh [bench args]   (edit the list below)
run() { python bench.py --steps 300 --warmup 30 --step-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo base $(run "$@") $(run "$@")
for v in 1 agg edge; do echo SIDE_STREAM=$v $(CLOUDAAE_SIDE_STREAM=$v run "$@"); done
echo base $(run "$@")
