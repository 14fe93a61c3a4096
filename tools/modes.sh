for m in "" "--no-graph" "--force-comm --graph-segments" "--force-comm"; do
  python bench.py $m --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MODE graph', d['config']['hipgraph'], 'comm', d['config']['comm_path'], round(d['ms_per_step'],3))"
done
