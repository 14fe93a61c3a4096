#!/bin/bash
# A/B two builds of libmi3d.so on the same GPU box: tools/ab.sh <old.so> [rounds]   (new = the in-tree library)
old=$1; n=${2:-3}
for i in $(seq $n); do
  a=$(MI3D_LIB_PATH=$old python bench.py --no-cpu-baseline --no-roofline --steps 40 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python bench.py --no-cpu-baseline --no-roofline --steps 40 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "old $a  new $b"
done
