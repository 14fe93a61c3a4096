ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))"; }
B="python bench.py --no-cpu-baseline --no-roofline --steps 30"
echo "== forced comm path (1-rank RCCL), eager; stand-in collective (duration scaled by bucket bytes) beside the encoder backward; calibrated exchange stream"
for i in 1 2; do
 a=$(timeout -k 10 150 $B --force-comm 2>/dev/null | ms)
 b=$(MI3D_EMULATE_COMM=16:250 timeout -k 10 150 $B --force-comm 2>/dev/null | ms)
 c=$(MI3D_EMULATE_COMM=32:250 timeout -k 10 150 $B --force-comm 2>/dev/null | ms)
 d=$(MI3D_EMULATE_COMM=32:250 MI3D_COMM_CUS=32 timeout -k 10 150 $B --force-comm 2>/dev/null | ms)
 f=$(MI3D_EMULATE_COMM=64:250 timeout -k 10 150 $B --force-comm 2>/dev/null | ms)
 g=$(timeout -k 10 150 $B 2>/dev/null | ms)
 echo "graph-nocomm $g | comm path: none $a  16wg $b  32wg $c  32wg+budget32 $d  64wg $f"
done
echo "== distill 128^3"
for i in 1 2; do a=$(timeout -k 10 150 $B --workload distill --size 128 2>/dev/null | ms); b=$(timeout -k 10 150 $B --workload distill --size 128 --serial-forwards 2>/dev/null | ms); echo "overlap $a  serial $b"; done
python - <<'P'
import torch, multimodal_segmentation_project_amd as mi
from multimodal_segmentation_project_amd.trainer import concurrent_stream
for i in range(3):
    s = concurrent_stream(torch.device("cuda",0))
    print("concurrent_stream ->", s, getattr(s,"mi3d_overlap_us",None), getattr(s,"mi3d_concurrent",None))
P
