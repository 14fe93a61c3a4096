"""CPU oracle for the 3D U-Net hot path — TEST INFRASTRUCTURE ONLY.

Two independent restatements of the reference's math (it delegates all arithmetic to PyTorch):
  * c_oracle   : plain C, double accumulation, one function per operator (mi3d_oracle.c)
  * torch_ref  : functional vanilla-torch fp32/fp64 restatement of the whole network / step
Both are pinned to tests/golden/*.npz (outputs of the reference itself, tools/gen_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
