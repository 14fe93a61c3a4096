"""Time single 3x3x3 conv layers through the per-operator C ABI: python tools/time_conv3.py [cin cout S N]...
Prints the average duration of `reps` back-to-back forward calls (weight pack + conv) measured with HIP events; run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_segmentation_project_amd import _lib  # noqa: E402
from multimodal_segmentation_project_amd._lib import call, ptr  # noqa: E402


def run(cin, cout, S, N, reps=30, bwd=False):
    dev = "cuda:0"
    x = (torch.randn(N, S, S, S, cin, device=dev) * 0.5).bfloat16()
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
    b = torch.randn(cout, device=dev)
    y = torch.empty(N, S, S, S, cout, device=dev, dtype=torch.bfloat16)
    wsb = _lib.lib().mi3d_conv3_workspace_bytes(cin, cout, N, S, S, S)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    gy = torch.randn_like(y)
    dx = torch.empty_like(x)
    dW = torch.empty_like(w)
    db = torch.empty_like(b)

    def once():
        if bwd:
            call("mi3d_conv3_backward", 1, 1, ptr(x), cin, cin, ptr(w), ptr(gy), cout, cout, ptr(dx), cin, ptr(dW), ptr(db), 0,
                 N, S, S, S, ptr(ws), wsb, s)
        else:
            call("mi3d_conv3_forward", 1, 1, ptr(x), cin, cin, ptr(w), ptr(b), ptr(y), cout, cout, N, S, S, S, ptr(ws), wsb, s)
    for _ in range(5):
        once()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        once()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / reps * 1e3
    gf = 2 * 27 * cin * cout * N * S ** 3 * (2 if bwd else 1) / 1e9
    print(f"conv3 {'bwd' if bwd else 'fwd'} {cin:3d}->{cout:3d} {S}^3 N={N}: {us:8.1f} us/call  {gf / us * 1e3:7.1f} TFLOP/s (incl. pack)")


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:] if v.lstrip('-').isdigit()]
    bwd = "--bwd" in sys.argv
    cases = [a[i:i + 4] for i in range(0, len(a), 4)] or [[32, 16, 96, 2], [16, 16, 96, 2], [16, 32, 48, 2], [32, 32, 48, 2]]
    for c in cases:
        run(*c, bwd=bwd)
