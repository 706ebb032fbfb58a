"""The two weight-stationary kernels of round 5 and their tiled counterparts, four launches each, for a rocprofv3 --pmc pass
(tools/ws_pmc.sh): GEGLU 460800 x 2560 x 320 and the LayerNorm-folded 460800 x 960 x 320."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_geglu
M, C = 460800, 320
x = torch.randn(M, C, device="cuda").half()
wp, bp = pack_geglu(torch.randn(2560, C, device="cuda") / math.sqrt(C), torch.randn(2560, device="cuda"))
wp, bp = wp.half(), bp.float().contiguous()
w = (torch.randn(960, C, device="cuda") / math.sqrt(C)).half()
bias, colsum = torch.randn(960, device="cuda"), 0.01 * torch.randn(960, device="cuda")
st = ops.row_stats(x, 1e-5)
for ws in (1, 0):
    ops.tune_set("GEMM_WS", ws)
    for _ in range(4):
        ops.linear(x, wp, bp, geglu=True)
        ops.linear(x, w, bias, ln_stats=st, ln_colsum=colsum)
    torch.cuda.synchronize()
ops.tune_set("GEMM_WS", 1)
