"""A/B of the two flash-attention kernels (knob FLASH_IMPL: 1 = phased v1, 2 = software-pipelined v2) on the UNet's self-attention
shapes, interleaved rounds in one process, median / min ms and TF/s, plus the largest difference between the two outputs.
    python tools/flash_ab.py [rounds] [library]
With a library built with -DVCX_FLASH2_ABLATIONS (tools/build_abl.sh flash -DVCX_FLASH2_ABLATIONS) the stream variants of the pipelined
kernel run side by side as well (knob EXP1 = 10 + VAR: 1 = permuted K rows / no half swaps, 2 = paired row sums)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
ABL = len(sys.argv) > 2
if ABL:
    _lib.LIB_PATH = os.path.join(ROOT, sys.argv[2])
from viewcrafter_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [(9216, 50, 5), (2304, 50, 10), (1152, 50, 10), (576, 50, 20)]      # (tokens, frames, heads)
torch.manual_seed(0)
for N, G, heads in shapes:
    C = heads * 64
    qk = torch.randn(G * N, 2 * C, device="cuda")
    qk[:, :C] *= (0.125 * ops.LOG2E)
    qk = qk.half()
    vt = torch.randn(C, G * N, device="cuda").half()
    variants = {"v1": (1, 0), "v2": (2, 0)}       # name: (FLASH_IMPL, EXP1); with tools/_abl/libvcx_abl.so also "v2-mfma-sum": (2, 2)
    if ABL:
        variants.update({"var0": (2, 10), "var1": (2, 11), "mfma-sum": (2, 2)})
    outs, times = {}, {k: [] for k in variants}
    for k in variants:
        outs[k] = torch.empty(G * N, C, device="cuda", dtype=torch.float16)

    def run(k):
        ops.flash_attn(qk, qk[:, C:], vt, outs[k], n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * C, ldk=2 * C,
                       ldvt=G * N, ldo=C, scale=0.125, log2_logits=True)
    for r in range(rounds):
        for k, (impl, e1) in variants.items():
            ops.tune_set("FLASH_IMPL", impl); ops.tune_set("EXP1", e1)
            run(k); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): run(k)
            b.record(); torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) / 5)
    ops.tune_set("FLASH_IMPL", 0); ops.tune_set("EXP1", 0)
    fl = 4.0 * G * heads * N * N * 64
    row = f"N={N:5d} G={G} heads={heads:2d}: "
    for k in variants:
        t = sorted(times[k]); med = t[len(t) // 2]
        d = (outs["v1"].float() - outs[k].float()).abs().max().item()
        row += f"{k} {med:6.3f} ms (min {t[0]:6.3f}) {fl / med / 1e9:5.0f} TF/s d={d:.1e} | "
    print(row, flush=True)
