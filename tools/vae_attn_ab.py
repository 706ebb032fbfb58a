"""VAE AttnBlock at the 576x1024 decode's shape (25 frames x 9216 tokens x 512 channels): the d = 512 flash kernel against the GEMM -> row
softmax -> GEMM sequence of rounds 1-4 (VCX_VAE_FUSED_ATTN).    python tools/vae_attn_ab.py [frames]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd.lvdm.modules.networks import ae_modules
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
blk = ae_modules.AttnBlock(512).cuda().eval()
with torch.no_grad():
    for p_ in blk.parameters():
        p_.normal_(0, 0.03)
    blk.norm.weight.fill_(1.0)
x = torch.randn(n, 72, 128, 512, device="cuda").half()
def t(fused, it=3):
    ae_modules.FUSED_ATTN = fused
    with torch.no_grad():
        y = blk(x); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): y = blk(x)
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it, y
for r in range(2):
    t1, y1 = t(True); t0, y0 = t(False)
    e = float((y1.float() - y0.float()).norm() / y0.float().norm())
    print(f"AttnBlock {n} x 9216 x 512: flash d512 {t1:.2f} ms, GEMM / softmax / GEMM per frame {t0:.2f} ms, rel-L2 between them {e:.2e}", flush=True)
