"""Does the feed-forward pair of level 0 (GEGLU projection 460800 x 2560 x 320 -> Linear 460800 x 320 x 1280 + residual) run faster in row
chunks - chunk i's projection directly followed by chunk i's output layer, so that the 1.18 GB intermediate is re-read from the 256 MB
Infinity Cache instead of HBM?   python tools/ff_chunk_ab.py [iters]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from viewcrafter_amd import ops
    from viewcrafter_amd.packing import pack_geglu
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = "cuda"
    for M, C in ((460800, 320), (115200, 640)):
        g = torch.Generator().manual_seed(3)
        t = torch.randn(M, C, generator=g).to(dev).half()
        w1, b1 = pack_geglu((torch.randn(8 * C, C, generator=g) / math.sqrt(C)).to(dev).half(), torch.randn(8 * C, generator=g).to(dev))
        w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).to(dev).half()
        b2 = torch.randn(C, generator=g).to(dev)
        inter = torch.empty(M, 4 * C, device=dev, dtype=torch.float16)
        out = torch.empty(M, C, device=dev, dtype=torch.float16)

        def pair(chunks):
            rows = M // chunks
            for c in range(chunks):
                r0, r1 = c * rows, (c + 1) * rows if c + 1 < chunks else M
                ops.linear(t[r0:r1], w1, b1, geglu=True, out=inter[r0:r1])
                ops.linear(inter[r0:r1], w2, b2, residual=t[r0:r1], out=out[r0:r1])
        ref = None
        for chunks in (1, 2, 4, 5, 8, 10, 16, 1):
            pair(chunks)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = torch.equal(out, ref)
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    pair(chunks)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters)
            print(f"M={M} C={C} chunks={chunks:2d}: {best:7.3f} ms per pair  (same bits as unchunked: {same})", flush=True)


if __name__ == "__main__":
    main()
