"""Does the 256 MB Infinity Cache serve the GroupNorm apply pass if the statistics pass of the SAME rows ran just before it?
GroupNorm (+SiLU) over [n_outer, pixels, C] as one stats + one apply launch (shipped) against the same work in chunks of k
outer rows (stats(chunk) -> apply(chunk) back to back: the apply pass re-reads 2 k pixels C bytes that were streamed k rows ago).
Same box, interleaved; prints ms and effective GB/s on the algorithmic 3 x bytes."""
import sys
import time

import torch

sys.path.insert(0, ".")
from viewcrafter_amd import ops  # noqa: E402


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = "cuda"
    cases = [("level 0 per-frame GN", 50, 9216, 320), ("level 1", 50, 2304, 640), ("level 0 temporal GN (stats over T)", 2, 25 * 9216, 320),
             ("level 0 skip-concat 640", 50, 9216, 640), ("level 1 concat 1280", 50, 2304, 1280)]
    for tag, n, pixels, C in cases:
        x = torch.randn(n, pixels, C, device=dev, dtype=torch.float16)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        mb = x.numel() * 2 / 1e6
        row = [f"{tag}: {mb:.0f} MB"]
        t = bench(lambda: ops.group_norm(x, g, b, 1e-5, True, out=out))
        row.append(f"whole {t:.3f} ms ({3 * mb / t:.0f} GB/s)")
        for k in (n // 2, n // 5, n // 10, n // 25):
            if k < 1 or n % k:
                continue

            def chunked(k=k):
                for i in range(0, n, k):
                    ops.group_norm(x[i:i + k], g, b, 1e-5, True, out=out[i:i + k])
            t = bench(chunked)
            row.append(f"k={k} ({k * pixels * C * 2 / 1e6:.0f} MB): {t:.3f}")
        print("  ".join(row), flush=True)


main()
