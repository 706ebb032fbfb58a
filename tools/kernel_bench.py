"""Micro-benchmark of the libvcx kernels at the shapes of one UNet forward.

Usage (on the GPU box):  python tools/kernel_bench.py [--quick]
Prints one line per case: time per launch, TFLOP/s or GB/s.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops  # noqa: E402
from viewcrafter_amd.packing import pack_conv, pack_geglu  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def rh(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).half()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    ops.require_gpu()
    print("device", torch.cuda.get_device_name(0))
    B = 2  # cond+uncond batched
    T, HW0 = 25, 72 * 128
    levels = [(320, 72, 128), (640, 36, 64), (1280, 18, 32), (1280, 9, 16)]

    print("== linear (tokens x C -> N) ==")
    for C, h, w in levels[:3]:
        M = B * T * h * w
        x = rh(M, C)
        for N, tag in [(C, "proj"), (3 * C, "qkv")]:
            wt = rh(N, C, scale=1 / math.sqrt(C))
            ms = timeit(lambda: ops.linear(x, wt))
            print(f"linear M={M} K={C} N={N} {tag}: {ms:.3f} ms  {2*M*N*C/ms/1e9:.1f} TF/s")
        w8 = torch.randn(8 * C, C, device=DEV) / math.sqrt(C)
        b8 = torch.randn(8 * C, device=DEV)
        wp, bp = pack_geglu(w8, b8)
        wp = wp.half()
        ms = timeit(lambda: ops.linear(x, wp, bp, geglu=True))
        print(f"geglu  M={M} K={C} N={8*C}: {ms:.3f} ms  {2*M*8*C*C/ms/1e9:.1f} TF/s")
        g = rh(M, 4 * C)
        w2 = rh(C, 4 * C, scale=1 / math.sqrt(4 * C))
        ms = timeit(lambda: ops.linear(g, w2, None, residual=x))
        print(f"ff2    M={M} K={4*C} N={C}: {ms:.3f} ms  {2*M*4*C*C/ms/1e9:.1f} TF/s")
        del g, w2, wp, w8

    print("== conv3x3 / temporal conv ==")
    for C, h, w in levels:
        x = rh(B * T, h, w, C)
        wc = pack_conv(rh(C, C, 3, 3, scale=1 / math.sqrt(9 * C)))
        bias = torch.randn(C, device=DEV)
        ms = timeit(lambda: ops.conv2d(x, wc, bias, kh=3, kw=3))
        fl = 2 * B * T * h * w * C * C * 9
        print(f"conv3x3 C={C} {h}x{w}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF/s")
        wt = pack_conv(rh(C, C, 3, 1, 1, scale=1 / math.sqrt(3 * C)))
        x5 = x.view(B, T, h * w, C)
        ms = timeit(lambda: ops.temporal_conv3(x5, wt, bias))
        print(f"tconv3  C={C} {h}x{w}: {ms:.3f} ms  {fl/3/ms/1e9:.1f} TF/s")

    print("== flash self-attention ==")
    for C, h, w in levels[:3]:
        heads, N, G = C // 64, h * w, B * T
        qk = rh(G * N, 2 * C)
        vt = rh(C, G * N)
        out = torch.empty(G * N, C, device=DEV, dtype=torch.float16)
        ms = timeit(lambda: ops.flash_attn(qk, qk[:, C:], vt, out, n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N,
                                           kv_div=1, ldq=2 * C, ldk=2 * C, ldvt=G * N, ldo=C, scale=0.125),
                    iters=3 if N > 5000 else 10, warm=1)
        fl = 4.0 * G * heads * N * N * 64
        print(f"flash N={N} heads={heads} groups={G}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF/s")
        del qk, vt, out

    print("== cross attention (77 text + 256 image keys) ==")
    C, h, w = levels[0]
    heads, N, G = C // 64, h * w, B * T
    q = rh(G * N, C)
    kt, vtt = rh(B * 80, C), rh(C, B * 80)
    ki, vit = rh(B * 256, C), rh(C, B * 256)
    out = torch.empty(G * N, C, device=DEV, dtype=torch.float16)

    def cross():
        ops.flash_attn(q, kt, vtt, out, n_groups=G, heads=heads, nq=N, nk=77, kv_rows=80, kv_div=T, ldq=C, ldk=C,
                       ldvt=B * 80, ldo=C, scale=0.125)
        ops.flash_attn(q, ki, vit, out, n_groups=G, heads=heads, nq=N, nk=256, kv_rows=256, kv_div=T, ldq=C, ldk=C,
                       ldvt=B * 256, ldo=C, scale=0.125, accumulate=True)
    ms = timeit(cross)
    print(f"cross N={N}: {ms:.3f} ms  {4.0*G*heads*N*333*64/ms/1e9:.1f} TF/s")

    print("== temporal attention ==")
    for C, h, w in levels[:3]:
        heads, P = C // 64, h * w
        qkv = rh(B * T * P, 3 * C)
        out = torch.empty(B * T * P, C, device=DEV, dtype=torch.float16)
        ms = timeit(lambda: ops.temporal_attn(qkv, out, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C,
                                              ldo=C, scale=0.125))
        byts = B * T * P * C * 2 * 4
        print(f"tattn C={C} P={P}: {ms:.3f} ms  {byts/ms/1e6:.0f} GB/s")

    print("== groupnorm / layernorm ==")
    for C, h, w in levels[:3]:
        x = rh(B * T, h * w, C)
        g, bta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        ms = timeit(lambda: ops.group_norm(x, g, bta, 1e-5, True))
        print(f"gn per-frame C={C}: {ms:.3f} ms  {x.numel()*6/ms/1e6:.0f} GB/s (3 passes x 2 B)")
        x5 = x.view(B, T * h * w, C)
        ms = timeit(lambda: ops.group_norm(x5, g, bta, 1e-5, True))
        print(f"gn per-video C={C}: {ms:.3f} ms  {x.numel()*6/ms/1e6:.0f} GB/s")
        x2 = x.view(-1, C)
        ms = timeit(lambda: ops.layer_norm(x2, g, bta))
        print(f"layernorm C={C}: {ms:.3f} ms  {x.numel()*4/ms/1e6:.0f} GB/s (2 B in + 2 B out)")

    if not args.quick:
        print("== VAE decoder convs (one 576x1024 frame) ==")
        for C, h, w in [(512, 144, 256), (256, 288, 512), (128, 576, 1024)]:
            x = rh(1, h, w, C)
            wc = pack_conv(rh(C, C, 3, 3, scale=1 / math.sqrt(9 * C)))
            bias = torch.randn(C, device=DEV)
            ms = timeit(lambda: ops.conv2d(x, wc, bias, kh=3, kw=3), iters=5)
            print(f"vae conv3x3 C={C} {h}x{w}: {ms:.3f} ms  {2*h*w*C*C*9/ms/1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
