"""Two independent trajectories per GPU on two HIP streams (review r4 item 7): does the aggregate rate of two concurrent B = 2 (cond +
uncond) UNet forwards beat two sequential ones?  The 21 ms of HBM-bound kernels and the HBM-bound K = 320 GEMMs of one trajectory could
run under the MFMA-bound kernels of the other - if the blocks of two kernels can be resident together.

    python tools/two_stream_ab.py [--workload ViewCrafter_25_576x1024x25] [--rounds 4]

Prints ms per PAIR of forwards: sequential on one stream, and issued on two streams (the host queues a whole forward - ~1000 launches, ~35 ms
of Python - on stream A, then one on stream B, and is far ahead of the GPU from the second pair on)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ViewCrafter_25_576x1024x25")
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    from bench import WORKLOADS, synth_conditioning
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    cfg, T, h, w = WORKLOADS[args.workload]
    model = build_diffusion_model(os.path.join(ROOT, "configs", cfg), device="cuda", conditioners="identity")
    randomize_parameters(model)
    x1, cond, uc = synth_conditioning(T, h, w, "cuda")
    x2 = torch.randn_like(x1)
    both = {"c_crossattn": [torch.cat([cond["c_crossattn"][0], uc["c_crossattn"][0]], 0)], "c_concat": cond["c_concat"]}
    ts = torch.full((1,), 499, device="cuda", dtype=torch.long)
    fs = torch.tensor([10], device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def fwd(x):
        return model.apply_model(x, ts, both, fs=fs, cfg_repeat=2)

    with torch.no_grad():
        y1 = fwd(x1).clone(); y2 = fwd(x2).clone()          # warm: weight packing, cross-attention K / V cache
        for s, x in ((sa, x1), (sb, x2)):
            with torch.cuda.stream(s):
                fwd(x)
        torch.cuda.synchronize()

        def timed(two_streams, pairs=3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            outs = None
            for _ in range(pairs):
                if two_streams:
                    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(sa):
                        a = fwd(x1)
                    with torch.cuda.stream(sb):
                        b = fwd(x2)
                    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
                else:
                    a = fwd(x1); b = fwd(x2)
                outs = (a, b)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / pairs, outs
        rows = {"sequential": [], "two streams": []}
        for _ in range(args.rounds):
            for k in rows:
                ms, outs = timed(k == "two streams")
                rows[k].append(ms)
        same = torch.equal(outs[0], y1) and torch.equal(outs[1], y2)
    print(f"{args.workload}: ms per pair of B = 2 forwards (two trajectories), {args.rounds} interleaved rounds of 3 pairs")
    for k, v in rows.items():
        print(f"  {k:12s} median {sorted(v)[len(v) // 2]:8.2f}   all {[round(t, 2) for t in v]}")
    m = {k: sorted(v)[len(v) // 2] for k, v in rows.items()}
    print(f"  two streams / sequential = {m['two streams'] / m['sequential']:.4f}   outputs bit-identical to the single-stream forwards: {same}")


if __name__ == "__main__":
    main()
