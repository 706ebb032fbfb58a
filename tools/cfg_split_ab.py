"""(The `cfg_streams` legs need tools/rejected/cfg_streams_fork.patch applied to the UNet; without it they fall back to the batched forward.)

One trajectory, classifier-free guidance: the two evaluations of a DDIM step as ONE B = 2 forward (shipped: shared prefix, batched
kernels) against TWO B = 1 forwards - sequential, and on two HIP streams (concurrency inside a single trajectory; tools/two_stream_ab.py
measured +8.6 % aggregate rate for two concurrent trajectories).

    python tools/cfg_split_ab.py [--workload ViewCrafter_25_576x1024x25]"""
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
    x, cond, uc = synth_conditioning(T, h, w, "cuda")
    both = {"c_crossattn": [torch.cat([cond["c_crossattn"][0], uc["c_crossattn"][0]], 0)], "c_concat": cond["c_concat"]}
    ts = torch.full((1,), 499, device="cuda", dtype=torch.long)
    fs = torch.tensor([10], device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def batched():
        return model.apply_model(x, ts, both, fs=fs, cfg_repeat=2)

    def single(c):
        return model.apply_model(x, ts, c, fs=fs)

    def split(two_streams):
        if not two_streams:
            return single(cond), single(uc)
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            a = single(cond)
        with torch.cuda.stream(sb):
            b = single(uc)
        cur.wait_stream(sa); cur.wait_stream(sb)
        return a, b

    model.model.diffusion_model.cfg_streams = False
    with torch.no_grad():
        y = batched()
        ya, yb = split(False)
        split(True)
        torch.cuda.synchronize()
        print("B = 1 halves equal the B = 2 forward bit for bit:", torch.equal(torch.cat([ya, yb], 0), y))

        def timed(fn, reps=3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        unet = model.model.diffusion_model

        def forked():
            unet.cfg_streams = True
            try:
                return batched()
            finally:
                unet.cfg_streams = False
        unet.cfg_streams = False
        print("forked-stream forward equals the batched one bit for bit:", torch.equal(forked(), y))
        def graphed(streams):
            unet.cfg_streams, unet.use_hip_graph = streams, True
            try:
                return batched()
            finally:
                unet.cfg_streams, unet.use_hip_graph = False, False
        for st in (False, True):
            print(f"hipGraph replay, cfg_streams = {st}: equals the eager batched forward bit for bit:", torch.equal(graphed(st), y))
        legs = {"one B = 2 forward, one stream": batched, "shared prefix, then two streams": forked,
                "one B = 2 forward, hipGraph": lambda: graphed(False), "shared prefix, then two streams, hipGraph": lambda: graphed(True),
                "two B = 1 forwards, one stream": lambda: split(False), "two B = 1 forwards, two streams": lambda: split(True)}
        rows = {k: [] for k in legs}
        for _ in range(args.rounds):
            for k, fn in legs.items():
                rows[k].append(timed(fn))
    print(f"{args.workload}: ms per DDIM step's worth of UNet evaluation (cond + uncond)")
    for k, v in rows.items():
        print(f"  {k:34s} median {sorted(v)[len(v) // 2]:8.2f}   all {[round(t, 2) for t in v]}")


if __name__ == "__main__":
    main()
