"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from oracle.weights import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SCHEDULE_BUFFERS = ("betas", "alphas", "sqrt_", "log_one", "posterior", "scale_arr", "lvlb", "logvar")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def load_synth(module, seed=0, skip=()):
    """Fill a module with the deterministic synthetic weights keyed by its own state-dict names."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(shapes, seed=seed, skip=skip)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return sd


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten() if not torch.is_tensor(a) else a.detach().double().cpu().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten() if not torch.is_tensor(b) else b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def psnr(a, b, peak=2.0):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    mse = float(((a - b) ** 2).mean())
    return 10 * np.log10(peak * peak / max(mse, 1e-30))


def write_tiny_entry_files(tmp_path, device_for_build="cpu"):
    """What `python inference.py --renderings ...` needs, for the tiny graph: a model YAML (the reference's class paths as
    targets, the tiny OpenCLIP towers given inline), a synthetic Lightning checkpoint with every key of the strict load,
    and point-cloud renders [T, H, W, 3] in [0, 1].  Returns (yaml path, ckpt path, renders path, (T, H, W))."""
    import yaml
    from tests.tiny_config import CLIP_TINY_CFG, IGS_H, IGS_T, IGS_W, igs_model_params
    from viewcrafter_amd.config import Config
    from viewcrafter_amd.utils.diffusion_utils import instantiate_from_config
    R = "lvdm.modules.encoders."
    params = igs_model_params("lvdm.modules.networks.openaimodel3d.UNetModel", "lvdm.models.autoencoder.AutoencoderKL",
                              R + "condition.FrozenOpenCLIPEmbedder", R + "condition.FrozenOpenCLIPImageEmbedderV2",
                              R + "resampler.Resampler", arch=CLIP_TINY_CFG)
    cfg = {"model": {"target": "lvdm.models.ddpm3d.VIPLatentDiffusion", "params": params}}
    ypath = os.path.join(str(tmp_path), "tiny_inference.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump(cfg, f)
    model = instantiate_from_config(Config.wrap(cfg["model"]))
    load_synth(model, skip=SCHEDULE_BUFFERS)
    cpath = os.path.join(str(tmp_path), "tiny.ckpt")
    torch.save({"state_dict": model.state_dict(), "global_step": 0}, cpath)      # Lightning layout (diffusion_utils.py:85-88)
    rpath = os.path.join(str(tmp_path), "renders.pt")
    g = torch.Generator().manual_seed(5)
    torch.save(torch.rand(IGS_T, IGS_H, IGS_W, 3, generator=g), rpath)
    return ypath, cpath, rpath, (IGS_T, IGS_H, IGS_W)


def write_synthetic_bpe(path):
    """A CLIP byte-pair vocabulary file in open_clip's format (bpe_simple_vocab_16e6.txt.gz: header line, one merge per line) with
    merges LEARNED from a small corpus by the textbook BPE procedure, padded with never-matching filler merges to CLIP's 48894 so
    that the special tokens land on 49406 / 49407.  The real file is data that is not available offline; the algorithm under test
    (rank-ordered merging, </w> word ends, byte-to-unicode alphabet, CLIP's regex) does not care which merges it is given.
    Returns the list of learned merges (pairs of symbols)."""
    import gzip
    from collections import Counter
    from viewcrafter_amd.lvdm.modules.encoders import condition as cond
    corpus = ("a photo of a large room with wooden furniture and a view of the garden , camera moving forward . it's the artist's 2 "
              "chairs ! naive cafe a sweeping view of the old town at night , 4k , highly detailed").split()
    byte = cond._bytes_to_unicode()
    words = [tuple(byte[b] for b in w.encode()) for w in corpus]
    words = [w[:-1] + (w[-1] + "</w>",) for w in words]
    merges = []
    for _ in range(90):
        c = Counter()
        for w in words:
            for i in range(len(w) - 1):
                c[(w[i], w[i + 1])] += 1
        if not c:
            break
        best = max(sorted(c), key=lambda p: c[p])
        merges.append(best)
        nw = []
        for w in words:
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1])
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            nw.append(tuple(out))
        words = nw
    lines = ["#version: synthetic"] + [" ".join(m) for m in merges] + [f"zq{i}x zq{i}y" for i in range(49152 - 256 - 2 - len(merges))]
    with gzip.open(path, "wb") as f:
        f.write("\n".join(lines).encode())
    return merges
