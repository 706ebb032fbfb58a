"""The entry points on the host side: `python inference.py` (BASELINE configs[0] plumbing) and its multi-GPU trajectory sharding.

No GPU here, and deliberately no CPU compute path: the CLI must construct the model from the YAML, strict-load the
checkpoint, build `noise_shape` and then fail loudly at the first kernel launch.  The torchrun path is driven with two
gloo ranks and a stub diffusion model (the same `inference.main` the GPUs run).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests.util import write_tiny_entry_files


def test_inference_cli_plumbing_until_the_first_kernel(tmp_path, monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_entry_gpu.py runs the command to the end")
    import inference
    import viewcrafter
    from viewcrafter_amd._lib import VcxError
    ypath, cpath, rpath, (T, H, W) = write_tiny_entry_files(tmp_path)
    built = {}
    real_init = viewcrafter.ViewCrafter.setup_diffusion

    def spy(self):
        real_init(self)
        built["noise_shape"], built["model"] = self.noise_shape, self.diffusion
    monkeypatch.setattr(viewcrafter.ViewCrafter, "setup_diffusion", spy)
    argv = ["--renderings", rpath, "--config", ypath, "--ckpt_path", cpath, "--out_dir", str(tmp_path / "out"), "--exp_name", "e",
            "--device", "cpu", "--ddim_steps", "5", "--video_length", str(T), "--height", str(H), "--width", str(W), "--prompt", ""]
    with pytest.raises(VcxError, match="no CPU fallback"):
        inference.main(argv)
    assert built["noise_shape"] == [1, 4, T, H // 8, W // 8]                      # viewcrafter.py:399-404
    m = built["model"]
    assert m.perframe_ae is True and m.model.conditioning_key == "hybrid" and not m.training
    # the checkpoint really went in (strict): a synthetic tensor, not the constructor's initialisation
    from oracle.weights import synth_tensor
    w = m.model.diffusion_model.out[2].weight
    assert torch.equal(w, synth_tensor("model.diffusion_model.out.2.weight", w.shape))
    assert os.path.isdir(tmp_path / "out" / "e")


def test_invalid_mode_and_missing_reference_are_reported(tmp_path, monkeypatch):
    import inference
    import viewcrafter
    monkeypatch.setattr(viewcrafter.ViewCrafter, "setup_diffusion", lambda self: None)
    monkeypatch.delenv("VIEWCRAFTER_REFERENCE", raising=False)
    with pytest.raises(RuntimeError, match="reference"):     # geometry stages need a reference checkout
        inference.main(["--out_dir", str(tmp_path), "--exp_name", "x", "--mode", "single_view_txt"])


# ------------------------------------------------------------------ torchrun path, two gloo ranks, stub model
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.model = torch.nn.Module()
        self.model.diffusion_model = torch.nn.Module()
        self.model.diffusion_model.out_channels = 4
        self.cond_stage_model = None
        self.w = torch.nn.Parameter(torch.randn(7))


def _worker(rank, world, port, tmp, q, sparse):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import inference
    import viewcrafter

    def build(config, device="cpu", ckpt_path=None, **kw):
        assert (ckpt_path is not None) == (rank == 0), "only rank 0 reads the checkpoint"
        torch.manual_seed(1000 + rank)                   # every rank starts with different "weights"
        return _StubModel()

    def synth(model, prompts, videos, noise_shape, *a, **kw):
        # depends on the clip, on the (broadcast) weights and on the per-clip seed -> [B, n_samples, 3, T, H, W]
        tag = 0.1 * videos.mean() + 0.01 * model.w.detach().sum() + 0.01 * torch.randn(())
        return (videos * 0 + tag).unsqueeze(1)
    viewcrafter.build_diffusion_model = build
    viewcrafter.image_guided_synthesis = synth
    argv = ["--config", "none.yaml", "--ckpt_path", os.path.join(tmp, "ckpt"), "--out_dir", os.path.join(tmp, "out"), "--exp_name", "e",
            "--device", "cpu", "--video_length", "3", "--height", "16", "--width", "16", "--seed", "11"]
    if sparse:     # a stand-in "reference checkout" whose nvs_sparse_view_interp loops over run_diffusion like viewcrafter.py:272-276
        argv += ["--mode", "sparse_view_interp", "--reference_root", os.path.join(tmp, "ref")]
    else:
        argv += ["--renderings", ",".join(os.path.join(tmp, f"r{i}.pt") for i in range(2)) + "," + os.path.join(tmp, "r_many.pt")]
    out = inference.main(argv)
    q.put((rank, None if out is None else [float(o.flatten()[0]) for o in (out if isinstance(out, list) else list(out.view(-1, 3, 16, 16, 3)))]))


_FAKE_REFERENCE = '''
import torch
class ViewCrafter:
    def __init__(self, opts, gradio=False):
        self.opts = opts
        self.setup_diffusion()
    def nvs_sparse_view_interp(self):
        import os
        open(os.path.join(os.path.dirname(__file__), "geometry_ran.rank" + os.environ.get("RANK", "0")), "w").close()
        renders = torch.arange(5 * 2 + 1, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 16, 16, 3) / 100      # 5 clips of 3 frames sharing ends
        res = [self.run_diffusion(renders[i * 2: 3 + i * 2]) for i in range(5)]
        return torch.cat(res)
'''


def _expected(clip_means, w_sum, seed):
    out = []
    for i, m in enumerate(clip_means):
        torch.manual_seed(seed + i)
        out.append(float(0.1 * (torch.tensor(m) * 2 - 1) + 0.01 * w_sum + 0.01 * torch.randn(())))
    return out


@pytest.mark.parametrize("sparse", [False, True])
def test_inference_main_shards_trajectories_over_two_ranks(tmp_path, sparse):
    tmp = str(tmp_path)
    open(os.path.join(tmp, "ckpt"), "w").write("x")
    os.makedirs(os.path.join(tmp, "ref"))
    open(os.path.join(tmp, "ref", "viewcrafter.py"), "w").write(_FAKE_REFERENCE)
    means = [0.1, 0.2, 0.3, 0.4, 0.5]
    for i in range(2):
        torch.save(torch.full((3, 16, 16, 3), means[i]), os.path.join(tmp, f"r{i}.pt"))
    torch.save(torch.stack([torch.full((3, 16, 16, 3), m) for m in means[2:]]), os.path.join(tmp, "r_many.pt"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q, sparse)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[1] is None and len(outs[0]) == 5           # results live on rank 0 only, in clip order
    torch.manual_seed(1000)
    w_sum = float(_StubModel().w.detach().sum())                     # rank 0's weights, which rank 1 must have received
    if sparse:
        # DUSt3R + render (here: the stand-in's nvs_sparse_view_interp) ran on rank 0 only; rank 1 received the clips by broadcast
        assert os.path.exists(os.path.join(tmp, "ref", "geometry_ran.rank0")) and not os.path.exists(os.path.join(tmp, "ref", "geometry_ran.rank1"))
        clip_means = [sum(range(i * 2, i * 2 + 3)) / 3 / 100 for i in range(5)]
    else:
        clip_means = means
    exp = _expected(clip_means, w_sum, 11)
    assert outs[0] == pytest.approx(exp, abs=1e-5), (outs[0], exp)
    assert os.path.exists(os.path.join(tmp, "out", "e", "diffusion.avi" if sparse else "diffusion4.avi")) or \
        os.path.exists(os.path.join(tmp, "out", "e", "diffusion.mp4" if sparse else "diffusion4.mp4"))


def test_viewcrafter_forwards_the_reference_class_surface(tmp_path, monkeypatch):
    """Methods and attributes of the reference's ViewCrafter that this repo does not re-implement (run_dust3r, load_initial_images,
    the scene state its methods leave behind) resolve on the attached reference object; without a checkout the error says so."""
    import viewcrafter
    os.makedirs(tmp_path / "ref")
    (tmp_path / "ckpt").write_text("x")
    (tmp_path / "ref" / "viewcrafter.py").write_text(_FAKE_REFERENCE + '''
    def run_dust3r(self, input_images, clean_pc=False):
        self.scene = ("scene of", len(input_images))
        return self.scene
''')
    monkeypatch.setattr(viewcrafter, "build_diffusion_model", lambda *a, **k: _StubModel())
    from configs.infer_config import get_parser
    base = ["--config", "none.yaml", "--ckpt_path", str(tmp_path / "ckpt"), "--out_dir", str(tmp_path / "out"), "--exp_name", "e",
            "--device", "cpu", "--video_length", "3", "--height", "16", "--width", "16"]
    opts = get_parser().parse_args(base + ["--mode", "sparse_view_interp", "--reference_root", str(tmp_path / "ref")])
    opts.save_dir = str(tmp_path / "out")
    os.makedirs(opts.save_dir, exist_ok=True)
    vc = viewcrafter.ViewCrafter(opts)
    assert vc.run_dust3r([1, 2, 3]) == ("scene of", 3) and vc.scene == ("scene of", 3)
    with pytest.raises(AttributeError):
        vc.no_such_thing
    bare = viewcrafter.ViewCrafter.__new__(viewcrafter.ViewCrafter)
    with pytest.raises(AttributeError, match="no reference checkout attached"):
        bare.run_dust3r


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it must itself become two ranks: the harness is run in
    --stub mode (gloo ranks, stand-in step, no model) and rank 0 prints ONE line with n_gpus = the real world size."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["backend"] == "gloo" and out["data"] == "stub" and out["scaling"] == "weak"
    assert len(out["per_rank_steps_per_s"]) == 2 and all(v > 0 for v in out["per_rank_steps_per_s"])
    assert out["value"] == pytest.approx(2 * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"]), rel=1e-6)
    assert out["value"] <= sum(out["per_rank_steps_per_s"]) * (1 + 1e-6)       # max-over-ranks time, barrier included
    assert "torch.distributed.run" in r.stderr


def test_bench_refuses_a_launch_line_that_disagrees_with_gpus():
    r, lines = _run_bench(["--gpus", "2", "--stub"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines and "must agree" in r.stderr
    r, lines = _run_bench(["--gpus", "1", "--steps", "2", "--stub"])
    assert r.returncode == 0 and lines[0]["n_gpus"] == 1 and lines[0]["backend"] is None
