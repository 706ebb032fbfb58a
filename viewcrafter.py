"""ViewCrafter pipeline object with the diffusion leg (setup_diffusion / run_diffusion) on the MI355X-native path.

Surface kept from the reference's viewcrafter.py: `ViewCrafter(opts, gradio=False)`, attributes `diffusion` and
`noise_shape`, `setup_diffusion()` (:384-404), `run_diffusion(renderings[T,H,W,3] in [0,1]) -> [T,H,W,3] in [-1,1]`
(:93-106) and the `nvs_*` modes.  DUSt3R and the PyTorch3D point render stay on the reference (BASELINE.json
north_star): the `nvs_*` modes import them lazily from a reference checkout and delegate everything except
run_diffusion to it.  `nvs_from_renderings` runs the diffusion leg alone on saved renders.
"""
import importlib
import os
import random
import sys

import numpy as np
import torch

from viewcrafter_amd import parallel
from viewcrafter_amd.builder import build_diffusion_model
from viewcrafter_amd.utils.diffusion_utils import image_guided_synthesis


def _reference_root(opts):
    root = getattr(opts, "reference_root", None) or os.environ.get("VIEWCRAFTER_REFERENCE")
    if root and os.path.isdir(root):
        return root
    return None


class ViewCrafter:
    def __init__(self, opts, gradio=False):
        self.opts = opts
        self.device = opts.device
        self.gradio = gradio
        self._ref = None
        self.setup_diffusion()
        rank, world = parallel.rank_world()
        # sparse-view mode under a torchrun launch: DUSt3R and the point render run on rank 0 ONLY (they are the reference's, and
        # identical on every rank); the clips they produce are broadcast and sharded (nvs_sparse_view_interp below)
        geometry_here = not (world > 1 and rank != 0 and getattr(opts, "mode", None) == "sparse_view_interp")
        if getattr(opts, "renderings", None) is None and not gradio and geometry_here:
            self._attach_reference_geometry()

    # ------------------------------------------------------------------ diffusion leg (this repo)
    def setup_diffusion(self):
        """Reference viewcrafter.py:384-404."""
        # = pytorch_lightning.seed_everything(seed): python, numpy and torch (CPU + every GPU) generators
        random.seed(self.opts.seed)
        np.random.seed(self.opts.seed)
        torch.manual_seed(self.opts.seed)
        root = _reference_root(self.opts)
        if root and root not in sys.path:
            sys.path.append(root)           # lets the YAML's CLIP / Resampler targets resolve to the reference
        rank, world = parallel.rank_world()
        # one process per GPU (torchrun): rank 0 reads the checkpoint, the others receive the weights in a few large RCCL
        # broadcasts over xGMI (parallel.broadcast_module_) instead of W reads of the same 10 GB file
        found = [os.path.exists(self.opts.ckpt_path) if rank == 0 else True]
        if world > 1:       # rank 0's verdict reaches everybody BEFORE the weight broadcast: all ranks fail together instead of
            torch.distributed.broadcast_object_list(found, src=0)     # the others waiting in a collective rank 0 never joins
        assert found[0], "Error: checkpoint Not Found!"
        model = build_diffusion_model(self.opts.config, device=self.device, ckpt_path=self.opts.ckpt_path if rank == 0 else None,
                                      perframe_ae=self.opts.perframe_ae, conditioners="config", init_on_device=False)
        if world > 1:
            parallel.broadcast_module_(model, src=0)
        if model.cond_stage_model is not None:
            model.cond_stage_model.device = self.device
        self.diffusion = model
        h, w = self.opts.height // 8, self.opts.width // 8
        channels = model.model.diffusion_model.out_channels
        self.noise_shape = [self.opts.bs, channels, self.opts.video_length, h, w]

    def run_diffusion(self, renderings):
        """Reference viewcrafter.py:93-106.  No autocast context is needed: the kernels are fp16-storage /
        fp32-accumulate by construction."""
        prompts = [self.opts.prompt]
        videos = (renderings * 2. - 1.).permute(3, 0, 1, 2).unsqueeze(0).to(self.device)
        condition_index = [0]
        with torch.no_grad():
            batch_samples = image_guided_synthesis(
                self.diffusion, prompts, videos, self.noise_shape, self.opts.n_samples, self.opts.ddim_steps,
                self.opts.ddim_eta, self.opts.unconditional_guidance_scale, self.opts.cfg_img, self.opts.frame_stride,
                self.opts.text_input, self.opts.multiple_cond_cfg, self.opts.timestep_spacing, self.opts.guidance_rescale,
                condition_index)
        return torch.clamp(batch_samples[0][0].permute(1, 2, 3, 0), -1., 1.)

    def run_diffusion_many(self, clips):
        """Independent trajectories / clips (each one `run_diffusion` call = one image_guided_synthesis) sharded over the
        ranks: rank r runs clips r, r + W, ... with NO collective inside the DDIM loop; the decoded clips are gathered on
        rank 0 with one all_gather.  Returns the list of results on rank 0, None elsewhere; with one
        process it is a plain loop.  Clip i is seeded with opts.seed + i, so the result does not depend on the world size
        (clip 0 equals the reference's single-process run; later clips differ from the reference's one sequential generator
        stream - `nvs_sparse_view_interp` therefore keeps the reference's own loop when there is one process, and the
        sharded launch is documented as a different, world-size-independent stream)."""
        _, world = parallel.rank_world()

        def one(clip, index):
            if world > 1 or index > 0:
                torch.manual_seed(self.opts.seed + index)
            return self.run_diffusion(clip)
        # VCX_CLIPS_PER_GPU=2: two of a rank's clips in flight at a time, step by step on two HIP streams (viewcrafter_amd/interleave.py;
        # same outputs).  OPT-IN since round 6: the builder's boxes measured +3 ... +9 % aggregate rate, the driver's box of round 5
        # -8.7 % (BENCH_r05.json extra.two_clips_per_gpu gain 0.913) - a mode whose sign depends on the box is not a default; it also
        # doubles the activation memory of a rank.  Default: one clip after the other.
        try:
            lanes = max(1, int(os.environ.get("VCX_CLIPS_PER_GPU", "1")))
        except ValueError:
            lanes = 1
        return parallel.run_sharded(one, list(clips), gather=True, lanes=lanes, model=self.diffusion)

    def nvs_from_renderings(self, path):
        """Diffusion leg only: `path` holds point-cloud renders [T, H, W, 3] in [0, 1] (.pt or .npy) - or several
        trajectories, as a comma-separated list of such files or one [N, T, H, W, 3] tensor; those are sharded over the
        ranks of a torchrun launch (run_diffusion_many).  Rank 0 writes diffusion<i>.pt / .mp4."""
        from viewcrafter_amd.utils.video_io import save_video

        def load(p):
            return (torch.load(p) if p.endswith(".pt") else torch.from_numpy(np.load(p))).float()
        clips = []
        for p in path.split(","):
            r = load(p.strip())
            clips.extend(list(r) if r.dim() == 5 else [r])
        rank, _ = parallel.rank_world()
        outs = self.run_diffusion_many(clips)
        if rank == 0:
            for i, out in enumerate(outs):
                torch.save(out.cpu(), os.path.join(self.opts.save_dir, f"diffusion{i}.pt"))
                # like the reference's nvs_single_view (viewcrafter.py:118-121): write the generated clip as a video as well
                save_video((out + 1.0) / 2.0, os.path.join(self.opts.save_dir, f"diffusion{i}.mp4"), fps=8, value_range=(0.0, 1.0))
        return outs[0] if (rank == 0 and len(clips) == 1) else outs

    # ------------------------------------------------------------------ geometry stages (reference)
    def _attach_reference_geometry(self):
        root = _reference_root(self.opts)
        if root is None:
            raise RuntimeError(
                "DUSt3R and the PyTorch3D point render stay on the reference implementation: pass --reference_root "
                "(or set VIEWCRAFTER_REFERENCE) to a Drexubery/ViewCrafter checkout, or use --renderings to run the "
                "diffusion leg on saved renders.")
        if root not in sys.path:
            sys.path.append(root)
        spec = importlib.util.spec_from_file_location("viewcrafter_reference", os.path.join(root, "viewcrafter.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ours = self

        class _Geometry(mod.ViewCrafter):
            _record = None                      # a list: run_diffusion only records its clips (multi-GPU sparse-view mode)

            def setup_diffusion(self):          # the diffusion model is ours
                self.diffusion, self.noise_shape = ours.diffusion, ours.noise_shape

            def run_diffusion(self, renderings):
                if self._record is not None:
                    self._record.append(renderings)
                    return torch.zeros_like(renderings)
                return ours.run_diffusion(renderings)
        self._ref = _Geometry(self.opts, gradio=self.gradio)

    def __getattr__(self, name):
        """Everything else the reference's class offers (run_dust3r, load_initial_images, the scene / image attributes its
        methods leave behind, ...) lives on the attached reference object."""
        ref = self.__dict__.get("_ref")
        if ref is not None and not name.startswith("__"):
            return getattr(ref, name)
        raise AttributeError(f"{type(self).__name__!s} has no attribute {name!r}"
                             + ("" if ref is not None else " (no reference checkout attached: geometry stages unavailable)"))

    def nvs_single_view(self, gradio=False):
        return self._ref.nvs_single_view(gradio)

    def nvs_sparse_view_interp(self):
        """Reference viewcrafter.py:196-277.  Its (N - 1) clips are independent `run_diffusion` calls (:272-274): under a
        torchrun launch the reference's method runs in recording mode on RANK 0 ONLY (DUSt3R and the render are the
        reference's; the clips are only collected), the clips are broadcast (parallel.broadcast_tensor_list), sharded over the GPUs and rank 0
        writes diffusion.mp4 (fps 8, the reference's writer default, pvd_utils.py:38) - instead of (N - 1) sequential 11 s
        generations on one GPU.  One process: the reference's own loop, i.e. its single sequential noise stream; N processes:
        clip i draws from seed + i (independent of N), so clips 1.. differ from the one-process run - by construction, a
        sharded launch cannot replay one sequential generator."""
        rank, world = parallel.rank_world()
        if world == 1:
            return self._ref.nvs_sparse_view_interp()
        from viewcrafter_amd.utils.video_io import save_video
        clips, error = None, None
        if rank == 0:       # geometry once: the reference's method in recording mode (run_diffusion only collects its clips)
            clips = []
            self._ref._record = clips
            try:
                self._ref.nvs_sparse_view_interp()
            except Exception as e:          # reported to every rank by the broadcast below: all fail together, nobody hangs
                error = f"{type(e).__name__}: {e}"
            finally:
                self._ref._record = None
        clips = parallel.broadcast_tensor_list(clips, src=0, error=error)
        outs = self.run_diffusion_many(clips)
        if rank != 0:
            return None
        result = torch.cat(outs)
        save_video((result + 1.0) / 2.0, os.path.join(self.opts.save_dir, "diffusion.mp4"), fps=8, value_range=(0.0, 1.0))
        return result

    def nvs_single_view_eval(self):
        return self._ref.nvs_single_view_eval()

    def run_gradio(self, *args, **kwargs):
        return self._ref.run_gradio(*args, **kwargs)
