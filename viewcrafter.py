"""ViewCrafter pipeline object with the diffusion leg (setup_diffusion / run_diffusion) on the MI355X-native path.

Surface kept from the reference's viewcrafter.py: `ViewCrafter(opts, gradio=False)`, attributes `diffusion` and
`noise_shape`, `setup_diffusion()` (:384-404), `run_diffusion(renderings[T,H,W,3] in [0,1]) -> [T,H,W,3] in [-1,1]`
(:93-106) and the `nvs_*` modes.  DUSt3R and the PyTorch3D point render stay on the reference (BASELINE.json
north_star): the `nvs_*` modes import them lazily from a reference checkout and delegate everything except
run_diffusion to it.  `nvs_from_renderings` runs the diffusion leg alone on saved renders.
"""
import importlib
import os
import sys

import numpy as np
import torch

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
        if getattr(opts, "renderings", None) is None and not gradio:
            self._attach_reference_geometry()

    # ------------------------------------------------------------------ diffusion leg (this repo)
    def setup_diffusion(self):
        """Reference viewcrafter.py:384-404."""
        torch.manual_seed(self.opts.seed)
        np.random.seed(self.opts.seed)
        root = _reference_root(self.opts)
        if root and root not in sys.path:
            sys.path.append(root)           # lets the YAML's CLIP / Resampler targets resolve to the reference
        assert os.path.exists(self.opts.ckpt_path), "Error: checkpoint Not Found!"
        model = build_diffusion_model(self.opts.config, device=self.device, ckpt_path=self.opts.ckpt_path,
                                      perframe_ae=self.opts.perframe_ae, conditioners="config", init_on_device=False)
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

    def nvs_from_renderings(self, path):
        """Diffusion leg only: `path` holds point-cloud renders [T, H, W, 3] in [0, 1] (.pt or .npy)."""
        r = torch.load(path) if path.endswith(".pt") else torch.from_numpy(np.load(path))
        out = self.run_diffusion(r.float())
        torch.save(out.cpu(), os.path.join(self.opts.save_dir, "diffusion0.pt"))
        # like the reference's nvs_single_view (viewcrafter.py:118-121): write the generated clip as a video as well
        from viewcrafter_amd.utils.video_io import save_video
        save_video((out + 1.0) / 2.0, os.path.join(self.opts.save_dir, "diffusion0.mp4"), fps=10, value_range=(0.0, 1.0))
        return out

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
            def setup_diffusion(self):          # the diffusion model is ours
                self.diffusion, self.noise_shape = ours.diffusion, ours.noise_shape

            def run_diffusion(self, renderings):
                return ours.run_diffusion(renderings)
        self._ref = _Geometry(self.opts, gradio=self.gradio)

    def nvs_single_view(self, gradio=False):
        return self._ref.nvs_single_view(gradio)

    def nvs_sparse_view_interp(self):
        return self._ref.nvs_sparse_view_interp()

    def nvs_single_view_eval(self):
        return self._ref.nvs_single_view_eval()

    def run_gradio(self, *args, **kwargs):
        return self._ref.run_gradio(*args, **kwargs)
