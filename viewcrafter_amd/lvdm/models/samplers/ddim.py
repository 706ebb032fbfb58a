"""DDIM sampler (reference lvdm/models/samplers/ddim.py) driving the gfx950 kernels.

Same public surface (DDIMSampler(model), make_schedule, sample, ddim_sampling, p_sample_ddim, stochastic_encode) and the
same arithmetic; what changes is where it runs:
  * all per-step scalars (a_t, a_prev, sigma_t, sqrt(1-a_t), dynamic-rescale ratio) are Python floats prepared once
    in make_schedule — the reference reads 6 device scalars per step (ddim.py:253-266), each a host sync;
  * classifier-free guidance runs cond and uncond as ONE batched UNet forward (B -> 2B) when both conditionings are
    dicts of tensors, instead of two sequential forwards (ddim.py:223-224);
  * CFG combine + guidance rescale (two per-sample std reductions) + v->eps/x0 + dynamic rescale + the x_{t-1} update
    are one fused launch pair, vcx_ddim_step_f32 (ddim.py:228-279 is ~30 element-wise kernels).
"""
import numpy as np
import torch

from .... import ops
from ....interleave import step_yield
from ...common import noise_like
from ..utils_diffusion import make_ddim_sampling_parameters, make_ddim_timesteps


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0
        self._cfg_cache = None          # p_sample_ddim is public (decode() and bench.py call it without ddim_sampling)
        # classifier-free guidance evaluates the denoiser on the SAME x / t / c_concat under several c_crossattn: the layers
        # ahead of the first cross-attention are computed once (UNetModel._forward, cfg_repeat) - bit-identical results
        self.share_cfg_prefix = True

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor):
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """Reference ddim.py:24-59."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        acp = self.model.alphas_cumprod.detach().float().cpu()
        assert acp.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        if self.model.use_dynamic_rescale:
            scale_arr = self.model.scale_arr.detach().float().cpu()
            self.ddim_scale_arr = scale_arr[self.ddim_timesteps]
            self.ddim_scale_arr_prev = torch.cat([scale_arr[0:1], self.ddim_scale_arr[:-1]])
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(alphacums=acp, ddim_timesteps=self.ddim_timesteps,
                                                                    eta=ddim_eta, verbose=verbose)
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        self.register_buffer("ddim_sigmas", f32(sigmas))
        self.register_buffer("ddim_alphas", f32(alphas))
        self.register_buffer("ddim_alphas_prev", f32(alphas_prev))
        self.register_buffer("ddim_sqrt_one_minus_alphas", f32(np.sqrt(1. - alphas)))
        # host copies of every scalar the loop needs (fp32-rounded like the reference's device tables)
        self._host = dict(
            sigma=np.asarray(sigmas, dtype=np.float32), a=np.asarray(alphas, dtype=np.float32),
            a_prev=np.asarray(alphas_prev, dtype=np.float32),
            sqrt_acp=self.model.sqrt_alphas_cumprod.detach().float().cpu().numpy(),
            sqrt_1m_acp=self.model.sqrt_one_minus_alphas_cumprod.detach().float().cpu().numpy(),
        )
        if self.model.use_dynamic_rescale:
            self._host["ratio"] = (self.ddim_scale_arr_prev / self.ddim_scale_arr).numpy()

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, schedule_verbose=False, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, precision=None, fs=None,
               timestep_spacing="uniform", guidance_rescale=0.0, **kwargs):
        """Reference ddim.py:62-134."""
        if conditioning is not None:
            first = conditioning[list(conditioning.keys())[0]] if isinstance(conditioning, dict) else conditioning
            cbs = (first[0] if isinstance(first, (list, tuple)) else first).shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta, verbose=schedule_verbose)
        if len(shape) == 3:
            size = (batch_size, *shape)
        elif len(shape) == 4:
            size = (batch_size, *shape)
        else:
            raise ValueError(f"shape must be (C,H,W) or (C,T,H,W), got {shape}")
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  precision=precision, fs=fs, guidance_rescale=guidance_rescale, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, verbose=True, precision=None, fs=None, guidance_rescale=0.0,
                      **kwargs):
        """Reference ddim.py:137-205."""
        if ddim_use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError("DDPM-step / quantised / score-corrected sampling is not on the ViewCrafter path")
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        if timesteps is None:
            timesteps = self.ddim_timesteps
        else:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        clean_cond = kwargs.pop("clean_cond", False)
        self._cfg_cache = None
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:   # vestigial in ViewCrafter (mask is always None)
                assert x0 is not None
                img_orig = x0 if clean_cond else self.model.q_sample(x0, ts)
                img = img_orig * mask + (1. - mask) * img
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, mask=mask, x0=x0,
                                              fs=fs, guidance_rescale=guidance_rescale, **kwargs)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            step_yield()           # two clips per GPU (viewcrafter_amd/interleave.py): the other clip's step is queued next; a no-op otherwise
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    # ------------------------------------------------------------------ CFG batching
    @staticmethod
    def _batchable(c, *others):
        """True if all conditionings are dicts of equally shaped tensor lists, i.e. can be stacked on the batch axis."""
        if not isinstance(c, dict):
            return False
        for uc in others:
            if not isinstance(uc, dict) or set(c.keys()) != set(uc.keys()):
                return False
            for k in c:
                if not (isinstance(c[k], (list, tuple)) and isinstance(uc[k], (list, tuple)) and len(c[k]) == len(uc[k])):
                    return False
                if not all(torch.is_tensor(a) and torch.is_tensor(u) and a.shape == u.shape for a, u in zip(c[k], uc[k])):
                    return False
        return True

    def _shares_prefix(self, conds):
        """True if the conditionings differ only in c_crossattn (every other entry is the very same tensor object), the
        denoiser is the native UNet and the optimisation is enabled."""
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        if not self.share_cfg_prefix or not hasattr(unet, "spatial_transformers"):
            return False
        return all(all(a is b_ for a, b_ in zip(conds[0][k], c[k])) for c in conds[1:] for k in c if k != "c_crossattn")

    def _cfg_cond(self, *conds, shared=False):
        """[cond ; uncond (; uncond_img)] stacked on the batch axis, built once per sample() call (so that the UNet's
        context-K/V cache, keyed on tensor identity, hits on every later step).  shared: only c_crossattn is stacked."""
        key = tuple(id(c) for c in conds) + (shared,)
        if self._cfg_cache is None or self._cfg_cache[0] != key:
            both = {k: (list(conds[0][k]) if (shared and k != "c_crossattn") else
                        [torch.cat(parts, dim=0) for parts in zip(*[c[k] for c in conds])]) for k in conds[0]}
            self._cfg_cache = (key, both, conds)
        return self._cfg_cache[1]

    def _apply_batched(self, x, t, conds, kwargs):
        """One UNet forward over len(conds) stacked conditionings; returns the per-conditioning outputs."""
        n, b = len(conds), x.shape[0]
        kw = dict(kwargs)
        if self._shares_prefix(conds):
            out = self.model.apply_model(x, t, self._cfg_cond(*conds, shared=True), cfg_repeat=n, **kw)
        else:
            if torch.is_tensor(kw.get("fs")):
                kw["fs"] = torch.cat([kw["fs"]] * n, 0)
            out = self.model.apply_model(torch.cat([x] * n, 0), torch.cat([t] * n, 0), self._cfg_cond(*conds), **kw)
        return [out[i * b:(i + 1) * b] for i in range(n)]

    def _model_outputs(self, x, t, c, unconditional_conditioning, unconditional_guidance_scale, kwargs):
        """Denoiser evaluations of one step (reference ddim.py:218-231).  Returns (v_cond, v_uncond | None, v_img | None,
        cfg_img)."""
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            return self.model.apply_model(x, t, c, **kwargs), None, None, 0.0
        if self._batchable(c, unconditional_conditioning):
            v_c, v_u = self._apply_batched(x, t, (c, unconditional_conditioning), kwargs)
        elif isinstance(c, (torch.Tensor, dict)):
            v_c = self.model.apply_model(x, t, c, **kwargs)
            v_u = self.model.apply_model(x, t, unconditional_conditioning, **kwargs)
        else:
            raise NotImplementedError
        return v_c, v_u, None, 0.0

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, uc_type=None,
                      conditional_guidance_scale_temporal=None, mask=None, x0=None, guidance_rescale=0.0, **kwargs):
        """Reference ddim.py:208-281."""
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError("not on the ViewCrafter path")
        b, device = x.shape[0], x.device
        x = x.float().contiguous()
        v_c, v_u, v_i, cfg_img = self._model_outputs(x, t, c, unconditional_conditioning, unconditional_guidance_scale, kwargs)
        guided = v_u is not None
        h = self._host
        step_t = int(self.ddim_timesteps[index])   # == t[i] for all i, by construction of ddim_sampling
        sigma = float(h["sigma"][index])
        coef = [float(h["sqrt_acp"][step_t]), float(h["sqrt_1m_acp"][step_t]), float(h["a_prev"][index]), sigma,
                float(h["ratio"][index]) if self.model.use_dynamic_rescale else 1.0, float(unconditional_guidance_scale),
                float(guidance_rescale) if guided else 0.0, 1.0 if self.model.parameterization == "v" else 0.0]
        if self.model.parameterization != "v":
            # eps-parameterisation reads the DDIM tables instead (ddim.py:259-260)
            coef[0], coef[1] = float(np.sqrt(h["a"][index])), float(np.sqrt(1. - h["a"][index]))
        # the reference draws the noise unconditionally (ddim.py:275), also when sigma_t = 0 (eta = 0): draw it too, so that a
        # fixed seed leaves the generator in the same state (the x_T of a following sample() call); the kernel skips it
        noise = noise_like(x.shape, device, repeat_noise)
        noise = noise * temperature if sigma != 0.0 else None
        x_prev, pred_x0 = ops.ddim_step(x, v_c.contiguous(), v_u.contiguous() if v_u is not None else None, noise, coef,
                                        v_img=v_i.contiguous() if v_i is not None else None, cfg_img=cfg_img)
        return x_prev, pred_x0

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """Reference ddim.py:284-303: denoise `x_latent` (a latent noised to schedule index t_start, e.g. by stochastic_encode)
        through the first t_start DDIM steps.  Needs make_schedule() first, like the reference; extra sampler options
        (fs, guidance_rescale) are not part of this signature there either."""
        if use_original_steps:
            raise NotImplementedError("DDPM-step decoding is not on the ViewCrafter path")
        steps = np.asarray(self.ddim_timesteps)[:t_start]
        x = x_latent
        for done, index in enumerate(range(len(steps) - 1, -1, -1)):
            ts = torch.full((x.shape[0],), int(steps[index]), device=x.device, dtype=torch.long)
            x, _ = self.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=unconditional_guidance_scale,
                                      unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(done)
        return x

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """Reference ddim.py:306-319."""
        if use_original_steps:
            sa, s1 = self.model.sqrt_alphas_cumprod, self.model.sqrt_one_minus_alphas_cumprod
        else:
            sa, s1 = torch.sqrt(self.ddim_alphas), self.ddim_sqrt_one_minus_alphas
        if noise is None:
            noise = torch.randn_like(x0)
        shape = (x0.shape[0],) + (1,) * (x0.dim() - 1)
        return sa.gather(-1, t).reshape(shape) * x0 + s1.gather(-1, t).reshape(shape) * noise
