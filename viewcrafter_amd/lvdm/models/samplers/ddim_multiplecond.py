"""Multi-condition CFG DDIM sampler (reference lvdm/models/samplers/ddim_multiplecond.py, `--multiple_cond_cfg`).

Differences from the plain sampler, both kept exactly:
  * three denoiser evaluations per step — (text, image), ("", image), ("", zero image) — combined as
        v = v_uncond + cfg_img * (v_img - v_uncond) + s * (v_cond - v_img)                      (:229-234)
    here run as ONE B=3 UNet forward and combined inside vcx_ddim_step3_f32;
  * `ddim_scale_arr_prev` starts from `ddim_scale_arr[0]`, not `scale_arr[0]` (the reference fixed that "bug" only in
    ddim.py, :33 here vs ddim.py:31-35).
`cfg_img` is a named argument of p_sample_ddim in the reference, so unlike `fs` it does NOT leak into the UNet call;
`unconditional_conditioning_img_nonetext` is read from kwargs and still forwarded (the UNet ignores it).
"""
import torch

from .ddim import DDIMSampler as _DDIMSampler


class DDIMSampler(_DDIMSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        super().make_schedule(ddim_num_steps, ddim_discretize=ddim_discretize, ddim_eta=ddim_eta, verbose=verbose)
        if self.model.use_dynamic_rescale:
            self.ddim_scale_arr_prev = torch.cat([self.ddim_scale_arr[0:1], self.ddim_scale_arr[:-1]])
            self._host["ratio"] = (self.ddim_scale_arr_prev / self.ddim_scale_arr).numpy()

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, uc_type=None,
                      cfg_img=None, mask=None, x0=None, guidance_rescale=0.0, **kwargs):
        """Reference ddim_multiplecond.py:207-291: the same step with `cfg_img` as a named parameter (in the slot where the plain
        sampler has conditional_guidance_scale_temporal)."""
        return super().p_sample_ddim(x, c, t, index, repeat_noise=repeat_noise, use_original_steps=use_original_steps,
                                     quantize_denoised=quantize_denoised, temperature=temperature, noise_dropout=noise_dropout,
                                     score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                     unconditional_guidance_scale=unconditional_guidance_scale,
                                     unconditional_conditioning=unconditional_conditioning, uc_type=uc_type, mask=mask, x0=x0,
                                     guidance_rescale=guidance_rescale, cfg_img=cfg_img, **kwargs)

    def _model_outputs(self, x, t, c, unconditional_conditioning, unconditional_guidance_scale, kwargs):
        cfg_img = kwargs.pop("cfg_img", None)
        if cfg_img is None:
            cfg_img = unconditional_guidance_scale
        uc_img = kwargs["unconditional_conditioning_img_nonetext"]
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            return self.model.apply_model(x, t, c, **kwargs), None, None, 0.0
        if self._batchable(c, unconditional_conditioning, uc_img):
            v_c, v_u, v_i = self._apply_batched(x, t, (c, unconditional_conditioning, uc_img), kwargs)
        else:
            v_c = self.model.apply_model(x, t, c, **kwargs)
            v_u = self.model.apply_model(x, t, unconditional_conditioning, **kwargs)
            v_i = self.model.apply_model(x, t, uc_img, **kwargs)
        return v_c, v_u, v_i, float(cfg_img)
