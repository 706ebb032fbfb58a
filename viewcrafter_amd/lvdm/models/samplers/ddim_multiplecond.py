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
