"""Multi-condition CFG DDIM sampler (reference lvdm/models/samplers/ddim_multiplecond.py, `--multiple_cond_cfg`).

SURVEY.md §8(f) ranks it as a follow-up row: it adds a third UNet forward per step with
    e = e_uncond + cfg_img * (e_img - e_uncond) + s * (e_cond - e_img)           (:229-234)
and uses the un-fixed `ddim_scale_arr_prev` (:33).  Not built yet — fail loudly rather than fall back.
"""


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        raise NotImplementedError("--multiple_cond_cfg (3-way CFG sampler) is a planned follow-up row (SURVEY.md §8f.2); "
                                  "use the default single-condition CFG sampler")
