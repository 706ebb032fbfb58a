"""Generate the golden fixtures in this directory by running the REFERENCE code itself.

Run in the build container only (needs /root/reference; the GPU box does not have it):
    python tests/golden/gen_golden.py
The reference (Drexubery/ViewCrafter) is imported unmodified with three module stubs (cv2,
pytorch_lightning, torchvision are absent here and only needed for training/IO), built on tiny
hyper-parameters, loaded with the deterministic synthetic weights of oracle/weights.py and run in fp32 on CPU.
Inputs are regenerated from their names (oracle.weights.synth_input), so the fixtures hold outputs only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("VCX_REFERENCE", "/root/reference")

from oracle.weights import synth_input, synth_state_dict  # noqa: E402
from tests.tiny_config import TINY_DDCONFIG, TINY_RESAMPLER, TINY_UNET, tiny_model_params  # noqa: E402


class AttrDict(dict):
    """Stand-in for OmegaConf: supports cfg['k'], cfg.k, 'k' in cfg, cfg.get."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in o.items()})
        if isinstance(o, (list, tuple)):
            return [AttrDict.wrap(v) for v in o]
        return o


def import_reference():
    """Stub the three absent third-party modules and put the reference on sys.path."""
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(torch.nn.Module):
            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")
        pl.LightningModule = LightningModule
        plu = types.ModuleType("pytorch_lightning.utilities")
        plu.rank_zero_only = lambda f: f
        pl.utilities = plu
        sys.modules["pytorch_lightning"] = pl
        sys.modules["pytorch_lightning.utilities"] = plu
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tv.utils = tvu
        import importlib.machinery
        tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)   # transformers probes find_spec("torchvision")
        tvu.__spec__ = importlib.machinery.ModuleSpec("torchvision.utils", None)
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu
    if REF not in sys.path:
        sys.path.insert(0, REF)


SCHEDULE_BUFFERS = ("betas", "alphas", "sqrt_", "log_one", "posterior", "scale_arr", "lvlb", "logvar")


def load_synth(module, seed=0, skip=()):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(shapes, seed=seed, skip=skip)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected
    return shapes


def gen_schedules(out):
    from lvdm.models import utils_diffusion as ud
    betas = ud.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    out["betas_linear"] = betas
    out["betas_zero_snr"] = ud.rescale_zero_terminal_snr(betas)
    for other in ("cosine", "sqrt_linear", "sqrt"):          # not used by the ViewCrafter YAMLs; part of make_beta_schedule's contract
        out[f"betas_{other}"] = np.asarray(ud.make_beta_schedule(other, 1000, linear_start=0.00085, linear_end=0.012), dtype=np.float64)
    for method, n in (("uniform_trailing", 50), ("uniform_trailing", 5), ("uniform_trailing", 10), ("uniform", 50), ("quad", 20)):
        out[f"ddim_timesteps_{method}_{n}"] = np.asarray(ud.make_ddim_timesteps(method, n, 1000, verbose=False))
    acp = torch.tensor(np.cumprod(1.0 - out["betas_zero_snr"]), dtype=torch.float32)
    ts = out["ddim_timesteps_uniform_trailing_50"]
    for eta in (0.0, 1.0):
        s, a, ap = ud.make_ddim_sampling_parameters(acp, ts, eta, verbose=False)
        out[f"ddim_sigmas_eta{eta}"] = np.asarray(s, dtype=np.float64)
        out[f"ddim_alphas_eta{eta}"] = np.asarray(a, dtype=np.float64)
        out[f"ddim_alphas_prev_eta{eta}"] = np.asarray(ap, dtype=np.float64)
    t = torch.tensor([0, 19, 500, 999])
    out["timestep_embedding_320"] = ud.timestep_embedding(t, 320).numpy()
    out["timestep_embedding_65"] = ud.timestep_embedding(t, 65).numpy()
    a = synth_input("cfg_a", (2, 4, 3, 8, 8))
    b = synth_input("cfg_b", (2, 4, 3, 8, 8))
    out["rescale_noise_cfg"] = ud.rescale_noise_cfg(a, b, guidance_rescale=0.7).numpy()
    # the VAE posterior class (lvdm/distributions.py:24-65)
    from lvdm.distributions import DiagonalGaussianDistribution
    moments, moments2 = synth_input("gauss_moments", (2, 8, 4, 6), scale=3.0), synth_input("gauss_moments2", (2, 8, 4, 6), scale=1.5)
    post, other = DiagonalGaussianDistribution(moments), DiagonalGaussianDistribution(moments2)
    x = post.sample(noise=synth_input("gauss_noise", (2, 4, 4, 6)))
    out["gauss_sample"], out["gauss_kl"], out["gauss_kl_other"], out["gauss_nll"] = x.numpy(), post.kl().numpy(), post.kl(other).numpy(), post.nll(x).numpy()


def gen_unet(out):
    from lvdm.modules.networks.openaimodel3d import UNetModel
    torch.manual_seed(0)
    unet = UNetModel(**TINY_UNET).eval()
    shapes = load_synth(unet)
    out["unet_keys"] = np.array(sorted(shapes.keys()))
    out["unet_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    cd = TINY_UNET["context_dim"]
    with torch.no_grad():
        for tag, (b, t, h, w, L) in {"perframe": (1, 4, 32, 16, 77 + 4 * 16), "shared": (2, 3, 16, 32, 77 + 40)}.items():
            x = synth_input(f"unet_x_{tag}", (b, 8, t, h, w))
            ts = torch.tensor([999, 399][:b])
            ctx = synth_input(f"unet_ctx_{tag}", (b, L, cd))
            fs = torch.tensor([10, 3][:b])
            y = unet(x, ts, context=ctx, fs=fs)
            out[f"unet_out_{tag}"] = y.numpy()
            print("unet", tag, tuple(y.shape), float(y.abs().mean()))


def gen_unet_ssn(out):
    """The same tiny graph with use_scale_shift_norm=True (openaimodel3d.py:221-225: FiLM-style conditioning of every ResBlock) -
    not used by the two shipped YAMLs, accepted by the reference's constructor, so accepted here."""
    from lvdm.modules.networks.openaimodel3d import UNetModel
    torch.manual_seed(0)
    hp = dict(TINY_UNET, use_scale_shift_norm=True)
    unet = UNetModel(**hp).eval()
    shapes = load_synth(unet)
    out["unet_keys"] = np.array(sorted(shapes.keys()))
    out["unet_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        b, t, h, w, L = 2, 3, 16, 32, 77 + 40
        x = synth_input("unet_ssn_x", (b, 8, t, h, w))
        ctx = synth_input("unet_ssn_ctx", (b, L, TINY_UNET["context_dim"]))
        y = unet(x, torch.tensor([999, 399]), context=ctx, fs=torch.tensor([10, 3]))
        out["unet_out"] = y.numpy()
        print("unet ssn", tuple(y.shape), float(y.abs().mean()))


def gen_unet_conv1x1(out):
    """The same tiny graph with use_linear=False (attention.py:266-267, 287-288, 331-336: proj_in / proj_out of both transformer kinds as
    1x1 Conv2d / Conv1d instead of Linear) - not used by the two shipped YAMLs, accepted by the reference's constructor, so accepted here."""
    from lvdm.modules.networks.openaimodel3d import UNetModel
    torch.manual_seed(0)
    unet = UNetModel(**dict(TINY_UNET, use_linear=False)).eval()
    shapes = load_synth(unet)
    out["unet_keys"] = np.array(sorted(shapes.keys()))
    out["unet_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        b, t, h, w, L = 2, 3, 16, 32, 77 + 40
        x = synth_input("unet_c11_x", (b, 8, t, h, w))
        ctx = synth_input("unet_c11_ctx", (b, L, TINY_UNET["context_dim"]))
        y = unet(x, torch.tensor([999, 399]), context=ctx, fs=torch.tensor([10, 3]))
        out["unet_out"] = y.numpy()
        print("unet conv1x1", tuple(y.shape), float(y.abs().mean()))


def _gen_unet_variant(out, tag, **flags):
    from lvdm.modules.networks.openaimodel3d import UNetModel
    torch.manual_seed(0)
    unet = UNetModel(**dict(TINY_UNET, **flags)).eval()
    shapes = load_synth(unet)
    out["unet_keys"] = np.array(sorted(shapes.keys()))
    out["unet_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        b, t, h, w, L = 2, 3, 16, 32, 77 + 40
        x = synth_input(f"unet_{tag}_x", (b, 8, t, h, w))
        ctx = synth_input(f"unet_{tag}_ctx", (b, L, TINY_UNET["context_dim"]))
        y = unet(x, torch.tensor([999, 399]), context=ctx, fs=torch.tensor([10, 3]))
        out["unet_out"] = y.numpy()
        print("unet", tag, flags, tuple(y.shape), float(y.abs().mean()))


def gen_unet_updown(out):
    """resblock_updown=True (openaimodel3d.py:441-451, 529-538: ResBlock(down=True) / ResBlock(up=True) - AvgPool2d / nearest 2x between SiLU and
    the first convolution and on the skip path - in the place of the strided / post-interpolation convolutions); not used by the shipped YAMLs."""
    _gen_unet_variant(out, "ud", resblock_updown=True)


def gen_unet_noconv(out):
    """conv_resample=False (openaimodel3d.py:70-72, 98-103: Downsample = AvgPool2d(2, 2), Upsample = nearest 2x, no parameters); not used by the shipped YAMLs."""
    _gen_unet_variant(out, "nc", conv_resample=False)


def gen_unet_causal(out):
    """use_causal_attention=True (attention.py:343-345, 377-384: a lower-triangular mask over the frames for every TemporalTransformer but init_attn);
    not used by the shipped YAMLs."""
    _gen_unet_variant(out, "ca", use_causal_attention=True)


def gen_unet_relpos(out):
    """use_relative_position=True (attention.py:20-40, 59-62, 104-108, 120-123) in every temporal attention; temporal_length = 2 with 5 frames, so that
    distances beyond +-R are clipped (the ViewCrafter_25 YAML, too, has temporal_length 16 for 25 frames); not used by the shipped YAMLs."""
    from lvdm.modules.networks.openaimodel3d import UNetModel
    torch.manual_seed(0)
    unet = UNetModel(**dict(TINY_UNET, use_relative_position=True, temporal_length=2)).eval()
    shapes = load_synth(unet)
    out["unet_keys"] = np.array(sorted(shapes.keys()))
    out["unet_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        b, t, h, w, L = 1, 5, 16, 16, 77 + 40
        x = synth_input("unet_rp_x", (b, 8, t, h, w))
        ctx = synth_input("unet_rp_ctx", (b, L, TINY_UNET["context_dim"]))
        y = unet(x, torch.tensor([599]), context=ctx, fs=torch.tensor([10]))
        out["unet_out"] = y.numpy()
        print("unet relpos", tuple(y.shape), float(y.abs().mean()))


def adapter_features(b, t, h, w, mc=TINY_UNET["model_channels"], mult=TINY_UNET["channel_mult"]):
    """What a T2I-adapter hands to UNetModel.forward(features_adapter=...): one [(b t), C, h, w] map per level, added behind input
    blocks 2, 5, 8, 11 (openaimodel3d.py:582-588)."""
    return [synth_input(f"adapter_{i}", (b * t, mc * m, h >> i, w >> i), scale=0.5) for i, m in enumerate(mult)]


def gen_unet_adapter(out):
    from lvdm.modules.networks.openaimodel3d import UNetModel
    torch.manual_seed(0)
    unet = UNetModel(**TINY_UNET).eval()
    load_synth(unet)
    with torch.no_grad():
        b, t, h, w, L = 1, 3, 16, 32, 77 + 40
        x = synth_input("unet_ad_x", (b, 8, t, h, w))
        ctx = synth_input("unet_ad_ctx", (b, L, TINY_UNET["context_dim"]))
        y = unet(x, torch.tensor([599]), context=ctx, fs=torch.tensor([10]), features_adapter=adapter_features(b, t, h, w))
        out["unet_out"] = y.numpy()
        print("unet adapter", tuple(y.shape), float(y.abs().mean()))


def gen_vae(out):
    from lvdm.models.autoencoder import AutoencoderKL
    torch.manual_seed(0)
    vae = AutoencoderKL(ddconfig=TINY_DDCONFIG, lossconfig=AttrDict(target="torch.nn.Identity"), embed_dim=4).eval()
    shapes = load_synth(vae)
    out["vae_keys"] = np.array(sorted(shapes.keys()))
    out["vae_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        z = synth_input("vae_z", (2, 4, 8, 16))
        out["vae_decode"] = vae.decode(z).numpy()
        img = synth_input("vae_img", (1, 3, 64, 32), scale=0.5)
        post = vae.encode(img)
        out["vae_encode_moments"] = post.parameters.numpy()
        out["vae_encode_mode"] = post.mode().numpy()
        print("vae", out["vae_decode"].shape, float(np.abs(out["vae_decode"]).mean()))


def gen_ddim(out):
    """Whole hot path through the reference classes: VIPLatentDiffusion + DDIMSampler + decode_first_stage."""
    from lvdm.models.ddpm3d import VIPLatentDiffusion
    from lvdm.models.samplers.ddim import DDIMSampler
    import lvdm.models.samplers.ddim as ddim_mod

    # DDIMSampler.register_buffer hard-codes .to("cuda") (ddim.py:18-22): CPU patch, test side only.
    def register_buffer(self, name, attr):
        setattr(self, name, attr)
    DDIMSampler.register_buffer = register_buffer

    params = AttrDict.wrap(tiny_model_params("lvdm.modules.networks.openaimodel3d.UNetModel",
                                             "lvdm.models.autoencoder.AutoencoderKL"))
    torch.manual_seed(0)
    model = VIPLatentDiffusion(**params).eval()
    shapes = load_synth(model, skip=SCHEDULE_BUFFERS)
    out["model_keys"] = np.array(sorted(shapes.keys()))
    out["model_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                 "scale_arr"):
        out[f"model_{name}"] = getattr(model, name).numpy()
    cd = TINY_UNET["context_dim"]
    b, t, h, w = 1, 4, 32, 16
    cond = {"c_crossattn": [synth_input("ddim_ctx", (b, 77 + 16 * t, cd))], "c_concat": [synth_input("ddim_cat", (b, 4, t, h, w), scale=0.8)]}
    uc = {"c_crossattn": [synth_input("ddim_uctx", (b, 77 + 16 * t, cd))], "c_concat": cond["c_concat"]}
    x_T = synth_input("ddim_xT", (b, 4, t, h, w))
    fs = torch.tensor([10] * b)
    with torch.no_grad():
        v = model.apply_model(x_T, torch.tensor([999]), cond, fs=fs)
        out["apply_model"] = v.numpy()
        for eta in (0.0, 1.0):
            counter = [0]

            def fake_noise(shape, device, repeat=False):
                counter[0] += 1
                return synth_input(f"ddim_noise_{counter[0]}", shape)
            ddim_mod.noise_like = fake_noise
            sampler = DDIMSampler(model)
            samples, inter = sampler.sample(S=5, conditioning=cond, batch_size=b, shape=[4, t, h, w], verbose=False,
                                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=eta,
                                            cfg_img=None, mask=None, x0=None, fs=fs, timestep_spacing="uniform_trailing",
                                            guidance_rescale=0.7, x_T=x_T, log_every_t=1,
                                            unconditional_conditioning_img_nonetext=None)
            out[f"ddim_samples_eta{eta}"] = samples.numpy()
            out[f"ddim_pred_x0_eta{eta}"] = torch.stack(inter["pred_x0"][1:]).numpy()
            if eta == 0.0:
                out["ddim_timesteps"] = np.asarray(sampler.ddim_timesteps)
                out["ddim_scale_arr"] = sampler.ddim_scale_arr.numpy()
                out["ddim_scale_arr_prev"] = sampler.ddim_scale_arr_prev.numpy()
                dec = model.decode_first_stage(samples)
                out["decode_first_stage_sub4"] = dec.numpy()[..., ::4, ::4]  # VAE decode is pinned in full by vae_tiny.npz
            print("ddim eta", eta, float(samples.abs().mean()))
        # multi-condition CFG sampler (--multiple_cond_cfg): third conditioning = ("" text, real image tokens)
        from lvdm.models.samplers.ddim_multiplecond import DDIMSampler as DDIMSamplerMulti
        DDIMSamplerMulti.register_buffer = register_buffer
        uc2 = {"c_crossattn": [torch.cat([uc["c_crossattn"][0][:, :77], cond["c_crossattn"][0][:, 77:]], 1)], "c_concat": cond["c_concat"]}
        sampler = DDIMSamplerMulti(model)
        samples, inter = sampler.sample(S=5, conditioning=cond, batch_size=b, shape=[4, t, h, w], verbose=False,
                                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0,
                                        cfg_img=3.0, mask=None, x0=None, fs=fs, timestep_spacing="uniform_trailing",
                                        guidance_rescale=0.7, x_T=x_T, log_every_t=1,
                                        unconditional_conditioning_img_nonetext=uc2)
        out["multicond_samples"] = samples.numpy()
        out["multicond_pred_x0"] = torch.stack(inter["pred_x0"][1:]).numpy()
        out["multicond_scale_arr_prev"] = sampler.ddim_scale_arr_prev.numpy()
        print("multicond", float(samples.abs().mean()))


def gen_resampler(out):
    """image_proj_model (lvdm/modules/encoders/resampler.py): pure torch, imported unmodified."""
    from lvdm.modules.encoders.resampler import Resampler
    torch.manual_seed(0)
    m = Resampler(**TINY_RESAMPLER).eval()
    shapes = load_synth(m)
    out["resampler_keys"] = np.array(sorted(shapes.keys()))
    out["resampler_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        for tag, (b, n1) in {"a": (2, 17), "b": (1, 40)}.items():
            x = synth_input(f"resampler_x_{tag}", (b, n1, TINY_RESAMPLER["embedding_dim"]))
            y = m(x)
            out[f"resampler_out_{tag}"] = y.numpy()
            print("resampler", tag, tuple(y.shape), float(y.abs().mean()))


def install_clip_stand_ins():
    """`open_clip` and `kornia` are third-party packages that are neither in /root/reference nor in this image; the reference's
    condition.py imports both at module level.  The stand-ins of oracle/clip_oracle.py (open_clip's module tree on
    torch.nn.MultiheadAttention, kornia's blur + bicubic resize) are installed under those names so that the reference's
    OWN embedder code (condition.py:174-240, 302-378) runs unmodified on top of them."""
    from oracle import clip_oracle as co
    oc = types.ModuleType("open_clip")
    oc.create_model_and_transforms = lambda arch, device=None, pretrained=None: (co.StandInCLIP(arch), None, None)

    def tokenize(texts, context_length=77):
        assert all(t == "" for t in texts), "the stand-in only knows the empty prompt"
        return co.tokenize_empty(len(texts), context_length)
    oc.tokenize = tokenize
    sys.modules["open_clip"] = oc
    k = types.ModuleType("kornia")
    k.geometry = types.ModuleType("kornia.geometry")
    k.enhance = types.ModuleType("kornia.enhance")

    def resize(x, size, interpolation="bilinear", align_corners=None, antialias=False):
        assert interpolation == "bicubic" and align_corners is True
        return co.kornia_resize(x, size, antialias)
    k.geometry.resize = resize
    k.enhance.normalize = co.kornia_normalize
    sys.modules["kornia"], sys.modules["kornia.geometry"], sys.modules["kornia.enhance"] = k, k.geometry, k.enhance


CLIP_TINY = "vcx-tiny-test"


def gen_clip(out):
    """The reference's two OpenCLIP embedders (condition.py) on the tiny stand-in towers."""
    install_clip_stand_ins()
    from lvdm.modules.encoders.condition import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2
    torch.manual_seed(0)
    txt = FrozenOpenCLIPEmbedder(arch=CLIP_TINY, device="cpu", freeze=True, layer="penultimate").eval()
    shapes = load_synth(txt)
    out["clip_text_keys"] = np.array(sorted(shapes.keys()))
    out["clip_text_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    img = FrozenOpenCLIPImageEmbedderV2(arch=CLIP_TINY, device="cpu", freeze=True).eval()
    shapes = load_synth(img)
    out["clip_image_keys"] = np.array(sorted(shapes.keys()))
    out["clip_image_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    with torch.no_grad():
        y = txt([""] * 2)
        out["clip_text_empty"] = y.numpy()
        rng = np.random.default_rng(7)
        tokens = torch.zeros(3, 77, dtype=torch.long)
        for i, n in enumerate((5, 20, 75)):
            tokens[i, 0] = 49406
            tokens[i, 1:1 + n] = torch.from_numpy(rng.integers(0, 49406, n))
            tokens[i, 1 + n] = 49407
        out["clip_text_tokens"] = tokens.numpy()
        y = txt.encode_with_transformer(tokens)
        out["clip_text_random"] = y.numpy()
        print("clip text", tuple(y.shape), float(y.abs().mean()))
        for tag, shp in {"down": (2, 3, 320, 448), "up": (1, 3, 96, 64)}.items():
            x = torch.tanh(synth_input(f"clip_image_{tag}", shp))
            y = img(x)
            out[f"clip_image_{tag}"] = y.numpy()
            print("clip image", tag, tuple(y.shape), float(y.abs().mean()))


def gen_igs(out):
    """The driver function itself: reference utils/diffusion_utils.py:117-201 `image_guided_synthesis` on the tiny graph with
    the reference's own embedder / Resampler / sampler / VAE code (cond + uncond assembly, get_latent_z, CFG 7.5, hybrid
    conditioning, n_samples stacking, and the multi-condition variant).  Every Gaussian draw is replaced by a named tensor
    (oracle.weights.NamedRandn); perframe_ae is switched off so that the posterior noise is ONE draw of [(b t), 4, h, w] -
    the per-frame loop of ddpm3d.py:634-639 computes the same thing frame by frame."""
    install_clip_stand_ins()
    from lvdm.models.ddpm3d import VIPLatentDiffusion
    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.samplers.ddim_multiplecond import DDIMSampler as DDIMSamplerMulti
    from utils.diffusion_utils import image_guided_synthesis
    from oracle.weights import NamedRandn
    from tests.tiny_config import IGS_H, IGS_T, IGS_W, igs_model_params

    def register_buffer(self, name, attr):
        setattr(self, name, attr)
    DDIMSampler.register_buffer = register_buffer
    DDIMSamplerMulti.register_buffer = register_buffer
    R = "lvdm.modules.encoders."
    params = AttrDict.wrap(igs_model_params("lvdm.modules.networks.openaimodel3d.UNetModel", "lvdm.models.autoencoder.AutoencoderKL",
                                            R + "condition.FrozenOpenCLIPEmbedder", R + "condition.FrozenOpenCLIPImageEmbedderV2",
                                            R + "resampler.Resampler", device="cpu"))
    torch.manual_seed(0)
    model = VIPLatentDiffusion(**params).eval()
    shapes = load_synth(model, skip=SCHEDULE_BUFFERS)
    model.perframe_ae = False
    out["igs_model_keys"] = np.array(sorted(shapes.keys()))
    out["igs_model_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes.keys())])
    videos = torch.tanh(synth_input("igs_videos", (1, 3, IGS_T, IGS_H, IGS_W)))
    noise_shape = [1, 4, IGS_T, IGS_H // 8, IGS_W // 8]
    real_randn = torch.randn
    for tag, kw in (("cfg", dict(n_samples=2, multiple_cond_cfg=False, cfg_img=None)),
                    ("multicond", dict(n_samples=1, multiple_cond_cfg=True, cfg_img=3.0))):
        torch.randn = NamedRandn(f"igs_{tag}_randn")
        try:
            with torch.no_grad():
                vid = image_guided_synthesis(model, [""], videos, noise_shape, ddim_steps=5, ddim_eta=1.0,
                                             unconditional_guidance_scale=7.5, fs=10, text_input=False,
                                             timestep_spacing="uniform_trailing", guidance_rescale=0.7, condition_index=[0], **kw)
            out[f"igs_{tag}_randn_calls"] = np.asarray(torch.randn.calls)
        finally:
            torch.randn = real_randn
        assert tuple(vid.shape) == (1, kw["n_samples"], 3, IGS_T, IGS_H, IGS_W)
        out[f"igs_{tag}_sub4"] = vid.numpy()[..., ::4, ::4]
        print("igs", tag, tuple(vid.shape), float(vid.abs().mean()), "randn calls", int(out[f"igs_{tag}_randn_calls"]))


def gen_state_dict_full(out):
    """Parameter / buffer names and shapes of the reference's UNetModel, AutoencoderKL and Resampler at the sizes of the two
    shipped YAMLs (built on the meta device: no memory, no arithmetic).  `load_state_dict(strict=True)` of a real checkpoint
    (utils/diffusion_utils.py:88) succeeds exactly when these match."""
    from viewcrafter_amd.config import load_yaml
    from lvdm.models.autoencoder import AutoencoderKL
    from lvdm.modules.encoders.resampler import Resampler
    from lvdm.modules.networks.openaimodel3d import UNetModel
    for cfg in ("inference_pvd_1024.yaml", "inference_pvd_512.yaml"):
        mp = load_yaml(os.path.join(ROOT, "configs", cfg))["model"]["params"]
        with torch.device("meta"):
            mods = {"unet": UNetModel(**dict(mp["unet_config"]["params"])),
                    "vae": AutoencoderKL(**dict(mp["first_stage_config"]["params"])),
                    "resampler": Resampler(**dict(mp["image_proj_stage_config"]["params"]))}
        for name, m in mods.items():
            sd = m.state_dict()
            keys = sorted(sd.keys())
            tag = f"{cfg.split('.')[0]}__{name}"
            out[tag + "__keys"] = np.array(keys)
            out[tag + "__shapes"] = np.array([",".join(str(d) for d in sd[k].shape) for k in keys])
            print(tag, len(keys), "entries,", sum(sd[k].numel() for k in keys) / 1e6, "M elements")


def parser_surface(parser):
    """(flag, type name, repr(default), repr(nargs), action class name) of every option of an argparse parser."""
    rows = []
    for a in parser._actions:
        for opt in a.option_strings:
            if opt.startswith("--") and opt != "--help":
                rows.append((opt, getattr(a.type, "__name__", str(a.type)), repr(a.default), repr(a.nargs), type(a).__name__))
    return sorted(rows)


def gen_cli(out):
    """The command-line surface of `python inference.py` (configs/infer_config.py:7-59), read from the reference's own parser."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_infer_config", os.path.join(REF, "configs", "infer_config.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = parser_surface(mod.get_parser())
    for i, name in enumerate(("flag", "type", "default", "nargs", "action")):
        out[name] = np.array([r[i] for r in rows])
    print("cli:", len(rows), "options")


API_SURFACE = {
    "utils/diffusion_utils.py": ["instantiate_from_config", "get_obj_from_str", "load_model_checkpoint", "image_guided_synthesis",
                                 "get_latent_z", "count_params", "check_istarget", "setup_dist"],
    "lvdm/models/samplers/ddim.py": ["DDIMSampler.__init__", "DDIMSampler.make_schedule", "DDIMSampler.sample", "DDIMSampler.ddim_sampling",
                                     "DDIMSampler.p_sample_ddim", "DDIMSampler.stochastic_encode", "DDIMSampler.decode"],
    "lvdm/models/samplers/ddim_multiplecond.py": ["DDIMSampler.__init__", "DDIMSampler.make_schedule", "DDIMSampler.sample",
                                                  "DDIMSampler.ddim_sampling", "DDIMSampler.p_sample_ddim"],
    "lvdm/modules/networks/openaimodel3d.py": ["UNetModel.__init__", "UNetModel.forward"],
    "lvdm/models/autoencoder.py": ["AutoencoderKL.__init__", "AutoencoderKL.encode", "AutoencoderKL.decode", "AutoencoderKL.forward"],
    "lvdm/models/ddpm3d.py": ["DDPM.__init__", "DDPM.q_sample", "DDPM.predict_start_from_z_and_v", "DDPM.predict_eps_from_z_and_v",
                              "LatentDiffusion.__init__", "LatentDiffusion.apply_model", "LatentDiffusion.decode_first_stage",
                              "LatentDiffusion.encode_first_stage", "LatentDiffusion.get_learned_conditioning", "LatentDiffusion.decode_core",
                              "LatentDiffusion.get_first_stage_encoding", "LatentVisualDiffusion.__init__",
                              "DiffusionWrapper.__init__", "DiffusionWrapper.forward"],
    "lvdm/modules/encoders/resampler.py": ["Resampler.__init__", "Resampler.forward"],
    "lvdm/modules/encoders/condition.py": ["FrozenOpenCLIPEmbedder.__init__", "FrozenOpenCLIPEmbedder.forward", "FrozenOpenCLIPEmbedder.encode",
                                           "FrozenOpenCLIPImageEmbedderV2.__init__", "FrozenOpenCLIPImageEmbedderV2.forward"],
    "viewcrafter.py": ["ViewCrafter.__init__", "ViewCrafter.run_diffusion", "ViewCrafter.setup_diffusion", "ViewCrafter.nvs_single_view",
                       "ViewCrafter.nvs_sparse_view_interp", "ViewCrafter.nvs_single_view_eval"],
    "utils/pvd_utils.py": ["save_video"],
    "lvdm/common.py": ["gather_data", "extract_into_tensor", "noise_like", "default", "exists"],
    "lvdm/models/utils_diffusion.py": ["timestep_embedding", "make_beta_schedule", "make_ddim_timesteps", "make_ddim_sampling_parameters",
                                       "rescale_noise_cfg", "rescale_zero_terminal_snr"],
}


def gen_api(out):
    """Call signatures (parameter order, default expressions, *args / **kwargs) of the functions and methods on the drop-in boundary
    (SURVEY.md section 8b), read from the reference's SOURCE with `ast` - viewcrafter.py and pvd_utils.py cannot be imported here."""
    import ast
    import json
    table = {}
    for rel, wanted in API_SURFACE.items():
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        found = {}

        def visit(node, prefix):
            for n in getattr(node, "body", []):
                if isinstance(n, ast.ClassDef):
                    visit(n, prefix + n.name + ".")
                elif isinstance(n, ast.FunctionDef) and prefix + n.name in wanted:
                    a = n.args
                    pos = [x.arg for x in a.posonlyargs + a.args]
                    dft = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
                    found[prefix + n.name] = {"args": list(zip(pos, dft)), "vararg": bool(a.vararg), "kwarg": bool(a.kwarg)}
        visit(tree, "")
        missing = sorted(set(wanted) - set(found))
        assert not missing, (rel, missing)
        table[rel] = found
    out["json"] = np.array(json.dumps(table, sort_keys=True))
    print("api:", sum(len(v) for v in table.values()), "callables in", len(table), "files")


def gen_yaml(out):
    """The reference's two model YAMLs, parsed (configs/inference_pvd_{1024,512}.yaml), as JSON: what `instantiate_from_config`
    receives when a user points this implementation at an unchanged reference checkout's config."""
    import json
    import yaml
    for name in ("inference_pvd_1024", "inference_pvd_512"):
        out[name] = np.array(json.dumps(yaml.safe_load(open(os.path.join(REF, "configs", name + ".yaml"))), sort_keys=True))
        print(name, len(str(out[name])), "characters")


def main():
    try:      # condition.py imports these; resolve transformers' lazy modules before the torchvision stub confuses its probes
        from transformers import T5Tokenizer, T5EncoderModel, CLIPTokenizer, CLIPTextModel  # noqa: F401
    except Exception as e:  # pragma: no cover
        print("transformers not importable:", e)
    import_reference()
    torch.set_num_threads(8)
    for name, fn in (("schedules", gen_schedules), ("unet_tiny", gen_unet), ("unet_tiny_ssn", gen_unet_ssn), ("unet_tiny_conv1x1", gen_unet_conv1x1), ("unet_tiny_updown", gen_unet_updown), ("unet_tiny_noconv", gen_unet_noconv), ("unet_tiny_causal", gen_unet_causal), ("unet_tiny_relpos", gen_unet_relpos), ("unet_tiny_adapter", gen_unet_adapter), ("vae_tiny", gen_vae), ("ddim_tiny", gen_ddim),
                     ("resampler_tiny", gen_resampler), ("clip_tiny", gen_clip), ("igs_tiny", gen_igs),
                     ("state_dict_full", gen_state_dict_full), ("cli_flags", gen_cli), ("api_signatures", gen_api), ("reference_yaml", gen_yaml)):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
