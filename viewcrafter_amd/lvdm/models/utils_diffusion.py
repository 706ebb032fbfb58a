"""Noise schedules and sampler tables for the DDIM hot path (host side, numpy/fp64 like the reference's
lvdm/models/utils_diffusion.py; the tables are tiny and computed once per sample() call)."""
import math

import numpy as np
import torch


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """Sinusoidal embedding (reference utils_diffusion.py:8-28).  On the GPU this is one libvcx kernel."""
    from ... import ops
    if repeat_only:
        return timesteps[:, None].float().expand(-1, dim).contiguous()
    return ops.timestep_embedding(timesteps, dim, float(max_period))


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """Reference utils_diffusion.py:31-53.  torch.linspace in fp64 so the tables are bit-identical."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu") ** 2
    elif schedule == "cosine":
        steps = torch.arange(n_timestep + 1, dtype=torch.float64, device="cpu") / n_timestep + cosine_s
        alphas = torch.cos(steps / (1 + cosine_s) * math.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = (1 - alphas[1:] / alphas[:-1]).clamp(0, 0.999)
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64, device="cpu")
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64, device="cpu") ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def rescale_zero_terminal_snr(betas):
    """Zero-terminal-SNR rescale (reference utils_diffusion.py:112-144, arXiv 2305.08891 Alg. 1)."""
    sqrt_ab = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    s0, sT = sqrt_ab[0].copy(), sqrt_ab[-1].copy()
    sqrt_ab = (sqrt_ab - sT) * (s0 / (s0 - sT))
    ab = sqrt_ab ** 2
    alphas = np.concatenate([ab[0:1], ab[1:] / ab[:-1]])
    return 1 - alphas


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """Reference utils_diffusion.py:56-76."""
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        steps_out = np.asarray(list(range(0, num_ddpm_timesteps, stride))) + 1
    elif ddim_discr_method == "uniform_trailing":
        stride = num_ddpm_timesteps / num_ddim_timesteps
        steps_out = np.flip(np.round(np.arange(num_ddpm_timesteps, 0, -stride))).astype(np.int64) - 1
    elif ddim_discr_method == "quad":
        steps_out = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int) + 1
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """Reference utils_diffusion.py:79-91: (sigmas, alphas, alphas_prev) as fp64 numpy arrays."""
    ac = np.asarray(alphacums, dtype=np.float64) if not torch.is_tensor(alphacums) else alphacums.double().cpu().numpy()
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule {sigmas}")
    return sigmas, alphas, alphas_prev


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """Reference utils_diffusion.py:147-158.  Kept for API parity (the sampler fuses it into vcx_ddim_step_f32)."""
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    return guidance_rescale * (noise_cfg * (std_text / std_cfg)) + (1 - guidance_rescale) * noise_cfg
