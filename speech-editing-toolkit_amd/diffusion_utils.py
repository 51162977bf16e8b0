"""Noise schedules (host-side fp64 numpy, init only).

Follows modules/speech_editing/spec_denoiser/diffusion_utils.py:16-45 (`vpsde`, `linear`, `cosine`,
`logsnr` modes of get_noise_schedule_list)."""
import numpy as np


def vpsde_beta_t(t, T, min_beta, max_beta):
    return 1.0 - np.exp(-min_beta / T - 0.5 * (max_beta - min_beta) * (2 * t - 1) / (T ** 2))


def get_noise_schedule_list(schedule_mode, timesteps, min_beta=0.0, max_beta=0.01, s=0.008):
    if schedule_mode == "linear":
        return np.linspace(0.000001, 0.01, timesteps)
    if schedule_mode == "cosine":
        steps = timesteps + 1
        x = np.linspace(0, steps, steps)
        ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    if schedule_mode == "vpsde":
        return np.array([vpsde_beta_t(t, timesteps, min_beta, max_beta) for t in range(1, timesteps + 1)])
    if schedule_mode == "logsnr":
        lo, hi = -20.0, 20.0
        b = np.arctan(np.exp(-0.5 * hi))
        a = np.arctan(np.exp(-0.5 * lo)) - b
        return np.array([-2.0 * np.log(np.tan(a * (t / timesteps) + b)) for t in range(1, timesteps + 1)])
    raise NotImplementedError(schedule_mode)
