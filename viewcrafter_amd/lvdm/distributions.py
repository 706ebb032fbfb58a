"""DiagonalGaussianDistribution (reference lvdm/distributions.py:24-65), inference subset."""
import torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        # the reference draws the noise on the CPU (distributions.py:35-40)
        if noise is None:
            noise = torch.randn(self.mean.shape)
        return self.mean + self.std * noise.to(device=self.parameters.device)

    def mode(self):
        return self.mean

    # ---- the two statistics the reference class also offers (distributions.py:42-63); not used by inference
    def kl(self, other=None):
        """KL(self || other) summed over (C, H, W); `other` defaults to the standard normal.  A deterministic posterior reports 0."""
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            terms = self.mean.pow(2) + self.var - 1.0 - self.logvar
        else:
            terms = (self.mean - other.mean).pow(2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar
        return 0.5 * terms.sum(dim=[1, 2, 3])

    def nll(self, sample, dims=(1, 2, 3)):
        """Negative log-likelihood of `sample` under the posterior, summed over `dims`."""
        if self.deterministic:
            return torch.Tensor([0.])
        log_2pi = 1.8378770664093453
        return 0.5 * (log_2pi + self.logvar + (sample - self.mean).pow(2) / self.var).sum(dim=list(dims))
