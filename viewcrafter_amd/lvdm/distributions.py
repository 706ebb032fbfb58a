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
