"""Config container for the reference's YAML model graphs without OmegaConf.

The reference passes sub-configs as objects into constructors and reads them both as mappings
(`cfg["target"]`, `"target" in cfg`, `cfg.get("params")`, utils/diffusion_utils.py:31-38) and as attributes
(`unet_config.params.temporal_length`, lvdm/models/ddpm3d.py:81), so the container supports both.
"""
import yaml


class Config(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = value

    @classmethod
    def wrap(cls, obj):
        if isinstance(obj, dict):
            return cls({k: cls.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return [cls.wrap(v) for v in obj]
        return obj


def load_yaml(path):
    """OmegaConf.load(path) equivalent (viewcrafter.py:387)."""
    with open(path, "r") as f:
        return Config.wrap(yaml.safe_load(f))
