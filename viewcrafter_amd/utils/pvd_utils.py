"""`utils/pvd_utils.py` of the reference is its point-cloud / camera toolbox (PyTorch3D, DUSt3R: not on this path and left to the
reference checkout).  The one function the diffusion leg's callers use from it is kept here under the same name and contract."""
import os

import numpy as np
import torch

from .video_io import save_video as _save_frames


def save_video(data, images_path, folder=None):
    """Reference utils/pvd_utils.py:38-48: `data` is a [T, H, W, 3] array / tensor with values in [0, 1] (scaled by 255 and
    truncated to uint8, as there), or a list of image file names inside `folder`; written at 8 fps, h264 crf 10 when
    torchvision's writer is available (an uncompressed .avi next to the requested path otherwise).  Returns the path written."""
    if isinstance(data, (list, tuple)):
        from PIL import Image
        folders = folder if isinstance(folder, (list, tuple)) else [folder] * len(data)
        frames = np.stack([np.array(Image.open(os.path.join(f, name) if f else name)) for f, name in zip(folders, data)], 0)
        frames = torch.from_numpy(frames).to(torch.uint8)
    elif isinstance(data, np.ndarray):
        frames = (torch.from_numpy(data) * 255).to(torch.uint8)
    elif isinstance(data, torch.Tensor):
        frames = (data.detach().cpu() * 255).to(torch.uint8)
    else:
        raise TypeError(f"save_video: unsupported data type {type(data).__name__}")
    return _save_frames(frames, images_path, fps=8, value_range=None)
