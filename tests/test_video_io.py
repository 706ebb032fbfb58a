"""utils/video_io.py: the dependency-free AVI writer behind `save_video` (reference utils/pvd_utils.py:38-48)."""
import numpy as np
import torch

from viewcrafter_amd.utils import video_io


def test_avi_round_trip(tmp_path):
    g = np.random.default_rng(0)
    frames = g.integers(0, 256, size=(5, 18, 33, 3), dtype=np.uint8)     # odd width: exercises the 4-byte row padding
    p = str(tmp_path / "x.avi")
    video_io.write_avi(frames, p, fps=8)
    back, fps = video_io.read_avi(p)
    assert fps == 8
    assert np.array_equal(back, frames)


def test_save_video_value_range(tmp_path):
    x = torch.linspace(-1.2, 1.2, 4 * 6 * 8 * 3).reshape(4, 6, 8, 3)
    out = video_io.save_video(x, str(tmp_path / "v.mp4"), fps=10)
    if out.endswith(".avi"):
        back, _ = video_io.read_avi(out)
        ref = (((x + 1) / 2).clamp(0, 1) * 255).round().to(torch.uint8).numpy()
        assert np.array_equal(back, ref)
