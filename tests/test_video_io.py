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


def test_pvd_utils_save_video_keeps_the_reference_contract(tmp_path):
    """utils/pvd_utils.py:38-48 of the reference: data in [0, 1] -> (x * 255) truncated to uint8, 8 fps; arrays, tensors and a
    list of image files inside `folder` are accepted."""
    import numpy as np
    import torch
    from PIL import Image
    from viewcrafter_amd.utils.pvd_utils import save_video
    from viewcrafter_amd.utils.video_io import read_avi
    g = torch.Generator().manual_seed(3)
    data = torch.rand(5, 16, 24, 3, generator=g)
    want = (data * 255).to(torch.uint8).numpy()
    for tag, d in (("t", data), ("n", data.numpy())):
        out = save_video(d, str(tmp_path / f"{tag}.mp4"))
        if out.endswith(".avi"):             # no torchvision writer in this image: lossless fallback, checked bit for bit
            frames, fps = read_avi(out)
            assert fps == 8 and np.array_equal(frames, want)
    names = []
    for i in range(3):
        Image.fromarray(want[i]).save(tmp_path / f"f{i}.png")
        names.append(f"f{i}.png")
    out = save_video(names, str(tmp_path / "l.mp4"), folder=str(tmp_path))
    if out.endswith(".avi"):
        frames, fps = read_avi(out)
        assert fps == 8 and np.array_equal(frames, want[:3])
