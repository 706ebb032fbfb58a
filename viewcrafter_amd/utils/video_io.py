"""Video output for the diffusion leg (reference: utils/pvd_utils.py:38-48 `save_video`, which needs torchvision.io / PyAV).

`save_video` keeps the reference's signature and value convention (frames [T, H, W, 3] float, `value_range`), uses
torchvision.io.write_video (h264) when that stack is installed, and otherwise writes an uncompressed AVI (RIFF / 'DIB '
24-bit BGR, bottom-up rows) with nothing but numpy - playable by ffmpeg / VLC, so that the CLI works on a machine that
has neither torchvision nor PyAV.  `read_avi` reads that container back (tests, and tools that post-process the frames).
"""
import os
import struct

import numpy as np
import torch


def _to_uint8(data, value_range):
    """[T, H, W, 3] float in value_range -> uint8, like the reference ((x - lo) / (hi - lo) * 255, clamped)."""
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(data)
    if data.dtype == torch.uint8 or value_range is None:        # already pixel values
        return data.detach().cpu().to(torch.uint8).numpy()
    data = data.detach().float().cpu()
    lo, hi = value_range
    data = ((data - lo) / (hi - lo)).clamp(0, 1) * 255.0
    return data.round().to(torch.uint8).numpy()


def _chunk(tag, payload):
    pad = b"\0" if len(payload) & 1 else b""
    return tag + struct.pack("<I", len(payload)) + payload + pad


def write_avi(frames_u8, path, fps=10):
    """frames_u8 [T, H, W, 3] uint8 RGB -> uncompressed AVI."""
    frames_u8 = np.ascontiguousarray(frames_u8)
    assert frames_u8.ndim == 4 and frames_u8.shape[-1] == 3 and frames_u8.dtype == np.uint8
    T, H, W, _ = frames_u8.shape
    stride = (W * 3 + 3) & ~3                                  # DIB rows are padded to 4 bytes
    frame_bytes = stride * H
    avih = struct.pack("<IIIIIIIIII4I", int(1e6 / fps), frame_bytes * fps, 0, 0x10, T, 0, 1, frame_bytes, W, H, 0, 0, 0, 0)
    strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"DIB ", 0, 0, 0, 0, 1, fps, 0, T, frame_bytes, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, frame_bytes, 0, 0, 0, 0)
    strl = b"LIST" + struct.pack("<I", 4 + len(_chunk(b"strh", strh)) + len(_chunk(b"strf", strf))) + b"strl" + \
        _chunk(b"strh", strh) + _chunk(b"strf", strf)
    hdrl_body = b"hdrl" + _chunk(b"avih", avih) + strl
    hdrl = b"LIST" + struct.pack("<I", len(hdrl_body)) + hdrl_body
    movi_chunks, index = [], []
    offset = 4
    row = np.zeros((H, stride), dtype=np.uint8)
    for t in range(T):
        bgr = frames_u8[t, ::-1, :, ::-1].reshape(H, W * 3)   # bottom-up, BGR
        row[:, :W * 3] = bgr
        payload = row.tobytes()
        movi_chunks.append(_chunk(b"00db", payload))
        index.append(struct.pack("<4sIII", b"00db", 0x10, offset, len(payload)))
        offset += len(movi_chunks[-1])
    movi_body = b"movi" + b"".join(movi_chunks)
    movi = b"LIST" + struct.pack("<I", len(movi_body)) + movi_body
    idx1 = _chunk(b"idx1", b"".join(index))
    riff_body = b"AVI " + hdrl + movi + idx1
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(riff_body)) + riff_body)


def read_avi(path):
    """Inverse of write_avi: -> (frames [T, H, W, 3] uint8 RGB, fps)."""
    buf = open(path, "rb").read()
    assert buf[:4] == b"RIFF" and buf[8:12] == b"AVI "
    i = buf.index(b"avih") + 8
    usec, _, _, _, T, _, _, _, W, H = struct.unpack("<10I", buf[i:i + 40])
    stride = (W * 3 + 3) & ~3
    frames = np.empty((T, H, W, 3), dtype=np.uint8)
    pos = buf.index(b"movi") + 4
    for t in range(T):
        assert buf[pos:pos + 4] == b"00db"
        n = struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        rows = np.frombuffer(buf, dtype=np.uint8, count=n, offset=pos + 8).reshape(H, stride)[:, :W * 3].reshape(H, W, 3)
        frames[t] = rows[::-1, :, ::-1]
        pos += 8 + n + (n & 1)
    return frames, int(round(1e6 / usec))


def save_video(data, images_path, folder=None, fps=10, value_range=(-1.0, 1.0)):
    """Reference utils/pvd_utils.py:38-48.  data: [T, H, W, 3] (or a list of image paths when `folder` is given, as in
    the reference).  Writes h264 through torchvision when available, an uncompressed .avi next to the requested path
    otherwise; returns the path actually written."""
    if isinstance(data, (list, tuple)) or folder is not None:
        raise NotImplementedError("save_video from image files needs an image decoder; pass a frame tensor")
    frames = _to_uint8(data, value_range)
    os.makedirs(os.path.dirname(os.path.abspath(images_path)), exist_ok=True)
    try:
        import torchvision.io as tvio   # noqa: F401
        tvio.write_video(images_path, torch.from_numpy(frames), fps=fps, video_codec="h264", options={"crf": "10"})
        return images_path
    except Exception:
        out = os.path.splitext(images_path)[0] + ".avi"
        write_avi(frames, out, fps=fps)
        return out
