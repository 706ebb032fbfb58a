"""TEST INFRASTRUCTURE ONLY - second, independent restatement of the image pre-processing in front of the OpenCLIP vision tower.

Reference: lvdm/modules/encoders/condition.py:322-329
    x = kornia.geometry.resize(x, (224, 224), interpolation='bicubic', align_corners=True, antialias=self.antialias)
    x = (x + 1.) / 2.
    x = kornia.enhance.normalize(x, self.mean, self.std)
`kornia` (requirements.txt:8, unpinned) is in neither /root/reference nor this image.  Its published algorithm (kornia/geometry/
transform/affwarp.py `resize`, kornia/filters/gaussian.py `gaussian_blur2d`, kornia/filters/kernels.py `gaussian`), restated here:

  * factors = (h / out_h, w / out_w); anti-aliasing only when max(factors) > 1 (down-scaling);
  * sigma per axis = max((factor - 1) / 2, 0.001)            (the scikit-image rule kornia cites);
  * kernel size per axis = int(max(2 * 2 * sigma, 3)), made odd by adding one;
  * separable Gaussian exp(-x^2 / (2 sigma^2)) normalised to sum 1, x = -(k // 2) .. k // 2, border 'reflect' (mirror WITHOUT
    repeating the edge sample), applied to the whole image;
  * then torch.nn.functional.interpolate(mode='bicubic', align_corners=True): source coordinate o * (in - 1) / (out - 1), Keys cubic
    convolution with A = -0.75 on the four neighbours floor(s) - 1 .. floor(s) + 2, their indices clamped to the image;
  * same-size input: returned unchanged.

oracle/clip_oracle.py::kornia_resize states the same thing with torch's own conv2d / interpolate (that is what wrote
tests/golden/clip_tiny.npz under the reference's embedder code).  THIS file shares no code with it and no code with torch: every
axis is one dense fp64 matrix built element by element from the formulas above (out = R . G . x, R the cubic-convolution matrix,
G the mirrored Gaussian matrix), so an error in either restatement - tap order, border rule, kernel-size rule, A - shows as a
disagreement in tests/test_oracle_golden.py.  It also pins the product kernel vcx_clip_preprocess_f32 (tests/test_kernels_gpu.py).
What it cannot pin is kornia's CODE (absent); the formulas are kornia's documentation and source as published, and the residual
risk is stated in DESIGN.md.  Only tests/ may import this file.
"""
import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _cubic_matrix(n_in, n_out, A=-0.75):
    """[n_out, n_in] fp64: align_corners=True cubic convolution (Keys, a = A) with clamped neighbour indices."""
    R = np.zeros((n_out, n_in), dtype=np.float64)
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    for o in range(n_out):
        s = o * scale
        f = int(np.floor(s))
        t = s - f
        for k in (-1, 0, 1, 2):
            d = abs(t - k)                      # distance of neighbour f + k from the sample point
            if d <= 1.0:
                wgt = (A + 2.0) * d ** 3 - (A + 3.0) * d ** 2 + 1.0
            elif d < 2.0:
                wgt = A * d ** 3 - 5.0 * A * d ** 2 + 8.0 * A * d - 4.0 * A
            else:
                wgt = 0.0
            R[o, min(max(f + k, 0), n_in - 1)] += wgt
    return R


def _gauss_matrix(n, factor):
    """[n, n] fp64: kornia's anti-aliasing blur along one axis as a matrix (mirror border without the edge sample)."""
    sigma = max((factor - 1.0) / 2.0, 0.001)
    ks = int(max(2.0 * 2.0 * sigma, 3))
    if ks % 2 == 0:
        ks += 1
    half = ks // 2
    x = np.arange(ks, dtype=np.float64) - half
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    g /= g.sum()
    G = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        for k in range(ks):
            j = i + k - half
            if j < 0:
                j = -j
            if j > n - 1:
                j = 2 * (n - 1) - j
            G[i, j] += g[k]
    return G


def resize_matrices(h, w, size, antialias=True):
    """(My [size, h], Mx [size, w]) with out = My @ img @ Mx.T for kornia.geometry.resize(img, (size, size), 'bicubic', True, antialias)."""
    if (h, w) == (size, size):
        return np.eye(h), np.eye(w)
    fy, fx = h / size, w / size
    My, Mx = _cubic_matrix(h, size), _cubic_matrix(w, size)
    if antialias and max(fy, fx) > 1:
        My, Mx = My @ _gauss_matrix(h, fy), Mx @ _gauss_matrix(w, fx)
    return My, Mx


def clip_preprocess(x, size=224, antialias=True, mean=CLIP_MEAN, std=CLIP_STD):
    """x [B, C, H, W] in [-1, 1] (any float array) -> fp64 [B, C, size, size]: resize, (x + 1) / 2, (x - mean) / std."""
    x = np.asarray(x, dtype=np.float64)
    My, Mx = resize_matrices(x.shape[-2], x.shape[-1], size, antialias)
    y = np.einsum("oh,bchw,pw->bcop", My, x, Mx, optimize=True)
    y = (y + 1.0) / 2.0
    m = np.asarray(mean, dtype=np.float64).reshape(1, -1, 1, 1)
    s = np.asarray(std, dtype=np.float64).reshape(1, -1, 1, 1)
    return (y - m) / s
