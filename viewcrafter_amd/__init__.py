"""viewcrafter_amd — MI355X (gfx950) native implementation of ViewCrafter's DDIM denoising hot path.

Hand-written HIP kernels live in csrc/ and are exposed through the C ABI of include/vcx.h
(libvcx.so, loaded with ctypes).  The Python side mirrors the reference's lvdm interface for this
path only (UNetModel, AutoencoderKL, DDIMSampler, image_guided_synthesis).
"""
__version__ = "0.1.0"
