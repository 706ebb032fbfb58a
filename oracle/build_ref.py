"""TEST INFRASTRUCTURE ONLY - recipe for oracle/_ref/: the REFERENCE's own denoiser / VAE code as the CPU baseline.

The reference is Python; its source stays where it lies (/root/reference, read-only) and is never copied into this repo.
What this script writes - and only under oracle/_ref/, which is git-ignored but travels to the GPU box like the built
libvcx.so - is CPython BYTECODE compiled from those sources (`py_compile` with an explicit output path: a "binary" in the
sense of the build rules, sourceless-importable as lvdm.* / utils.*).  With it, bench.py's `cpu_baseline` leg times the
reference implementation itself on the GPU box's host cores (`"kind": "reference"`), and tests/test_oracle_golden.py
cross-checks the oracle restatement against it at run time on top of the committed golden vectors.

    python oracle/build_ref.py            (also run by __graft_entry__.build() when /root/reference is present)

Modules compiled: the inference path only - lvdm/{basics,common,distributions,ema}.py, lvdm/models/{autoencoder,ddpm3d,
utils_diffusion}.py, lvdm/models/samplers/{ddim,ddim_multiplecond}.py, lvdm/modules/{attention}.py, lvdm/modules/networks/
{openaimodel3d,ae_modules}.py, lvdm/modules/encoders/resampler.py and utils/diffusion_utils.py.
"""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("VCX_REFERENCE", "/root/reference")
MODULES = [
    "lvdm/basics.py", "lvdm/common.py", "lvdm/distributions.py", "lvdm/ema.py",
    "lvdm/models/autoencoder.py", "lvdm/models/ddpm3d.py", "lvdm/models/utils_diffusion.py",
    "lvdm/models/samplers/ddim.py", "lvdm/models/samplers/ddim_multiplecond.py",
    "lvdm/modules/attention.py", "lvdm/modules/networks/openaimodel3d.py", "lvdm/modules/networks/ae_modules.py",
    "lvdm/modules/encoders/resampler.py", "utils/diffusion_utils.py",
]


def build(ref=REF, out=OUT, quiet=False):
    if not os.path.isdir(ref):
        if not quiet:
            print(f"[build_ref] {ref} not present: oracle/_ref is only (re)built in the build container; keeping what is there")
        return False
    tag = f"cpython-{sys.version_info.major}{sys.version_info.minor}"
    final, out = out, out + ".tmp"           # compiled aside and swapped in at the end: a compile error leaves the previous tree alone
    if os.path.isdir(out):
        shutil.rmtree(out)
    packages = set()
    for rel in MODULES:
        src = os.path.join(ref, rel)
        dst = os.path.join(out, rel[:-3] + ".pyc")          # legacy (sourceless) layout: pkg/mod.pyc next to where mod.py would be
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(src, cfile=dst, dfile="reference:" + rel, doraise=True, optimize=0)
        d = os.path.dirname(rel)
        while d:
            packages.add(d)
            d = os.path.dirname(d)
    for pkg in sorted(packages):            # packages: the reference's own __init__.py where it has one, an empty module otherwise
        init_src = os.path.join(ref, pkg, "__init__.py")
        dst = os.path.join(out, pkg, "__init__.pyc")
        if os.path.exists(init_src):
            py_compile.compile(init_src, cfile=dst, dfile="reference:" + pkg + "/__init__.py", doraise=True, optimize=0)
        else:
            empty = os.path.join(out, pkg, "__empty__.py")
            open(empty, "w").close()
            py_compile.compile(empty, cfile=dst, dfile="generated:empty", doraise=True, optimize=0)
            os.remove(empty)
    with open(os.path.join(out, "BUILD_INFO"), "w") as f:
        f.write(f"bytecode ({tag}) of {len(MODULES)} reference modules compiled from {ref} by oracle/build_ref.py; no sources here\n")
    if os.path.isdir(final):
        shutil.rmtree(final)
    os.rename(out, final)
    if not quiet:
        print(f"[build_ref] wrote {len(MODULES)} modules + {len(packages)} packages under {final}")
    return True


if __name__ == "__main__":
    sys.exit(0 if build() else 1)
