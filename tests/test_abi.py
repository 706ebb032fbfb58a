"""The C-ABI library loads on a GPU-less host and exports every function include/vcx.h declares (no compute calls)."""
import ctypes
import os
import re

from viewcrafter_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vcx_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 20
    L = _lib.lib()
    for n in names:
        assert getattr(L, n) is not None, n
    assert sorted(_lib.SYMBOLS) == names, "viewcrafter_amd/_lib.py SYMBOLS must list exactly the header's functions"


def test_abi_version_and_error_string():
    L = _lib.lib()
    src = open(os.path.join(ROOT, "include", "vcx.h")).read()
    assert L.vcx_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define VCX_ABI_VERSION (\d+)", src).group(1))
    assert isinstance(L.vcx_last_error(), bytes)


def test_tune_knobs_default_to_the_product_and_round_trip():
    """include/vcx.h VCX_TUNE_*: the Python table mirrors the header's indices, every knob starts at its default (no VCX_TUNE_*
    in the test environment), set returns the previous value, an unknown knob is rejected."""
    L = _lib.lib()
    src = open(os.path.join(ROOT, "include", "vcx.h")).read()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define VCX_TUNE_([A-Z0-9_]+) (\d+)", src)}
    count = header.pop("COUNT")
    assert {k: v[0] for k, v in _lib.TUNE.items()} == header and count == len(header)
    for name, (idx, dflt) in _lib.TUNE.items():
        if "VCX_TUNE_" + name in os.environ:
            continue
        assert L.vcx_tune_get(idx) == dflt, name
        assert L.vcx_tune_set(idx, 7) == dflt and L.vcx_tune_get(idx) == 7
        assert L.vcx_tune_set(idx, dflt) == 7
    assert L.vcx_tune_set(99, 1) == -1 and b"knob" in L.vcx_last_error()


def header_gemm_desc_fields():
    """(name, ctypes type) of every field of struct vcx_gemm_desc, parsed from include/vcx.h."""
    src = open(os.path.join(ROOT, "include", "vcx.h")).read()
    body = re.search(r"typedef struct vcx_gemm_desc \{(.*?)\} vcx_gemm_desc;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    want = []
    ctypes_of = {"int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "float": ctypes.c_float, "size_t": ctypes.c_size_t}
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].rsplit(None, 1)
        ctype, names = first[0], [first[1]] + [n.strip() for n in names[1:]]
        for n in names:
            ptr = "*" in ctype or n.startswith("*")
            want.append((n.lstrip("*"), ctypes.c_void_p if ptr else ctypes_of[ctype]))
    return want


def test_gemm_desc_layout_matches_header():
    """The ctypes mirror follows the C struct of include/vcx.h field by field (names and order parsed from the header; C types
    mapped to ctypes) and in total size: size_t, 12 pointers, four int64, 24 int32, two floats - 240 bytes, no padding."""
    want = header_gemm_desc_fields()
    assert [(f[0], f[1]) for f in _lib.GemmDesc._fields_] == want
    assert ctypes.sizeof(_lib.GemmDesc) == 8 + 12 * 8 + 4 * 8 + 24 * 4 + 2 * 4 == 240
    assert _lib.GemmDesc().struct_size == 240 and _lib.GemmDesc(M=3).struct_size == 240


def test_integration_md_stub_matches_header():
    """The binding stub a maintainer copies out of INTEGRATION.md is executed as written and compared with the header - the doc
    cannot go stale again (round 3 shipped a 144-byte ABI-1 stub next to a 168-byte struct)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes, torch\n(.*?)```", md, re.S).group(1)
    cls = re.search(r"(class GemmDesc\(ctypes\.Structure\):.*?)\nvcx\.vcx_gemm_f16\.argtypes", block, re.S).group(1)
    ns = {"ctypes": ctypes}
    exec(cls, ns)
    stub = ns["GemmDesc"]
    assert [(f[0], f[1]) for f in stub._fields_] == header_gemm_desc_fields()
    assert ctypes.sizeof(stub) == ctypes.sizeof(_lib.GemmDesc)
    assert "struct_size=ctypes.sizeof(GemmDesc)" in block, "the stub's call must fill in struct_size"
    assert f"ABI {_lib.ABI_VERSION}" in cls


def test_gemm_rejects_a_descriptor_of_another_size():
    """ABI 5: a descriptor whose struct_size is not sizeof(vcx_gemm_desc) - an old binding's 168-byte struct starts with the A
    pointer there - is refused before any field is trusted."""
    L = _lib.lib()
    d = _lib.GemmDesc()
    d.struct_size = 168
    assert L.vcx_gemm_f16(ctypes.byref(d), None) == -1 and b"struct_size" in L.vcx_last_error()
    d.struct_size = 0
    assert L.vcx_gemm_f16(ctypes.byref(d), None) == -1 and b"struct_size" in L.vcx_last_error()


def test_argument_validation_without_gpu():
    """Validation happens before any launch, so bad descriptors are rejected even on a GPU-less host."""
    L = _lib.lib()
    d = _lib.GemmDesc()
    assert L.vcx_gemm_f16(ctypes.byref(d), None) == -1
    assert b"null" in L.vcx_last_error()
    assert L.vcx_layernorm_f16(None, None, None, None, 4, 64, 1e-5, None) == -1


def test_flash_v2_listing_passes_the_static_audit():
    """tools/isa_audit.py: the hand-scheduled attention kernel compiles to gfx950 with its 132 asm-owned AGPRs, no scratch,
    no compiler-generated AGPR traffic, and none of the hazards hipcc does not pad inside asm statements."""
    import importlib.util
    import shutil
    import pytest
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not (os.path.exists(mod.HIPCC) or shutil.which(mod.HIPCC)):
        pytest.skip("hipcc not available")
    results = mod.audit_all(mod.compile_listing())
    assert results, "no flash2 kernel in the listing"
    for kernel, problems, summary in results:
        assert not problems, (kernel, problems)
        assert summary["agpr_count"] == 132 and summary["mfma"] in (16 + 3 * 40 + 2 * 24, 16 + 3 * 32 + 2 * 16)      # prologue, 3 full steps, 2 tails


def test_no_kernel_overwrites_the_data_registers_of_a_wide_store_right_behind_it():
    """tools/isa_audit.py --stores: every kernel source of libvcx compiled to gfx950 assembly (no GPU needed) and scanned for a VALU
    write to the data registers of a >= 96-bit LDS / memory store within two issue slots.  hipcc emitted that once (the LNFOLD_T
    epilogue's second strip vector into the registers its first ds_write_b128 was still reading) and the GPU suite caught it as a
    run-dependent error in single output columns; this keeps it from coming back anywhere."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit_library_store_hazards() == []
    # the scanner itself: the pattern that was emitted, and the same store with a slot of distance
    bad = "_Zk:\n\tds_write_b128 v141, v[150:153]\n\tv_pk_mul_f32 v[150:151], v[150:151], v[162:163]\n"
    ok = "_Zk:\n\tds_write_b128 v141, v[150:153]\n\ts_nop 1\n\tv_pk_mul_f32 v[150:151], v[150:151], v[162:163]\n"
    assert len(mod.store_data_hazards(bad)) == 1 and mod.store_data_hazards(ok) == []
    # ... and the packed fma form that was not bit-reproducible (LNFOLD_T epilogue, 128x128 tile): low result from a high source half
    assert len(mod.store_data_hazards("_Zk:\n\tv_pk_fma_f32 v[60:61], v[60:61], v[4:5], v[0:1] op_sel:[0,1,1]\n")) == 1
    assert mod.store_data_hazards("_Zk:\n\tv_pk_fma_f32 v[60:61], v[60:61], v[4:5], v[0:1] op_sel_hi:[1,0,0]\n") == []


def test_product_build_reads_no_scratch_knob_on_a_launch_path():
    """VCX_TUNE_EXP0 / EXP1 are "free for one-off experiments" (vcx.h) - a default build must not consult them, or an
    unrelated experiment that sets one changes (or, as in round 3, breaks) a production dispatch.  Every read in csrc/ has to sit
    inside a preprocessor region that only an ablation / experiment build enables."""
    csrc = os.path.join(ROOT, "viewcrafter_amd", "csrc")
    offenders = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        stack = []                                   # (macro, active_in_default_build)
        for ln, line in enumerate(open(os.path.join(csrc, name)), 1):
            t = line.strip()
            m = re.match(r"#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(.*)", t)
            if m:
                kind, rest = m.group(1), m.group(2).split("//")[0].strip()
                if kind in ("ifdef", "if"):
                    stack.append((rest, not re.search(r"VCX_\w*(ABLATION|EXPERIMENT)\w*", rest)))
                elif kind == "ifndef":
                    stack.append((rest, True))
                elif kind == "else" and stack:
                    macro, active = stack.pop()
                    was_ifndef_of_ablation = bool(re.search(r"VCX_\w*(ABLATION|EXPERIMENT)\w*", macro)) and active
                    stack.append((macro, (not active) and not was_ifndef_of_ablation))
                elif kind == "endif" and stack:
                    stack.pop()
                continue
            if re.search(r"vcx_tune\(\s*VCX_TUNE_EXP[01]\s*\)", t) and all(active for _, active in stack):
                offenders.append(f"{name}:{ln}: {t}")
    assert not offenders, offenders


def test_ddim_workspace_contract():
    """ABI 6: every entry point that takes a workspace has a size query and is TOLD the size it gets.  The DDIM step
    refuses a workspace smaller than vcx_ddim_ws_bytes(B, n) - checked ahead of any launch, so this runs without a GPU."""
    import ctypes
    from viewcrafter_amd import _lib
    L = _lib.lib()
    n = 4 * 25 * 72 * 128
    need = L.vcx_ddim_ws_bytes(2, n)
    assert 0 < need <= 8192 * 2 and need % 8 == 0 and L.vcx_ddim_ws_bytes(0, n) == 0
    assert L.vcx_ddim_ws_bytes(1, 100) == 32            # one block, four fp64 partial sums
    coef = (ctypes.c_float * 9)(*([0.5] * 9))
    fake = 1 << 20                                      # never dereferenced: the size check comes first
    rc = L.vcx_ddim_step3_f32(fake, fake, fake, None, None, fake, fake, fake, need - 8, 2, n, coef, None)
    assert rc == -1 and b"vcx_ddim_ws_bytes" in L.vcx_last_error()          # VCX_EINVAL
