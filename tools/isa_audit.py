"""Static audit of the hand-scheduled flash-attention kernel (viewcrafter_amd/csrc/attention_v2.hip).

The kernel owns the accumulator half of the register file by NAME inside asm statements and issues every MFMA and every
softmax VALU operation as asm, so hipcc neither allocates those registers nor pads the hazards of those instructions.  This
script compiles the file to gfx950 assembly (device side only, a few seconds, no GPU) and checks what the compiler cannot:

  1. resources: no scratch, no VGPR spills, exactly the 132 asm-owned AGPRs (a[0:63] O^T, a[64:95] Q, a[96:127] row sums, a[128:131] ones), arch VGPRs <= 256;
  2. the compiler itself never touches an AGPR: no v_accvgpr_* and no a-register operand outside ;;#ASMSTART / ;;#ASMEND;
  3. every v_mfma sits inside an asm statement (no builtin MFMA whose register form the allocator would choose);
  4. MFMA result -> VALU: no non-MFMA instruction reads or writes a VGPR of an MFMA destination tuple within MIN_MFMA_GAP
     wait states behind that MFMA (8-pass XDL write -> VALU read needs 11-12; an instruction is 1, s_nop N is N + 1);
  5. VALU write -> v_permlane32_swap of the same register: at least 2 instructions in between;
  6. transcendental forwarding: the instruction right behind a v_exp_f32 does not read its result;
  7. VALU write -> MFMA source (A / B / C in VGPRs): at least 2 wait states (instructions or s_nop states) in between.
Checks 4-7 walk the listing linearly (basic blocks in layout order), which is how the hot loop executes.

    python tools/isa_audit.py            # exit code 0 = clean; prints a per-check summary
    python tools/isa_audit.py --stores   # library-wide: no VALU write to the data registers of a wide store right behind it
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "viewcrafter_amd", "csrc", "attention_v2.hip")
KERNELS = ("flash2_d64_kernelILi0ELi0E", "flash2_d64_kernelILi0ELi1E")      # <no ablation, row sums by MFMA | by VALU, any stream variant>
MIN_MFMA_GAP = 16
N_AGPR = 132
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only", "-fno-slp-vectorize",
         "-S", "--cuda-device-only"]


def compile_listing(path=None):
    out = path or os.path.join(tempfile.mkdtemp(prefix="vcx_isa_"), "attention_v2.s")
    r = subprocess.run([HIPCC] + FLAGS + ["-o", out, SRC], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-3000:])
    return open(out).read()


def vregs(tok, prefix="v"):
    """'v[12:15]' -> {12..15}, 'v7' -> {7}; anything else -> {}."""
    m = re.fullmatch(prefix + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(prefix + r"(\d+)", tok)
    return {int(m.group(1))} if m else set()


def parse(listing, kernel):
    body = listing[listing.index(next(l for l in listing.split("\n") if l.startswith("_Z") and kernel in l.split(":")[0] and ":" in l)):]
    body = body[:body.index(".Lfunc_end")]
    ins, in_asm = [], False
    for raw in body.split("\n")[1:]:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        ins.append(dict(op=op, ops=ops, asm=in_asm, text=t))
    md = listing[listing.index("amdhsa.kernels"):]
    at = md.index(kernel)                 # one YAML entry per kernel, keys in alphabetical order: .agpr_count ... .name ... .vgpr_count
    start = md.rindex("- .agpr_count", 0, at)
    end = md.find("- .agpr_count", at)
    md = md[start:end if end > 0 else len(md)]
    meta = {k: int(v) for k, v in re.findall(r"\.(agpr_count|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", md)}
    return ins, meta


def writes_reads(i):
    """(written VGPR set, read VGPR set) of one instruction, by mnemonic class; conservative for what matters here."""
    op, ops = i["op"], i["ops"]
    if not ops or op.startswith("s_") or op.startswith("buffer_") or op.startswith("global_store"):
        return set(), set().union(*[vregs(o) for o in ops]) if ops else set()
    if op.startswith("v_permlane"):
        both = vregs(ops[0]) | vregs(ops[1])
        return both, both
    if op.startswith("v_cmp"):
        return set(), set().union(*[vregs(o.split(" ")[0]) for o in ops])
    w = vregs(ops[0])
    r = set().union(*[vregs(o.split(" ")[0].lstrip("-|").rstrip("|")) for o in ops[1:]]) if len(ops) > 1 else set()
    return w, r


def wait_states(i):
    return int(i["ops"][0]) + 1 if i["op"] == "s_nop" and i["ops"] else 1


def audit(listing, kernel=None):
    ins, meta = parse(listing, kernel or KERNELS[0])
    problems = []
    arch = meta.get("vgpr_count", 0) - meta.get("agpr_count", 0)
    if meta.get("private_segment_fixed_size", 1) != 0 or meta.get("vgpr_spill_count", 1) != 0 or meta.get("sgpr_spill_count", 1) != 0:
        problems.append(f"1: scratch / spills: {meta}")
    if meta.get("agpr_count") != N_AGPR or arch > 256:
        problems.append(f"1: expected exactly {N_AGPR} asm-owned AGPRs and <= 256 arch VGPRs, got {meta}")
    n_mfma = 0
    for k, i in enumerate(ins):
        touches_a = i["op"].startswith("v_accvgpr") or any(vregs(o, "a") for o in i["ops"])
        if touches_a and not i["asm"]:
            problems.append(f"2: compiler-generated AGPR access: {i['text']}")
        if i["op"].startswith("v_mfma"):
            n_mfma += 1
            if not i["asm"]:
                problems.append(f"3: MFMA outside an asm statement: {i['text']}")
            dst = vregs(i["ops"][0])
            if dst:          # VGPR destination: nobody but the accumulate chain touches it for MIN_MFMA_GAP wait states
                states = 0
                for j in ins[k + 1:k + 1 + MIN_MFMA_GAP]:
                    if states >= MIN_MFMA_GAP:
                        break
                    if not (j["op"].startswith("v_mfma") or j["op"].startswith("ds_read") and not (vregs(j["ops"][0]) & dst)):
                        w, r = writes_reads(j)
                        if (w | r) & dst:
                            problems.append(f"4: '{j['text']}' touches the destination of '{i['text']}' only {states} wait states behind it")
                            break
                    states += wait_states(j)
            srcs = set().union(*[vregs(o) for o in i["ops"][1:]])
            states = 0
            for j in reversed(ins[max(0, k - 4):k]):
                if j["op"] == "s_nop":
                    states += wait_states(j)
                    continue
                w, _ = writes_reads(j)
                if j["op"].startswith("v_") and not j["op"].startswith("v_mfma") and w & srcs and states < 2:
                    problems.append(f"7: '{j['text']}' writes a source of '{i['text']}' only {states} wait state(s) ahead")
                states += 1
                if states >= 2:
                    break
        if i["op"].startswith("v_permlane"):
            regs = vregs(i["ops"][0]) | vregs(i["ops"][1])
            states = 0
            for j in reversed(ins[max(0, k - 3):k]):
                if j["op"] == "s_nop":
                    states += wait_states(j)
                    continue
                if states >= 2:
                    break
                w, _ = writes_reads(j)
                if j["op"].startswith("v_") and w & regs:
                    problems.append(f"5: '{j['text']}' writes an operand of '{i['text']}' only {states} wait state(s) ahead")
                states += 1
        if i["op"].startswith("v_exp_f32") and k + 1 < len(ins):
            w, _ = writes_reads(i)
            _, r = writes_reads(ins[k + 1])
            if w & r:
                problems.append(f"6: '{ins[k + 1]['text']}' reads the transcendental result of '{i['text']}' right behind it")
    summary = dict(meta, arch_vgprs=arch, instructions=len(ins), mfma=n_mfma, asm_statements=sum(1 for i in ins if i["asm"]),
                   s_nop=sum(1 for i in ins if i["op"] == "s_nop"))
    return problems, summary


def audit_all(listing):
    out = []
    for k in KERNELS:
        if k in listing:
            problems, summary = audit(listing, k)
            out.append((k, problems, summary))
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# Library-wide check (round 3, after a run-dependent failure of the LNFOLD_T epilogue): hipcc may place a VALU instruction that
# overwrites the DATA registers of a wide (>= 96-bit) LDS or memory store one or two slots behind that store.  The store still reads
# its data then (observed with ds_write_b128 followed by v_pk_mul_f32 into the same registers: single elements of the stored vector
# went out stale, in some runs).  No kernel of libvcx may contain that pattern.
# ---------------------------------------------------------------------------------------------------------------------------
CSRC = os.path.join(ROOT, "viewcrafter_amd", "csrc")
LIB_SOURCES = ("gemm.hip", "gemm_dma.hip", "gemm_ws.hip", "attention.hip", "attention_v2.hip", "norm.hip", "elementwise.hip")
WIDE_STORES = ("ds_write_b128", "ds_write_b96", "ds_write2_b64", "ds_write2st64_b64", "buffer_store_dwordx4", "buffer_store_dwordx3",
               "global_store_dwordx4", "global_store_dwordx3", "flat_store_dwordx4", "flat_store_dwordx3")


def store_data_hazards(listing):
    """[(kernel, store, offending instruction, slots behind)] of one hipcc -S listing."""
    lines = [l.strip() for l in listing.split("\n")]
    found, kern = [], "?"
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l:
            kern = l.split(":")[0]
        op, _, rest = l.partition(" ")
        if op not in WIDE_STORES:
            continue
        data = set()
        for o in [x.strip() for x in rest.split(",")][:3]:
            r = vregs(o)
            if len(r) >= 3 or (op.startswith("ds_write2") and len(r) == 2):
                data |= r
        slots, j = 0, i + 1
        while j < len(lines) and slots < 2:
            t = lines[j]
            j += 1
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            top, _, trest = t.partition(" ")
            if top == "s_nop":
                slots += int(trest.split()[0]) + 1
                continue
            slots += 1
            if top.startswith("v_") and not top.startswith("v_cmp") and not top.startswith("v_mfma") and trest:
                if vregs(trest.split(",")[0].strip()) & data:
                    found.append((kern, l, t, slots))
    # second pattern: a packed fp32 multiply-add whose LOW result selects the HIGH half of a source pair (op_sel:[...] with a 1).
    # hipcc emitted `v_pk_fma_f32 v[60:61], v[60:61], v[4:5], v[0:1] op_sel:[0,1,1]` in ONE kernel (128x128 tile, LNFOLD_T epilogue,
    # odd row group) and exactly those results - low halves, lanes 48-63 - were run-dependent on the MI355X, with every operand
    # long since written.  The only other packed fp32 instruction with such a select in the library, the horizontal
    # `v_pk_add_f32 x, x, x op_sel:[0,1] op_sel_hi:[1,0]` of gn_stats_kernel, is bit-reproducible (test_groupnorm_and_folded_projections_
    # are_bit_reproducible) and is the one allowed exception; any NEW occurrence of the form fails the audit.
    kern = "?"
    for l in lines:
        if l.startswith("_Z") and ":" in l:
            kern = l.split(":")[0]
        m = re.match(r"(v_pk_(?:fma|mul|add)_f32)\s.*\bop_sel:\[([01,]+)\]", l)
        if m and "1" in m.group(2) and not (m.group(1) == "v_pk_add_f32" and "gn_stats_kernel" in kern):
            found.append((kern, l, "low result of a packed fp32 operation takes a high source half", 0))
    return found


# Kernels that OWN accumulator registers by name (asm statements read / write them; the compiler must not allocate AGPRs of its own in
# them, it does not know the owned ones are live): kernel name fragment -> number of v_accvgpr_write / v_accvgpr_read the source itself issues.
AGPR_OWNERS = {"gemm_ws320_geglu_kernel": (160, 0),
               # + colsum / bias' of the lane's columns in a[160:223], read back by the source itself: 64 reads in each of the 4 epilogue copies
               "gemm_ws320_lnf_kernel": (224, 256)}


def agpr_ownership(listing):
    """[(kernel, what)] - compiler-generated AGPR traffic or scratch in a kernel that owns its accumulator registers."""
    found = []
    for frag, (writes, reads) in AGPR_OWNERS.items():
        m = re.search(r"^(_Z\w*" + frag + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", listing, re.S | re.M)
        if not m:
            continue
        name, body = m.group(1), m.group(2)
        nw = len(re.findall(r"^\s*v_accvgpr_write_b32", body, re.M))
        nr = len(re.findall(r"^\s*v_accvgpr_(?:read|mov)_b32", body, re.M))
        if nw != writes or nr != reads:
            found.append((name, "", f"{nw} v_accvgpr_write (the source issues {writes}) and {nr} v_accvgpr_read / mov (the source issues {reads}): the compiler parks values in owned accumulator registers", 0))
        if re.search(r"^\s*scratch_", body, re.M):
            found.append((name, "", "scratch traffic", 0))
    return found


# Kernels whose hand-written vector-memory wait counts ASSUME how many stores the compiled epilogue of a tile issues (gemm_ws320_kernel:
# `S = 12`, 22 with column moments - gemm_ws.hip; a compiler that split or merged those stores would leave DMA pieces in flight behind
# the wait and the MFMAs would read a stale LDS stage - ADVICE r5): kernel name fragment -> vector-memory stores in the whole kernel body.
STORE_COUNTS = {"gemm_ws320_kernelILi2E": 12, "gemm_ws320_kernelILi3E": 22}


def assumed_store_counts(listing):
    found = []
    for frag, want in STORE_COUNTS.items():
        m = re.search(r"^(_Z\w*" + frag + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", listing, re.S | re.M)
        if not m:
            continue
        n = len(re.findall(r"^\s*(?:buffer|global|flat)_store_", m.group(2), re.M))
        if n != want:
            found.append((m.group(1), "", f"{n} vector-memory stores in the kernel, its vmcnt immediates assume {want} per tile", 0))
    return found


def audit_library_store_hazards():
    out = []
    with tempfile.TemporaryDirectory(prefix="vcx_isa_") as tmpdir:
        for name in LIB_SOURCES:
            tmp = os.path.join(tmpdir, name + ".s")
            flags = makefile_flags(name)
            r = subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", tmp, os.path.join(CSRC, name)],
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on " + name + ":\n" + r.stderr[-2000:])
            out += [(name,) + h for h in store_data_hazards(open(tmp).read())]
            out += [(name,) + h for h in agpr_ownership(open(tmp).read())]
            out += [(name,) + h for h in assumed_store_counts(open(tmp).read())]
    return out


def makefile_flags(name):
    """The compile flags libvcx.so is built with, read from csrc/Makefile (CXXFLAGS plus the per-object additions), so that the audited
    listing is the code that ships (a hand-copied flag list can drift)."""
    flags, extra = None, []
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"CXXFLAGS\s*:=\s*(.*)", line)
        if m:
            flags = m.group(1).replace("$(ARCH)", "gfx950").split()
        m = re.match(r"build/(\w+)\.o:\s*CXXFLAGS\s*\+=\s*(.*)", line)
        if m and m.group(1) + ".hip" == name:
            extra += m.group(2).split()
    if flags is None:
        raise RuntimeError("no CXXFLAGS in csrc/Makefile")
    return [f for f in flags if f not in ("-Wall", "-Wno-unused-function")] + extra


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--stores":
        hz = audit_library_store_hazards()
        for h in hz:
            print("HAZARD", h)
        print("clean" if not hz else f"{len(hz)} store-data hazard(s)")
        return 1 if hz else 0
    bad = 0
    for k, problems, summary in audit_all(compile_listing(sys.argv[1] if len(sys.argv) > 1 else None)):
        print(k + ":", summary)
        for p in problems:
            print("PROBLEM", p)
        bad += len(problems)
    print("clean" if not bad else f"{bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
