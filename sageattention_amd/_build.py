"""Builds sageattention_amd/libsageattn_hip.so (the C-ABI library, include/sageattn_hip.h) with hipcc for
gfx950.  In-tree on purpose: the .so travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsageattn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# (source, extra flags).  The quantizers need exact IEEE semantics (bit-exact vs the oracle).
SOURCES = [
    ("sage_quant.hip", ["-ffp-contract=off"]),
    # packed f32 VALU is slower beside MFMAs; contraction off: the fused Q-quantizer prologue must round exactly like
    # K1 (sage_quant.hip) -- the tile loop spells its fmas out (__builtin_fmaf), so it is unaffected
    ("sage_attn.hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),
    ("sage_fp8.hip", []),
    ("sage_misc.hip", []),
    ("sage_op.hip", []),
]
# -Wno-inline-asm: lds_dma16 names M0 in its clobber list, which clang reports as "reserved register" (see the function)
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
          "-Rpass-analysis=kernel-resource-usage"]


def _check_no_scratch(src, compiler_output):
    """No kernel of this library may spill: the attention kernels issue their LDS-DMA from inline asm (invisible to
    the compiler's s_waitcnt bookkeeping), which is only safe while the compiler adds no scratch traffic of its own,
    and a spilling variant is a >3x performance cliff anyway.  Parsed from -Rpass-analysis=kernel-resource-usage."""
    import re
    name = None
    for line in compiler_output.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) != 0:
            raise RuntimeError(f"{src}: kernel {name} uses {m.group(1)} bytes/lane of scratch (register spill)")


def _check_occupancy(src, compiler_output):
    """The head_dim-64 kernels in their dispatched geometry (4 waves, no attn_mask) are tuned for THREE waves per SIMD
    (<= 168 VGPRs); hipcc's allocation sits within a few registers of that line and an innocent-looking source change can
    push a variant over it (-5 ... -10 % on the head_dim-64 shapes, silently).  Fail the build instead."""
    import re
    name = None
    for line in compiler_output.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r" VGPRs: (\d+)", line)
        if m and name and re.search(r"attn_i8_kernelILi64ELi4E(Lb[01]E){4}Lb0EEE", name) and int(m.group(1)) > 168:
            raise RuntimeError(f"{src}: kernel {name} uses {m.group(1)} VGPRs (> 168: two waves per SIMD instead of three)")


def _check_m0_private(obj):
    """lds_dma16 (sage_attn_common.h) sets M0 and declares it clobbered instead of saving and restoring it; clang does not
    honour clobbers of reserved registers, so this is only sound while nothing else in the device code touches M0.
    Checked on the object that is about to be linked: every M0 reference must be one of those scalar moves."""
    import glob
    import re
    import shutil
    import tempfile
    objdump = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin", "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    tmp = tempfile.mkdtemp(prefix="sage_m0_")
    try:
        local = os.path.join(tmp, "o.o")
        shutil.copy(obj, local)
        subprocess.run([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        dev = glob.glob(local + ".*" + ARCH)
        if not dev:
            raise RuntimeError(f"{obj}: no {ARCH} code object found")
        dis = subprocess.run([objdump, "-d", dev[0]], stdout=subprocess.PIPE, text=True, check=True).stdout
        bad = [l.strip() for l in dis.splitlines() if re.search(r"\bm0\b", l) and not re.search(r"s_mov_b32 m0, s\d+", l)]
        if bad:
            raise RuntimeError(f"{obj}: M0 is used outside lds_dma16:\n" + "\n".join(bad[:5]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _digest(*parts) -> str:
    import hashlib
    h = hashlib.sha256()
    for part in parts:
        h.update(part if isinstance(part, bytes) else str(part).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _stamp_ok(target, key) -> bool:
    """A target is current iff it exists and its stamp file holds `key` = a digest of everything it was built from
    (source text, every header, the compiler flags): a flag change or a header edit rebuilds, a touched file does not."""
    try:
        return os.path.exists(target) and _read(target + ".stamp").decode().strip() == key
    except OSError:
        return False


def _write_stamp(target, key):
    with open(target + ".stamp", "w") as f:
        f.write(key + "\n")


def build_variant(out: str, extra_flags, verbose: bool = False, only=None) -> str:
    """Timing-only variant build (ablations, A/B): separate objects, separate output library.  `only`: the sources the
    flags apply to -- the others are linked from the product objects (build() first)."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="sage_variant_")
    objs, procs = [], []
    for src, extra in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(CSRC, src.replace(".hip", ".o")))
            continue
        o = os.path.join(tmp, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [HIPCC] + COMMON + extra + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", o]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, pr in procs:
        o_, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{o_}")
    r = subprocess.run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is out of date and link.  The library's stamp covers every source, header and flag, so a snapshot
    that carries the built .so and its stamp but no objects (the GPU box: *.o stay behind, .gpurunignore) builds nothing."""
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "sageattn_hip.h"))
    hdr_key = _digest(*[_read(h) for h in hdrs])
    keys = {src: _digest(HIPCC, ARCH, " ".join(COMMON + extra), _read(os.path.join(CSRC, src)), hdr_key)
            for src, extra in SOURCES}
    lib_key = _digest(*[keys[src] for src, _ in SOURCES])
    if not force and _stamp_ok(LIB, lib_key):
        return LIB
    objs = []
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _stamp_ok(o, keys[src]):
            cmd = [HIPCC] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, o, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, o, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        try:
            _check_no_scratch(src, out)
            _check_occupancy(src, out)
        except RuntimeError:
            os.remove(o)  # never link (or cache) a spilling object
            raise
        _write_stamp(o, keys[src])
        if verbose:
            print("\n".join(l for l in out.splitlines() if "remark:" not in l))
    # whenever sage_attn.o is LINKED (not only when it was just compiled): nothing but lds_dma16 may touch M0.  The check
    # is a text search of the disassembly for `m0` operands; instructions that use M0 implicitly (s_movrel*, GWS,
    # s_sendmsg) would not be caught -- none of them occurs in code hipcc emits for these sources.
    _check_m0_private(os.path.join(CSRC, "sage_attn.o"))
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    _write_stamp(LIB, lib_key)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
