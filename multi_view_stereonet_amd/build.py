"""Build libmvsn_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no torch extension:
the library is a plain C-ABI shared object (include/mvsn_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmvsn_hip.so")
SOURCES = ["mvsn_error.hip", "mvsn_setup.hip", "mvsn_warp.hip", "mvsn_chain.hip", "mvsn_chain_wino.hip", "mvsn_chain_band.hip", "mvsn_chain_slab.hip", "mvsn_chain_steps.hip", "mvsn_conv.hip", "mvsn_conv_bf16x3.hip", "mvsn_conv_wino.hip", "mvsn_misc.hip", "mvsn_consistency.hip", "mvsn_prepare.hip", "mvsn_metrics.hip", "mvsn_tower.hip"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


MANIFEST = LIB + ".sources"
HEADERS = ["mvsn_common.h", "mvsn_conv_bf16x3.h", "mvsn_chain.h", "mvsn_conv_wino.h"]


def source_digest() -> str:
    """sha256 over every source the library is built from (+ the extra compiler flags): written next to the binary,
    so "is this .so the build of THESE sources" is a content check, not a file-time check (a snapshot copied to
    another box keeps its binary but not its mtimes)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, x) for x in HEADERS] + \
        [os.path.join(HERE, "..", "include", "mvsn_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(os.environ.get("MVSN_HIPCC_FLAGS", "").encode())
    return h.hexdigest()


def built_from_current_sources() -> bool:
    try:
        with open(MANIFEST) as f:
            return os.path.exists(LIB) and f.read().strip() == source_digest()
    except OSError:
        return False


def needs_build() -> bool:
    return not built_from_current_sources()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        # -fno-slp-vectorize: packed f32 VALU (v_pk_add_f32 ...) next to MFMAs costs more issue time than the two
        # scalar ops it replaces (measured: +1.3 % end to end, +3-4 % on the Winograd kernels)
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-c",
               os.path.join(CSRC, s), "-o", o]
        cmd += os.environ.get("MVSN_HIPCC_FLAGS", "").split()   # experiments: -D switches for A/B builds
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(MANIFEST, "w") as f:
        f.write(source_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
