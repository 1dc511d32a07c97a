"""Build-time guard for the MVSN_VIS10 rule (csrc/mvsn_common.h): a kernel that receives a by-value struct carrying
device pointers must repeat those buffers as plain pointer arguments, because a captured hipGraph derives the cache
maintenance between kernel nodes from the pointer arguments it can see.  (Round 3: with the banded chain's buffers only
inside ChainArgs, ~0.3 % of graph-replayed batch-1 forwards were wrong by 1e-3..4e-3; tools/soak.py.)"""
import glob
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi_view_stereonet_amd", "csrc")


def _sources():
    out = {}
    for path in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")):
        with open(path) as f:
            out[os.path.basename(path)] = f.read()
    return out


def _structs(sources):
    """name -> body of every struct defined in csrc (and of the C ABI structs the kernels may take)."""
    found = {}
    for text in sources.values():
        for m in re.finditer(r"\bstruct\s+(\w+)\s*\{", text):
            depth, i = 1, m.end()
            while depth and i < len(text):
                depth += {"{": 1, "}": -1}.get(text[i], 0)
                i += 1
            found[m.group(1)] = text[m.end():i - 1]
    return found


def _carries_pointers(name, structs, seen=()):
    body = structs.get(name)
    if body is None or name in seen:
        return False
    body = re.sub(r"//[^\n]*", "", body)
    body = re.sub(r"\b(static|constexpr)\b[^;]*;", "", body)          # compile-time members
    if "*" in re.sub(r"\([^)]*\)", "", body):                         # (ignore member-function parameter lists)
        return True
    return any(_carries_pointers(t, structs, seen + (name,)) for t in re.findall(r"\b([A-Z]\w+)\s+\w+", body))


def test_struct_pointer_kernels_repeat_their_buffers_as_arguments():
    sources = _sources()
    structs = _structs(sources)
    checked, offenders = [], []
    for fname, text in sources.items():
        for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(([^{;]*?)\)\s*\{", text, re.S):
            kernel, params = m.group(1), m.group(2)
            by_value = [t for t in re.findall(r"(?:^|,)\s*(?:const\s+)?([A-Z]\w+)\s+\w+\s*(?=,|$)", params)
                        if _carries_pointers(t, structs)]
            if by_value:
                checked.append(kernel)
                if "MVSN_VIS10" not in params:
                    offenders.append(f"{fname}:{kernel}({', '.join(by_value)})")
    assert not offenders, "kernels with pointers hidden in by-value structs and no MVSN_VIS10: " + "; ".join(offenders)
    # the rule is in force where it is known to matter
    for k in ("chain_band_kernel", "chain_wino_kernel", "chain_kernel", "tower_kernel", "conv_wino_kernel"):
        assert k in checked, (k, checked)
