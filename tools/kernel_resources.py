#!/usr/bin/env python3
"""Build-container aid (no GPU): VGPRs, SGPRs, LDS, scratch and spill counts of every kernel in the built objects
(csrc/*.o -> .hip_fatbin -> gfx950 code object -> the AMDGPU metadata notes).  What decides how many workgroups share a
CU, and where a spill reload (an `s_waitcnt vmcnt(0)` in disguise, HISTORY 11.9) may sit.
    python tools/kernel_resources.py [substring]      # after python -m multi_view_stereonet_amd.build"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
want = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    for obj in sorted(glob.glob(os.path.join(ROOT, "multi_view_stereonet_amd", "csrc", "*.o"))):
        base = os.path.basename(obj)[:-2]
        fat, co = os.path.join(tmp, base + ".fat"), os.path.join(tmp, base + ".co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co):
            continue   # no device code in this object
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
            if want not in name:
                continue
            v = get("vgpr_count")
            waves = 512 // max(1, -(-int(v) // 8) * 8) if v.isdigit() else 0
            print("%-16s %-78s vgpr %3s (%d waves/SIMD) sgpr %3s lds %6s scratch %4s spills v%s s%s" % (
                base, name[:78], v, min(8, waves), get("sgpr_count"), get("group_segment_fixed_size"),
                get("private_segment_fixed_size"), get("vgpr_spill_count"), get("sgpr_spill_count")))
