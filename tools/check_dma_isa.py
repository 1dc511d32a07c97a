#!/usr/bin/env python3
"""Build-time ISA check of the convolution kernels that issue their LDS-DMA as inline assembly
(conv_wino_kernel<.., DIL = 1, .., RIDE > 0>: mvsn_conv_wino.hip, wn_dma16 / wait_landed).

Those launches guard the DMA ring and the carried pass's loads with hand-counted `s_waitcnt vmcnt(N)`: the compiler's
waitcnt pass does not see the DMA, so the count is only right while the compiled instruction stream keeps the shape the
source assumes.  This script compiles the file to assembly with the library's flags and asserts, per such kernel:

  1. every LDS-DMA instruction sits in an intact inline-assembly block `s_mov_b32 m0, sN; s_nop 0; global_load_lds_dwordx4`
     (U, the carried job's residual) or `...; buffer_load_dwordx4 vN, s[..], 0 offen lds` (raw tiles, round 4) --
     no builtin-form DMA mixed in, nothing scheduled into the block;
  2. M0 is written nowhere else in the kernel (nothing can redirect a piece);
  3. no scratch (a spill is a VMEM instruction the counts do not know);
  4. (reported, not asserted) for every hand-placed counted wait `s_waitcnt vmcnt(N)`, N > 0, every control-flow path
     into it is walked back until N VMEM instructions have been passed: paths on which those are DMA pieces only are
     counted as `pieces-only`, the others as `mixed`.  A mixed path is not an error: vmcnt retires in order, so any
     extra load / store among the N youngest only makes the wait stricter; and WHICH counted wait runs is chosen at run
     time from `rd_young`, the number of pieces the issuing code itself counted since the carried loads -- a
     path-insensitive walk also follows combinations the counter excludes.  What the static walk cannot decide, the
     1000-launch bit-identity stress test (tests/test_hip_parity.py::test_carried_dilation1_launches_stress_bit_identical)
     covers dynamically.

   python tools/check_dma_isa.py            # exit code 0 = all kernels pass; prints one line per kernel
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "multi_view_stereonet_amd", "csrc", "mvsn_conv_wino.hip")
VMEM = re.compile(r"^\s*(global_(load|store|atomic)|buffer_(load|store|atomic)|flat_(load|store|atomic)|scratch_(load|store))")
KERNEL = re.compile(r"^(_ZN4mvsn16conv_wino_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)E(?:Lb[01]E|Li\d+E)?(?:Lb[01]E)?E\S*):")


def assemble() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "wino.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-S",
               "--cuda-device-only", SRC, "-o", out] + os.environ.get("MVSN_HIPCC_FLAGS", "").split()
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(text):
    lines = text.split("\n")
    starts = [(i, KERNEL.match(l)) for i, l in enumerate(lines)]
    starts = [(i, m) for i, m in starts if m]
    for k, (i, m) in enumerate(starts):
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        mode, ks, nstage, dil, vol, ride = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7))
        yield m.group(1), dict(mode=mode, ks=ks, nstage=nstage, dil=dil, vol=vol, ride=ride), lines[i + 1:end]


def check(name, p, body):
    errs = []
    # basic blocks: [label, [(text, asm_block_id or None)]]; inline-assembly blocks are numbered
    bbs, blocks, blk = [["<entry>", []]], [], None
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            blk = len(blocks)
            blocks.append([])
            continue
        if t.startswith(";;#ASMEND"):
            blk = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            bbs.append([m.group(1), []])
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        bbs[-1][1].append((t, blk))
        if blk is not None:
            blocks[blk].append(t)
    index = {bb[0]: i for i, bb in enumerate(bbs)}
    preds = {i: set() for i in range(len(bbs))}
    for i, (label, ins) in enumerate(bbs):
        falls = True
        for t, _ in ins:
            m = re.match(r"s_(c?branch\S*)\s+(\.LBB\d+_\d+)", t)
            if m:
                preds[index[m.group(2)]].add(i)
                if m.group(1) == "branch":
                    falls = False
            if t.startswith(("s_endpgm", "s_setpc")):
                falls = False
        # (a conditional branch in the middle of a block: clang ends blocks at terminators, so branches are last)
        if falls and i + 1 < len(bbs):
            preds[i + 1].add(i)
    stream = [x for _, ins in bbs for x in ins]
    def is_dma(x):   # flat form (U, the carried job's residual) or descriptor form (raw tiles: hardware range check)
        return x.startswith("global_load_lds_dwordx4") or (x.startswith("buffer_load_dwordx4") and x.endswith(" lds"))
    dma_blocks = {i for i, b in enumerate(blocks) if any(is_dma(x) for x in b)}
    # 1. block integrity, no DMA outside blocks
    for i in dma_blocks:
        b = blocks[i]
        if not (len(b) == 3 and b[0].startswith("s_mov_b32 m0, s") and b[1] == "s_nop 0" and is_dma(b[2])):
            errs.append(f"DMA block {i} is not [s_mov_b32 m0; s_nop 0; global_load_lds_dwordx4 | buffer_load_dwordx4 .. lds]: {b}")
    for t, bk in stream:
        if "load_lds" in t or (t.startswith("buffer_load") and " lds" in t):
            if bk is None:
                errs.append(f"LDS-DMA outside an inline-assembly block: {t}")
    # 2. no other M0 write
    for t, bk in stream:
        ops = t.split(None, 1)
        if len(ops) == 2 and ops[1].split(",")[0].strip() == "m0" and bk not in dma_blocks:
            errs.append(f"M0 written outside the DMA blocks: {t}")
    # 3. no scratch
    if any(t.startswith("scratch_") for t, _ in stream):
        errs.append("scratch instructions present (spills)")
    # 4. counted waits: on EVERY control-flow path into the wait, the N youngest VMEM instructions are DMA pieces
    counted = paths_total = mixed = 0
    for bi, (label, ins) in enumerate(bbs):
        for k, (t, bk) in enumerate(ins):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
            if not (m and bk is not None and int(m.group(1)) > 0):
                continue
            n = int(m.group(1))
            counted += 1
            # walk back: state = (block, position just before which we look, pieces still to see); memoised per block
            work, done, bad = [(bi, k, n)], set(), None
            while work and bad is None:
                b, pos, need = work.pop()
                j = pos - 1
                while j >= 0 and need > 0:
                    tj, bj = bbs[b][1][j]
                    if VMEM.match(tj):
                        if bj not in dma_blocks:
                            bad = tj
                            break
                        need -= 1
                    j -= 1
                if bad is not None or need == 0:
                    paths_total += need == 0
                    continue
                if not preds[b]:
                    bad = "<kernel entry reached with %d pieces still expected>" % need
                    break
                for pb in preds[b]:
                    if (pb, need) not in done:
                        done.add((pb, need))
                        work.append((pb, len(bbs[pb][1]), need))
            mixed += bad is not None
    return errs, len(dma_blocks), counted, (paths_total, mixed)


S2_KERNEL = re.compile(r"^(_ZN4mvsn19conv_wino_s2_kernel\S*):")


def check_s2(text):
    """conv_wino_s2_kernel waits for a step's DMA pieces with `s_waitcnt vmcnt(8)` at the top of a tile's first step
    (MVSN_S2_CNTWAIT): behind the pieces the wave has issued the finished tile's output stores, and the count is right
    only while AT LEAST eight vector-memory instructions sit between the last piece and the wait on every path that ran
    a tile epilogue (more only make the wait stricter; fewer -- a store the compiler merged or dropped -- would let a piece
    be in flight when the tile is read).  Asserted here; plus the
    DMA-block integrity and no-scratch properties of the other inline-assembly kernels."""
    lines = text.split("\n")
    start = next((i for i, l in enumerate(lines) if S2_KERNEL.match(l)), None)
    if start is None:
        return ["conv_wino_s2_kernel not found"], 0
    end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
    bbs, blk, nblk = [["<entry>", []]], None, 0
    for l in lines[start + 1:end]:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            blk = nblk
            nblk += 1
            continue
        if t.startswith(";;#ASMEND"):
            blk = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            bbs.append([m.group(1), []])
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if t:
            bbs[-1][1].append((t, blk))
    index = {bb[0]: i for i, bb in enumerate(bbs)}
    preds = {i: set() for i in range(len(bbs))}
    for i, (label, ins) in enumerate(bbs):
        falls = True
        for t, _ in ins:
            m = re.match(r"s_(c?branch\S*)\s+(\.LBB\d+_\d+)", t)
            if m:
                preds[index[m.group(2)]].add(i)
                if m.group(1) == "branch":
                    falls = False
            if t.startswith(("s_endpgm", "s_setpc")):
                falls = False
        if falls and i + 1 < len(bbs):
            preds[i + 1].add(i)

    def is_dma(x):
        return x.startswith("global_load_lds_dwordx4") or (x.startswith("buffer_load_dwordx4") and x.endswith(" lds"))
    errs, waits = [], 0
    if any(t.startswith("scratch_") for _, ins in bbs for t, _ in ins):
        errs.append("scratch instructions present (spills)")
    for bi, (label, ins) in enumerate(bbs):
        for k, (t, bk) in enumerate(ins):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
            if not (m and bk is not None and int(m.group(1)) > 0):
                continue
            n = int(m.group(1))
            waits += 1
            # every path back from the wait: count non-DMA VMEM instructions until the first DMA piece
            work, done, full = [(bi, k, 0)], set(), 0
            while work:
                b, pos, seen = work.pop()
                j, hit = pos - 1, False
                while j >= 0:
                    tj = bbs[b][1][j][0]
                    if VMEM.match(tj):
                        if is_dma(tj):
                            hit = True
                            break
                        seen += 1
                        if seen >= n:
                            hit = True
                            break
                    j -= 1
                if hit:
                    full += seen >= n
                    # (seen == 0: the path of a step that did not end a tile -- the walk is path-insensitive, and the
                    # source takes the counted wait only when the step before DID end one: `cc == 0 && step > 0`)
                    if 0 < seen < n:
                        errs.append(f"vmcnt({n}) with only {seen} vector-memory instructions behind the last DMA piece on a path through {bbs[b][0]}")
                    continue
                for pb in preds[b]:
                    if (pb, seen) not in done:
                        done.add((pb, seen))
                        work.append((pb, len(bbs[pb][1]), seen))
            if not full:
                errs.append(f"vmcnt({n}): no path with {n} vector-memory instructions between the last DMA piece and the wait")
    return errs, waits


def main():
    text = assemble()
    bad = 0
    checked = 0
    errs, waits = check_s2(text)
    print(f"conv_wino_s2_kernel: {waits} counted wait(s), each behind >= its count of younger vector-memory instructions: "
          f"{'OK' if not errs else 'FAIL'}")
    for e in errs[:10]:
        print("   ", e)
    bad += bool(errs)
    for name, p, body in kernels(text):
        if not (p["ride"] > 0 and p["dil"] == 1):
            continue
        errs, ndma, nwait, ndec = check(name, p, body)
        checked += 1
        tag = "mode=%(mode)d ks=%(ks)d stages=%(nstage)d dil=%(dil)d vol=%(vol)d ride=%(ride)d" % p
        print(f"conv_wino_kernel<{tag}>: {ndma} DMA sites, {nwait} counted waits ({ndec[0]} pieces-only paths, {ndec[1]} waits with a mixed path): {'OK' if not errs else 'FAIL'}")
        for e in errs[:10]:
            print("   ", e)
        bad += bool(errs)
    if checked == 0:
        print("no inline-assembly DMA kernel found: the check is stale")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
