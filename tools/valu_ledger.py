#!/usr/bin/env python3
"""Static half of the instruction ledger of the fused 4:2:0 -> RGB kernel: compiles zj_kernels.hip with
--save-temps, cuts the hot instantiation's ISA into its phases and counts instructions per phase and per kind.  The dynamic
half (what the hardware executed, phase by phase) comes from tools/valu_ledger.sh on the GPU box; profiles/r04_valu_ledger.txt
holds both and reconciles them.

    python tools/valu_ledger.py [--asm FILE.s] [--kernel MANGLED] [--blocks]

Phases are found by structure, not by label numbers: the two s_barrier instructions, the inlined functions' exit labels LLVM
keeps as comments (%_ZN2zj...exit), the v_dot2 / v_mad_i32_i24 density of the two transforms, and the "; zj-rare-branch"
comment ZJ_NO_IF_CONVERT() leaves at the head of every rarely taken branch ("cold": edge tiles, the wide redo).
A count here is per WAVE that executes the phase; which waves do is stated per phase."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = "_ZN2zj15zj_fused_kernelILi2ELi2ELi0ELi1ELb1ELb1EEEvNS_6ParamsE"

KINDS = [("dot2", r"v_dot2"), ("mul/mad i24,u24,u32,u64", r"v_(mul|mad)_(i32_i24|u32_u24|lo_u32|hi_u32|u64_u32|i64_i32)"),
         ("pk_mul/mad", r"v_pk_(mul|mad)"), ("pk_add/sub", r"v_pk_(add|sub)"), ("pk_shift", r"v_pk_(ashrrev|lshlrev|lshrrev)"),
         ("pk_min/max", r"v_pk_(min|max)"), ("perm/alignbit/bfi", r"v_(perm_b32|alignbit|bfi|bfe|and_or|lshl_or|or3)"),
         ("sat_pk_u8", r"v_sat_pk"), ("sad", r"v_sad"), ("dpp mov", r"v_mov_b32_dpp"), ("cndmask/cmp", r"v_(cndmask|cmp)"),
         ("add/sub/shift 32", r"v_(add|sub|subrev|ashrrev|lshlrev|lshrrev|lshl_add|add_lshl|add3|mov|and|or|xor|readfirstlane|med3|cvt|rcp|mul_f32)")]


def compile_asm():
    d = tempfile.mkdtemp(prefix="zj_ledger_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                           "-Wno-unused-function", "-Wno-pass-failed", "--save-temps", "-c",
                           os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_kernels.hip"), "-o", os.path.join(d, "k.o")], cwd=d,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(d, "zj_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
        if on:
            out.append(ln.rstrip("\n"))
            if ln.startswith(".Lfunc_end"):
                break
    if not out:
        sys.exit(f"{name} not found in {path}")
    return out


def blocks_of(lines):
    blocks, cur = [], {"label": "entry", "comment": "", "ins": [], "cold": False}
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", s)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "comment": (m.group(2) or ""), "ins": [], "cold": False}
            continue
        if s.startswith(";;#ASMSTART") or s.startswith(";;#ASMEND"):
            continue
        if s.startswith("; zj-rare-branch"):   # ZJ_NO_IF_CONVERT() at the head of a rarely taken, wave-uniform branch
            cur["cold"] = True
            continue
        m = re.match(r"^; %bb\.\d+:", s)
        if m:                     # fall-through block without a label: keep it separate so that "cold" stays local
            blocks.append(cur)
            cur = {"label": s.split(":")[0][2:], "comment": s, "ins": [], "cold": False}
            continue
        if not s or s.startswith((";", ".")):
            continue
        cur["ins"].append(s)
    blocks.append(cur)
    return blocks


def kind_of(op):
    if op.startswith("v_"):
        for name, rx in KINDS:
            if re.match(rx, op):
                return "V:" + name
        return "V:other"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_barrier":
        return "s_barrier"
    if op.startswith("s_"):
        return "S"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "VMEM"
    if op.startswith("scratch_"):
        return "scratch"
    return "?"


def count(blks):
    c = collections.Counter()
    for b in blks:
        for s in b["ins"]:
            c[kind_of(s.split()[0])] += 1
    return c


def totals(c):
    v = sum(n for k, n in c.items() if k.startswith("V:"))
    return v, c["S"], c["LDS"], c["VMEM"], c["s_nop"], c["s_waitcnt"], c["scratch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--kernel", default=HOT)
    ap.add_argument("--blocks", action="store_true", help="also print every basic block")
    a = ap.parse_args()
    path = a.asm or compile_asm()
    B = blocks_of(kernel_lines(path, a.kernel))
    idx_bar = [i for i, b in enumerate(B) if any(s.split()[0] == "s_barrier" for s in b["ins"])]
    has = lambda b, rx, n=1: sum(bool(re.match(rx, s.split()[0])) for s in b["ins"]) >= n  # noqa: E731
    dot_blocks = [i for i, b in enumerate(B) if has(b, r"v_dot2", 100)]
    packed = dot_blocks[0]                                   # the full form of the packed transform
    sparse = dot_blocks[1] if len(dot_blocks) > 1 else None  # its sparse form (columns 5..7 empty below the first row)
    wide = [i for i, b in enumerate(B) if has(b, r"v_mad_i32_i24", 100)]
    classify = next(i for i, b in enumerate(B) if has(b, r"v_sad", 16))

    def find(rx, start=0):
        for i in range(start, len(B)):
            if re.search(rx, B[i]["comment"]):
                return i
        return None
    bar1 = idx_bar[0]                                        # the block waves' first barrier (their path comes first in the layout)
    halo_loc = find(r"halo_locate.*exit")
    halo_end = find(r"halo_filter.*exit")
    # the halo wave's path starts a few scalar blocks in front of halo_locate's exit label: at the block behind the last
    # block of the block waves' path (the one that carries finish_block's exit label)
    fb_exit = max(i for i in range(packed, halo_loc) if "finish_block" in B[i]["comment"])
    halo_start = fb_exit + 1
    # (LLVM does not always keep phase_color's exit label; then the staging writes are counted with the colour phase)
    st0 = find(r"stage_item.*exit", halo_end)
    pc0 = find(r"phase_colorINS_3CfgILi2ELi2ELi0EEELi2ELi2ELi0ELi1ELb1ELb1E.*exit", halo_end)
    if pc0 is None or pc0 > st0:
        pc0 = st0
    co0 = find(r"color_copyout.*exit", st0)
    st1 = find(r"stage_item.*exit", co0 + 1)
    pc1 = find(r"phase_color.*exit", co0 + 1)
    if pc1 is None or pc1 > st1:
        pc1 = st1
    co1 = find(r"color_copyout.*exit", st1)
    # the staged-store rounds start where the wide redo path (tile_wide, inlined behind the second barrier) ends: the
    # last block before pc0 that begins with the kernel-argument reloads of phase_color
    ts0 = max(i for i in range(wide[-1] if wide else packed, pc0) if any(s.startswith("s_load_dwordx4") for s in B[i]["ins"]) and not B[i]["cold"])
    after_full = sparse if sparse is not None else halo_start
    phases = [
        ("tile decode, stagger test; block waves: addresses, load issue, table + LUT staging (to their first barrier)", range(0, bar1 + 1), "3 block waves (the halo wave runs the first ~20 blocks of it: tile decode)"),
        ("classify_block (DC-only test, packed-IDCT guard) + the test for the sparse form", range(bar1 + 1, packed), "3 block waves"),
        ("packed IDCT, FULL form (v_dot2_i32_i16)", range(packed, packed + 1), "block waves with a block that has anything below row 0 in columns 5..7"),
        ("  its results -> LDS (luma bytes / chroma i16 rows)", range(packed + 1, after_full), "the same"),
    ]
    if sparse is not None:
        phases += [("packed IDCT, SPARSE form (three column transforms of pass 1 skipped)", range(sparse, sparse + 1), "block waves where no block has"),
                   ("  its results -> LDS, DC-only splats", range(sparse + 1, halo_start), "the same")]
    phases += [
        ("halo wave: addresses, loads, table staging, column pass, row pass, vertical filter of the halo columns", range(halo_start, halo_end + 1), "the halo wave"),
        ("colour round 0: luma unpack, vertical + horizontal chroma filters, YCbCr->RGB, clamp + interleave", range(ts0, pc0), "all 4 waves"),
        ("round 0 staging (ds_write) ", range(pc0, st0), "all 4 waves"),
        ("round 0 copy-out (addresses, ds_read, global_store)", range(st0, co0), "all 4 waves"),
        ("colour round 1", range(co0, pc1), "all 4 waves"),
        ("round 1 staging", range(pc1, st1), "all 4 waves"),
        ("round 1 copy-out", range(st1, co1), "all 4 waves"),
    ]
    covered = set()
    print(f"{'phase':112s} {'VALU':>5s} {'SALU':>5s} {'LDS':>4s} {'VMEM':>4s} {'nop':>4s} {'wait':>4s} | cold (edge tiles, not executed by interior tiles): VALU SALU")
    for name, rng, who in phases:
        hot = [B[i] for i in rng if not B[i]["cold"] and i not in wide]
        cold = [B[i] for i in rng if B[i]["cold"]]
        covered.update(rng)
        v, s, l, m, nop, w, _ = totals(count(hot))
        cv, cs, *_ = totals(count(cold))
        print(f"{name:112s} {v:5d} {s:5d} {l:4d} {m:4d} {nop:4d} {w:4d} | {cv:4d} {cs:4d}   [{who}]")
    rest = [B[i] for i in range(len(B)) if i not in covered]
    v, s, l, m, nop, w, sc = totals(count(rest))
    print(f"{'not on the hot path: wide IDCT fall-back, tile_wide redo (Q1), generic tails':112s} {v:5d} {s:5d} {l:4d} {m:4d} {nop:4d} {w:4d} | scratch instructions {sc}")
    print()
    kinds = [("packed IDCT, full form", range(packed, packed + 1)), ("classify_block", range(classify, classify + 1)), ("colour round 0 (hot blocks)", range(ts0, pc0))]
    if sparse is not None:
        kinds.insert(1, ("packed IDCT, sparse form", range(sparse, sparse + 1)))
    for name, rng in kinds:
        c = count([B[i] for i in rng if not B[i]["cold"]])
        items = ", ".join(f"{k[2:]} {n}" for k, n in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("V:"))
        print(f"{name}: {items}")
    if a.blocks:
        print()
        for i, b in enumerate(B):
            v, s, l, m, nop, w, sc = totals(count([b]))
            if v + s + l + m:
                print(f"{i:4d} {b['label']:12s} {'cold' if b['cold'] else '':4s} V{v:5d} S{s:4d} L{l:3d} M{m:3d} X{sc:3d}  {b['comment'][:80]}")


if __name__ == "__main__":
    main()
