#!/usr/bin/env python3
"""Instruction budget of a kernel of libaomarl_hip.so from its gfx950 assembly: per basic block, the number of
matrix / transcendental / packed / DPP / other vector instructions, LDS and memory instructions, scalar instructions,
waits and s_nops -- the static side of the counters in profiles/r*_pmc_frame_kernel_summary_*.txt.

    python tools/isa_budget.py [--kernel REGEX] [--asm FILE] [--lines]

Without --asm the environment's translation unit is compiled to assembly first (hipcc --cuda-device-only -S, ~40 s).
Default kernel: the bench's frame-kernel instantiation k_frame_wave<3, 1, true, false, false, false>.
profiles/r06_frame_kernel_isa_budget.txt is this output, annotated by hand with the source construct of every block."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ao_marl_amd", "csrc")
TRANS = ("v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32")


def category(op, text):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if "dpp" in op or " row_" in text or "quad_perm" in text:
        return "dpp"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_wait"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


ORDER = ("mfma", "trans", "v_pk", "dpp", "valu", "lds", "vmem", "smem", "salu", "branch", "s_wait", "s_nop", "barrier")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=r"_Z12k_frame_waveILi3ELi1ELb1ELb0ELb0ELb0EE")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--source", default="aomarl_capi.hip", help="translation unit under ao_marl_amd/csrc")
    ap.add_argument("--lines", action="store_true", help="print every instruction with its category")
    a = ap.parse_args()
    asm = a.asm
    if asm is None:
        asm = os.path.join(tempfile.gettempdir(), "aomarl_isa_budget.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm",
                               "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S", os.path.join(CSRC, a.source), "-o", asm],
                              stderr=subprocess.DEVNULL)
    txt = open(asm).read().split("\n")
    begin = end = None
    for i, l in enumerate(txt):
        if begin is None and re.match(r"^(%s\S*):" % a.kernel, l):
            begin = i
        elif begin is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if begin is None:
        sys.exit("kernel %r not found in %s" % (a.kernel, asm))
    body = txt[begin:end]
    print("# %s: %d lines of assembly" % (txt[begin].rstrip(":"), len(body)))
    for l in txt[end:end + 60]:
        m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m:
            print("#   %s = %s" % (m.group(1), m.group(2)))
    blocks, cur = [], ["entry", 0, [], ""]
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), i, [], (m.group(2) or "").strip("; ")]
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        op = t.split()[0]
        cur[2].append((i, op, t))
        if op.startswith(("s_cbranch", "s_branch")):
            blocks.append(cur)
            cur = [cur[0] + "+", i + 1, [], cur[3]]
    blocks.append(cur)
    total = collections.Counter()
    print("# %-16s %-11s %s   (loop annotation)" % ("block", "lines", " ".join("%7s" % k for k in ORDER)))
    for name, start, ins, note in blocks:
        if not ins:
            continue
        c = collections.Counter(category(op, t) for _, op, t in ins)
        total.update(c)
        print("  %-16s %5d-%-5d %s   %s" % (name, ins[0][0], ins[-1][0], " ".join("%7d" % c.get(k, 0) for k in ORDER), note))
        if a.lines:
            for i, op, t in ins:
                print("        %5d  %-7s %s" % (i, category(op, t), t))
    print("  %-16s %11s %s" % ("TOTAL (static)", "", " ".join("%7d" % total.get(k, 0) for k in ORDER)))


if __name__ == "__main__":
    main()
