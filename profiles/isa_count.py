#!/usr/bin/env python
"""Instruction-class census of a gfx950 kernel's hot loop, from hipcc's own assembly.

    python profiles/isa_count.py text2pos-cvpr2022_amd/csrc/ws_sa2.hip 'k_ws_sa2ILi256ELi256E' [-D FLAG ...]

Compiles the source with the product's flags (build.py) to assembly (device side only), finds the kernel whose mangled name
contains the pattern, and counts instructions per class (MFMA / VALU / SALU / LDS / VMEM / waitcnt / branch) for the whole kernel
and for its LARGEST basic-block loop (the backward branch that spans the most MFMAs): the batch loop of the SA kernels.
Registers, spills, LDS and occupancy come from the kernel's .amdhsa_ directives.  The notebook quotes these figures.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S"]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm":
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop") or op.startswith("s_setprio") or op.startswith("s_sleep"):
        return "nop/prio"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("v_accvgpr") or op.startswith("v_mov"):
        return "v_mov"
    if op.startswith("v_"):
        return "valu"
    return "other"


def census(lines):
    c = {}
    for op in lines:
        k = classify(op)
        c[k] = c.get(k, 0) + 1
    return c


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    out = f"/tmp/isa_{os.path.basename(src)}.s"
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", out, os.path.join(ROOT, src) if not os.path.isabs(src) else src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr)
    text = open(out).read().split("\n")
    # kernel bodies: "<name>:" ... ".end_amdhsa_kernel" region after it holds the directives
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and pat in l]
    if not starts:
        sys.exit(f"no kernel matching {pat!r} in {out}")
    for st in starts:
        name = text[st].split(":")[0]
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        # the last s_endpgm of the function: walk until .section / .Lfunc_end
        fe = next(i for i in range(st, len(text)) if text[i].startswith(".Lfunc_end"))
        body = text[st + 1: fe]
        ops, labels, idx = [], {}, 0
        for l in body:
            s = l.strip()
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = idx
                continue
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            ops.append(s)
            idx += 1
        opnames = [o.split()[0] for o in ops]
        total = census(opnames)
        # loops = backward branches; pick the one enclosing the most MFMAs
        best = None
        for i, o in enumerate(ops):
            m = re.match(r"^s_cbranch\w*\s+(\.LBB\w+)|^s_branch\s+(\.LBB\w+)", o)
            if not m:
                continue
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is None or tgt > i:
                continue
            n_mfma = sum(1 for x in opnames[tgt: i + 1] if x.startswith("v_mfma"))
            if best is None or n_mfma > best[0]:
                best = (n_mfma, tgt, i)
        meta = {}
        for l in text[fe: fe + 400]:
            m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)", l)
            if m:
                meta[m.group(1)] = m.group(2)
            if ".end_amdhsa_kernel" in l:
                break
        spill = [l for l in text[st: fe + 400] if "ScratchSize" in l or "Occupancy" in l or "SGPRSpill" in l or "VGPRSpill" in l]
        print(f"== {name}")
        print("   directives:", meta)
        for l in spill[:6]:
            print("   ", l.strip())
        print("   whole kernel:", dict(sorted(total.items())), "total", sum(total.values()))
        if best:
            loop = census(opnames[best[1]: best[2] + 1])
            print(f"   hot loop ({best[2] - best[1] + 1} instructions, {best[0]} MFMAs):", dict(sorted(loop.items())))
            non = sum(v for k, v in loop.items() if k != "mfma")
            print(f"   non-MFMA instructions per MFMA in the hot loop: {non / max(1, best[0]):.2f}")


if __name__ == "__main__":
    main()
