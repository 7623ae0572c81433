#!/usr/bin/env python3
"""Instruction mix of the node fast path of the traversal kernels, read from the gfx950 ISA hipcc emits.

    python3 tools/isa_mix.py > profiles/rN_node_step_isa_mix.txt

Compiles raytracer_2022_amd/csrc/hip/pt_wavefront.hip to assembly with the library's flags (device side only), finds in each
listed wf_trace instance the innermost loop that holds the node step (the loop with the node record's LDS / global loads),
and counts its instructions per turn = per node step of every lane that is in it:
  f64 VALU   v_*_f64 (one SIMD issue slot = a quad-cycle each)        32-bit VALU   every other v_* (two can share a slot)
  SALU       s_* but waits / nops / branches                          DS            ds_read / ds_write
  VMEM       global_* / flat_* / scratch_*                            branch, wait  s_cbranch / s_branch, s_waitcnt / s_nop
A static count of one loop, not a profile: the measured counterparts (SQ_INSTS_VALU by type over the whole kernel) are in the
bench line (`roofline.valu.mix`).
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(HERE, "raytracer_2022_amd", "csrc", "hip", "pt_wavefront.hip")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-math-errno", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm",
         "--cuda-device-only", "-S"]
KERNELS = [
    ("headline (book-2 final scene): wf_trace<16,false,6,false,1024,1740,false,false>, node table in LDS", "ILi16ELb0ELj6ELb0ELi1024ELi1740ELb0ELb0EEE"),
    ("C2 (random spheres): wf_trace<16,false,0,false,1024,600,false,true>, node table and sphere pools in LDS", "ILi16ELb0ELj0ELb0ELi1024ELi600ELb0ELb1EEE"),
    ("C4 (Cornell box): wf_trace<16,false,0,...,1740,false,false>", "ILi16ELb0ELj0ELb0ELi1024ELi1740ELb0ELb0EEE"),
    ("C5 (wwscene, 1.7 M nodes): wf_trace<30,false,3,false,256,0,false,false>, nodes from L1 / L2", "ILi30ELb0ELj3ELb0ELi256ELi0ELb0ELb0EEE"),
    ("s1e6 (1 M spheres): wf_trace<22,false,0,false,256,0,false,false>, nodes from L2 / HBM", "ILi22ELb0ELj0ELb0ELi256ELi0ELb0ELb0EEE"),
]


def classify(op):
    if op.startswith("v_"):
        return "f64 VALU" if "_f64" in op or op.startswith("v_mad_u64") or op.startswith("v_lshl_add_u64") or op.endswith("_b64") else "32-bit VALU"
    if op.startswith("ds_"):
        return "DS"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait / nop"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def node_loop(lines):
    """The innermost loop that holds the node step: the blocks tagged with the loop header that precedes the first node-record load."""
    first = next((i for i, l in enumerate(lines) if "v_pk_fma_f32" in l), None)   # (the single-precision slab test: only the fast path is written that way)
    f32 = first is not None
    if first is None:
        first = next((i for i, l in enumerate(lines) if "v_min_f64" in l), None)  # (the double-precision slab test's min / max: likewise)
    if first is None:
        return [], []
    j = first
    while j > 0 and "This Inner Loop Header" not in lines[j]:
        j -= 1
    k = j
    while k > 0 and not lines[k].startswith(".LBB"):
        k -= 1
    header = lines[k].split(":")[0]                    # '.LBBx_y'
    tag = "Header=%s " % header[2:]
    # block starts: '.LBBn_m:' labels and '; %bb.N:' comments; a block belongs to the loop if it is the header or carries the tag
    starts = [i for i, l in enumerate(lines) if l.startswith(".LBB") or l.startswith("; %bb.")]
    body, rare = [], []
    for n, st in enumerate(starts):
        en = starts[n + 1] if n + 1 < len(starts) else len(lines)
        head = lines[st] + (lines[st + 1] if st + 1 < len(lines) and lines[st + 1].lstrip().startswith(";") else "")
        if lines[st].startswith(header + ":") or tag in head:
            # (single-precision loop: the block that fetches the double-precision record is the second opinion of the few lanes
            # the float test leaves undecided — skipped by the wave otherwise, counted apart)
            if f32 and any("global_load" in l for l in lines[st:en]):
                rare += lines[st:en]
            else:
                body += lines[st:en]
    return body, rare


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "wf.s")
        if os.environ.get("ISA"):                          # (an assembly file made earlier with the same flags)
            out = os.environ["ISA"]
        p = subprocess.run(["true"]) if os.environ.get("ISA") else subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, SRC], capture_output=True, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-2000:])
            sys.exit(1)
        text = open(out).read().split("\n")
    print("# tools/isa_mix.py: instructions per turn of the node fast path (= per node step), from hipcc's gfx950 assembly of pt_wavefront.hip")
    print("# flags: %s" % " ".join(FLAGS[:-2]))
    for label, key in KERNELS:
        start = next((i for i, l in enumerate(text) if l.startswith("_ZN6rt20228wf_trace" + key) and l.rstrip().endswith(":") or (l.startswith("_ZN6rt20228wf_trace" + key) and ": " in l)), None)
        if start is None:
            print("%s: kernel not found" % label)
            continue
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        body, rare = node_loop(text[start:end])

        def count(block):
            counts = {}
            for l in block:
                l = l.strip()
                if not l or l.startswith((";", ".", "//")):
                    continue
                op = l.split()[0]
                c = "f32 packed / 3-operand" if op in ("v_pk_fma_f32", "v_max3_f32", "v_min3_f32") else classify(op)
                counts[c] = counts.get(c, 0) + 1
            return counts
        counts = count(body)
        if counts.get("f32 packed / 3-operand"):
            counts["32-bit VALU"] = counts.get("32-bit VALU", 0) + counts.pop("f32 packed / 3-operand")
        total = sum(counts.values())
        valu = counts.get("f64 VALU", 0) + counts.get("32-bit VALU", 0)
        # issue slots: an f64 instruction holds a quad-cycle; two 32-bit ones can share one (only across waves)
        print("%s" % label)
        print("    " + "  ".join("%s %d" % (k, counts.get(k, 0)) for k in ("f64 VALU", "32-bit VALU", "SALU", "DS", "VMEM", "branch", "wait / nop", "other")) + "  | all %d, VALU %d" % (total, valu))
        print("    vector issue slots per node step: between %.1f (every 32-bit pair shares a slot) and %d (none does)"
              % (counts.get("f64 VALU", 0) + counts.get("32-bit VALU", 0) / 2.0, valu))
        if rare:
            rc = count(rare)
            print("    (single-precision slab test; the double-precision second opinion, skipped unless a lane is undecided: %d instructions, %d of them f64)"
                  % (sum(rc.values()), rc.get("f64 VALU", 0)))


if __name__ == "__main__":
    main()
