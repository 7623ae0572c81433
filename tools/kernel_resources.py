#!/usr/bin/env python3
"""Register / LDS / scratch / occupancy table of every kernel of the library, from hipcc's own remarks.

    python3 tools/kernel_resources.py [-D...] > profiles/rN_kernel_resources.txt

Compiles raytracer_2022_amd/csrc/hip/*.hip with the library's flags plus -Rpass-analysis=kernel-resource-usage
(objects go to a temporary directory) and prints one row per kernel: the demangled name, VGPRs, SGPRs, spilled
SGPRs / VGPRs, scratch bytes per lane, LDS bytes per workgroup, waves per SIMD. --only=<file.hip> restricts the run to one
source; other arguments are passed to hipcc.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(HERE, "raytracer_2022_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm",
         "-Rpass-analysis=kernel-resource-usage"]
FIELDS = [("VGPRs", "VGPRs"), ("TotalSGPRs", "SGPRs"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
          ("ScratchSize [bytes/lane]", "scratch"), ("LDS Size [bytes/block]", "lds"), ("Occupancy [waves/SIMD]", "waves")]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("rt2022::", "").replace("(anonymous namespace)::", "")
    name = name.replace("(unsigned int)", "").replace("(int)", "").replace("(bool)", "")
    depth = 0
    for i, ch in enumerate(name):                          # cut the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    extra = [a for a in sys.argv[1:] if not a.startswith("--only=")]
    only = [a[7:] for a in sys.argv[1:] if a.startswith("--only=")]       # e.g. --only=pt_wavefront.hip
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("pt_wavefront.hip", "pt_kernel.hip", "rt_api.hip"):
            if only and src not in only:
                continue
            p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, "hip", src), "-o", os.path.join(tmp, src + ".o")],
                               capture_output=True, text=True, cwd=CSRC)
            if p.returncode != 0:
                sys.stderr.write(p.stderr[-2000:])
                sys.exit(1)
            cur = None
            for line in p.stderr.splitlines():
                m = re.search(r"remark: Function Name: (\S+)", line)
                if m:
                    cur = {"mangled": m.group(1), "file": src}
                    rows.append(cur)
                    continue
                if cur is None:
                    continue
                for key, col in FIELDS:
                    m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
                    if m:
                        cur[col] = int(m.group(1))
    names = demangle([r["mangled"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = short(n)
    rows.sort(key=lambda r: (r["file"], r["name"]))
    print("# hipcc -Rpass-analysis=kernel-resource-usage, flags: %s" % " ".join(FLAGS[:-1] + extra))
    print("# wf_trace<STACK, STATS, FEAT, PROBE, WG, CACHE, PARTIAL, PRIMS>: FEAT bits 1 = triangles/rings, 2 = movers/lists, 4 = boxes/media")
    print("%-78s %5s %5s %10s %10s %8s %7s %5s" % ("kernel", "VGPR", "SGPR", "sgpr_spill", "vgpr_spill", "scratch", "LDS", "waves"))
    for r in rows:
        print("%-78s %5d %5d %10d %10d %8d %7d %5d" % (r["name"][:78], r.get("VGPRs", -1), r.get("SGPRs", -1), r.get("sgpr_spill", -1),
                                                      r.get("vgpr_spill", -1), r.get("scratch", -1), r.get("lds", -1), r.get("waves", -1)))


if __name__ == "__main__":
    main()
