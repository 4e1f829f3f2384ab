#!/usr/bin/env python3
"""No kernel may use scratch unless it is listed: scratch traffic shares vmcnt with the hand-managed LDS-DMA and
plane-store pipelines, so a new spill is a performance AND a hazard-surface regression.

    audit_scratch.py <file.s> [<file.s> ...]

Reads `; ScratchSize: N` per kernel symbol from hipcc's assembly, demangles the symbol (c++filt) and compares with
torch-nerf_amd/csrc/scratch_allow.txt (`<max bytes> <demangled kernel, anonymous namespaces and arguments removed>`).
Exit status 1 on any kernel over its allowance (0 for everything not listed)."""
import os
import re
import subprocess
import sys

ALLOW = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-nerf_amd", "csrc", "scratch_allow.txt")


def short(demangled):
    d = re.sub(r"\(anonymous namespace\)::", "", demangled)
    d = re.sub(r"^void ", "", d)
    return re.sub(r"\(.*\)$", "", d).strip()          # drop the argument list


def kernels(path):
    cur, out = None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\S+|[A-Za-z_]\w*):\s*(;.*)?$", ln)
        if m:
            cur = m.group(1)
        m = re.match(r"; ScratchSize: (\d+)", ln)
        if m and cur:
            out[cur] = int(m.group(1))
    names = subprocess.run(["c++filt"] + list(out), capture_output=True, text=True, check=True).stdout.splitlines()
    return {short(n): v for n, v in zip(names, out.values())}


def main(paths):
    allow = {}
    for ln in open(ALLOW):
        ln = ln.split("#")[0].strip()
        if ln:
            size, name = ln.split(None, 1)
            allow[name.strip()] = int(size)
    bad, seen = [], 0
    for p in paths:
        for name, size in kernels(p).items():
            seen += 1
            if size > allow.get(name, 0):
                bad.append(f"{os.path.basename(p)}: {name} uses {size} B of scratch (allowed {allow.get(name, 0)})")
    for b in bad:
        print(b)
    print(f"scratch audit: {seen} kernels, {len(bad)} over their allowance")
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
