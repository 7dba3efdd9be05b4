#!/usr/bin/env python3
"""Register, LDS and occupancy figures of every kernel of the engine, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_resources.py [substring ...] [--csv out.csv]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-search_amd"))
import build as ssa_build  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def resources(src, extra):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ssa_build.COMMON + extra + [
        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ssa_build.CSRC, src), "-o", "/dev/null"]
    text = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"kernel": val, "file": src}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    if csv in args:
        args.remove(csv)
    rows = []
    for src, extra in ssa_build.SOURCES:
        if src.endswith(".hip"):
            rows += resources(src, extra)
    names = demangle([r["kernel"] for r in rows])
    for r, n in zip(rows, names):
        r["kernel"] = re.sub(r"\(.*", "", n.replace("void ", "").replace("ssmap::", ""))
    cols = ("file", "kernel", "VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]",
            "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")
    lines = [",".join(cols)]
    for r in rows:
        if args and not any(a in r["kernel"] for a in args):
            continue
        lines.append(",".join('"%s"' % r.get(c, "") if c == "kernel" else str(r.get(c, "")) for c in cols))
    text = "\n".join(lines)
    print(text)
    if csv:
        with open(csv, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
