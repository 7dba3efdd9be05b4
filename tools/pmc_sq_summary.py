"""Per-kernel means of the two SQ counter passes of tools/pmc_bench.sh, with per-wave figures.

    python tools/pmc_sq_summary.py <tag>      reads gpurun_out/pmc_<tag>_{A,B}, writes profiles/<tag>_pmc_sq_*
"""
import collections, csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP_ROWS = 3200   # rows of each pass kept under profiles/ (100 steps of the default workload)


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ssmap::" in name:
            acc[name.replace("ssmap::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: round(sum(v) / len(v), 1) for c, v in cs.items()} for k, cs in acc.items()}


def main(tag):
    out = {}
    for letter in "AB":
        src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{letter}", "p_counter_collection.csv")
        kept = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_pass_{letter.lower()}_counter_collection.csv")
        with open(src) as fh, open(kept, "w") as out_fh:   # the first KEEP_ROWS rows: the steps repeat
            for i, line in enumerate(fh):
                if i > KEEP_ROWS:
                    break
                out_fh.write(line)
        for k, cs in per_kernel(kept).items():
            out.setdefault(k, {}).update(cs)
    for k, c in out.items():
        waves = c.get("SQ_WAVES", 0.0)
        if waves:
            c["valu_insts_per_wave"] = round(c.get("SQ_INSTS_VALU", 0.0) / waves, 1)
            c["vmem_rd_insts_per_wave"] = round(c.get("SQ_INSTS_VMEM_RD", 0.0) / waves, 1)
            c["vmem_wr_insts_per_wave"] = round(c.get("SQ_INSTS_VMEM_WR", 0.0) / waves, 1)
            c["lds_insts_per_wave"] = round(c.get("SQ_INSTS_LDS", 0.0) / waves, 1)
        if c.get("SQ_ACTIVE_INST_ANY"):
            c["valu_share_of_issue_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_ACTIVE_INST_ANY"], 3)
        if c.get("SQ_WAVE_CYCLES"):
            c["wait_share_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
    doc = {"_how": "rocprofv3 --pmc (two passes of 8 SQ counters, tools/pmc_bench.sh) over `python3 bench.py --steps 3 "
                   "--warmup 1 --no-cpu-baseline`; mean per dispatch of every engine kernel; per-wave figures = counter / SQ_WAVES",
           "log_odds_c40_E16": out}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_counters.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, c in out.items():
        print(k, {n: c[n] for n in ("valu_insts_per_wave", "vmem_rd_insts_per_wave", "vmem_wr_insts_per_wave", "lds_insts_per_wave",
                                    "valu_share_of_issue_cycles", "wait_share_of_wave_cycles") if n in c})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r2")
