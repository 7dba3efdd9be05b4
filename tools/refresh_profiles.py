"""Copies the rocprofv3 summaries of tools/refresh_profiles.sh into profiles/ and rewrites
profiles/pmc_traffic.json (per-launch HBM bytes of every engine kernel).

    python tools/refresh_profiles.py <tag>        e.g. r1

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half the
bytes of a coalesced streaming read (checked against the streaming kernel, whose reads are known).
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")


def per_kernel(path):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ssmap::" in name:
            acc[name.replace("ssmap::", "")].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(tag):
    stats = glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    fetch = glob.glob(os.path.join(SRC, "fetch", "**", "*counter_collection.csv"), recursive=True)[0]
    write = glob.glob(os.path.join(SRC, "write", "**", "*counter_collection.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(DST, f"{tag}_bench_log_odds_c40_E16_kernel_stats.csv"))
    for other in glob.glob(os.path.join(SRC, "stats_*")):   # tools/refresh_workload_profiles.sh
        if os.path.isdir(other):
            found = glob.glob(os.path.join(other, "**", "*kernel_stats.csv"), recursive=True)
            if found:
                shutil.copy(found[0], os.path.join(DST, f"{tag}_bench_{os.path.basename(other)[6:]}_E16_kernel_stats.csv"))
    shutil.copy(fetch, os.path.join(DST, f"{tag}_pmc_fetch_size_counter_collection.csv"))
    shutil.copy(write, os.path.join(DST, f"{tag}_pmc_write_size_counter_collection.csv"))
    f, w = per_kernel(fetch), per_kernel(write)
    kernels = {}
    for k in sorted(f):
        kernels[k] = {"fetch_size_kib": round(f[k], 1), "write_size_kib": round(w.get(k, 0.0), 1),
                      "hbm_bytes_per_launch": round((2 * f[k] + w.get(k, 0.0)) * 1024)}
    stream = next(k for k in kernels if k.startswith("k_stream_reg<40,"))
    out = {
        "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py --steps 4 "
                "--warmup 1 --no-cpu-baseline` (profiles/%s_pmc_*_counter_collection.csv), mean per dispatch; "
                "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts half the bytes of a "
                "coalesced streaming read (2*FETCH_SIZE of the streaming kernel = its logits + update-slot words)" % tag,
        "log_odds_c40": {"E16": {"k_stream_bytes_per_launch": kernels[stream]["hbm_bytes_per_launch"],
                                 "kernels": kernels}},
    }
    with open(os.path.join(DST, "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    total = sum(v["hbm_bytes_per_launch"] for v in kernels.values())
    for k, v in kernels.items():
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB")
    print(f"{'sum over engine kernels':40s} {total / 1e6:9.1f} MB per step")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
