"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats CSV run (first argument: output directory)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
