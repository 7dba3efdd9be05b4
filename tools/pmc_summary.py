"""Per-kernel mean of a rocprofv3 --pmc counter_collection.csv (values per dispatch)."""
import csv, sys, collections
def main(path):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        acc[(row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:62s} {c:12s} n={len(v):4d} mean={sum(v)/len(v):14.1f} (KiB per dispatch -> {sum(v)/len(v)*1024/1e6:10.2f} MB)")
if __name__ == "__main__":
    main(sys.argv[1])
