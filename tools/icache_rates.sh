cd "${GRAFT_REPO_ROOT}"
export TMPDIR=/tmp
for cfg in "--nseq 1000000" "--nseq 1000000 --order 1" "--nseq 1000000 --ss" "--nseq 1000000 --order 3" "--config c4"; do
  rm -rf /tmp/pm; timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d /tmp/pm -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 12 $cfg > /dev/null 2>&1
  python3 - "$cfg" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "k_em" in r["Kernel_Name"] or "k_m_" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v[-3:]) / len(v[-3:]) for c, v in d.items()}
    print(sys.argv[1], "|", k, "| req %.3g miss %.3g dup %.3g  miss rate %.4f %%" % (m["SQC_ICACHE_REQ"], m["SQC_ICACHE_MISSES"], m["SQC_ICACHE_MISSES_DUPLICATE"], 100 * (m["SQC_ICACHE_MISSES"] + m["SQC_ICACHE_MISSES_DUPLICATE"]) / m["SQC_ICACHE_REQ"]))
PY
done
