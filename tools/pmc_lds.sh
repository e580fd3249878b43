cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_lds -- python3 tools/layer_bench.py ${PMC_LDS_ARGS:-64 128 bf16 3 "^G[2-4]|^D1"} > gpurun_out/pmc_lds.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = max(glob.glob("gpurun_out/pmc_lds/*/*counter_collection.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60] + " grid " + r["Grid_Size"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_INSTS_LDS": cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:16]:
    a, c, n = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), max(cnt[k], 1)
    print(f"{k:90s} launches {n:3d} LDS active/launch {a/n/1e6:8.2f} Mcyc conflict {c/n/1e6:8.2f} Mcyc ({100*c/max(a,1):4.1f}%) insts/launch {v.get('SQ_INSTS_LDS',0)/n/1e6:7.2f} M")
PY
