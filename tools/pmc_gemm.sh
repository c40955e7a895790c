# SQ counters of the contraction kernels, symmetric form and full product (one pass each, --kernel-trace only)
export TMPDIR=/tmp
O=gpurun_out/pmc_gemm
rm -rf $O; mkdir -p $O
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sym -- python3 tools/run_once.py 512 2 > $O/sym.log 2>&1
echo "sym done"
SOSRT_CONTRACT=full rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/full -- python3 tools/run_once.py 512 2 > $O/full.log 2>&1
echo "full done"
python3 - <<'PY'
import csv, glob, collections
for tag in ("sym", "full"):
    f = glob.glob("gpurun_out/pmc_gemm/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f:
        print(tag, "no counter file"); continue
    rows = list(csv.DictReader(open(f[0])))
    # dense launches of k_jn_gemm<...> only (grid 876544)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in rows:
        name = r["Kernel_Name"]
        if "k_jn_gemm" not in name: continue
        key = name.split("(")[0].replace("void sosrt::(anonymous namespace)::", "")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (key, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did); cnt[key] += 1
    for key, c in acc.items():
        n = cnt[key]
        print(tag, key, "dispatches", n, " ".join("%s=%.3g" % (k, v / n) for k, v in sorted(c.items())))
PY
rm -rf $O/sym $O/full
