cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python bench.py --steps 1 --warmup 1 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_rocprof_1stream.json 2> $O/stats1.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
# keep only the small summaries: stats csv + per-kernel aggregation of the counter csvs
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get("O") or os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_r02")
for d in ("pmc_FETCH_SIZE","pmc_WRITE_SIZE","pmc_sq"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); 
            cnt[(k,row["Counter_Name"])]+=1
    with open(os.path.join(O,d+"_summary.txt"),"w") as out:
        for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
            out.write(k[:90]+": "+", ".join("%s=%.4g (n=%d)"%(c,v/cnt[(k,c)],cnt[(k,c)]) for c,v in agg[k].items())+"\n")
import shutil
for d in ("pmc_FETCH_SIZE","pmc_WRITE_SIZE","pmc_sq"):
    shutil.rmtree(os.path.join(O,d),ignore_errors=True)
for s in ("stats","stats1"):
    for f in glob.glob(os.path.join(O,s,"**","*kernel_trace.csv"),recursive=True): os.remove(f)
PY
ls -R $O | head -30
