# Round-4 profile set (run on the GPU box through gpurun; summaries are copied into profiles/ by hand afterwards):
#   bench lines, rocprofv3 --kernel-trace --stats of the default bench and of one context, --pmc passes (separate runs per counter set, as
#   MI355X_MICROARCH.md prescribes) over one context: HBM traffic (FETCH_SIZE / WRITE_SIZE) and the VALU pass with the dynamic class counters
#   (SQ_INSTS_VALU_INT32 / _INT64), and the same counters over the probe kernels of csrc/valu_probe.hip (what the counters call each class).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd $R
python tools/valu_probe_run.py > $O/valu_probe.json 2> $O/valu_probe.err
python bench.py > $O/bench_recursive.json 2> $O/bench_recursive.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python bench.py --steps 1 --warmup 1 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_rocprof_1stream.json 2> $O/stats1.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
SQC="SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
rocprofv3 --pmc $SQC --output-format csv -d $O/pmc_sq -- python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_pmc_sq.json 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc_probe -- python tools/valu_probe_run.py > $O/valu_probe_under_pmc.json 2> $O/pmc_probe.err
python - <<'PY'
import csv, glob, os, collections, shutil
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_r04")
for d in ("pmc_FETCH_SIZE","pmc_WRITE_SIZE","pmc_sq","pmc_probe"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    with open(os.path.join(O,d+"_summary.txt"),"w") as out:
        for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
            out.write(k[:90]+": "+", ".join("%s=%.4g (n=%d)"%(c,v/cnt[(k,c)],cnt[(k,c)]) for c,v in agg[k].items())+"\n")
    shutil.rmtree(os.path.join(O,d),ignore_errors=True)
for s in ("stats","stats1"):
    for f in glob.glob(os.path.join(O,s,"**","*kernel_trace.csv"),recursive=True): os.remove(f)
    for f in glob.glob(os.path.join(O,s,"**","*_agent_info.csv"),recursive=True): os.remove(f)
PY
ls -R $O | head -40
