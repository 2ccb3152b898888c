# Round-3 profile set (run on the GPU box through gpurun; summaries are copied into profiles/ by hand afterwards):
#   rocprofv3 --kernel-trace --stats of the default bench (8 contexts) and of one context, --pmc passes (separate runs per counter set, as
#   MI355X_MICROARCH.md prescribes) over one context for HBM traffic and VALU issue, the LDE and natural-NTT stall pictures, bench JSON lines.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd $R
python bench.py > $O/bench_recursive.json 2> $O/bench_recursive.err
python bench.py --workload lde > $O/bench_lde.json 2> $O/bench_lde.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python bench.py --steps 1 --warmup 1 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_rocprof_1stream.json 2> $O/stats1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lde -- python tools/prof_lde.py 80 > /dev/null 2> $O/stats_lde.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
  rocprofv3 --pmc $c --output-format csv -d $O/ldepmc_$c -- python tools/prof_lde.py > /dev/null 2> $O/ldepmc_$c.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_pmc_sq.json 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/ldepmc_sq -- python tools/prof_lde.py > /dev/null 2> $O/ldepmc_sq.err
bash tools/prof_lde_pmc.sh > $O/lde_pmc_sets.txt 2>&1
bash tools/prof_ntt_pmc.sh > $O/ntt_pmc_sets.txt 2>&1
python - <<'PY'
import csv, glob, os, collections, shutil
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_r03")
for d in ("pmc_FETCH_SIZE","pmc_WRITE_SIZE","pmc_sq","ldepmc_FETCH_SIZE","ldepmc_WRITE_SIZE","ldepmc_sq"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    with open(os.path.join(O,d+"_summary.txt"),"w") as out:
        for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
            out.write(k[:90]+": "+", ".join("%s=%.4g (n=%d)"%(c,v/cnt[(k,c)],cnt[(k,c)]) for c,v in agg[k].items())+"\n")
    shutil.rmtree(os.path.join(O,d),ignore_errors=True)
for s in ("stats","stats1","stats_lde"):
    for f in glob.glob(os.path.join(O,s,"**","*kernel_trace.csv"),recursive=True): os.remove(f)
    for f in glob.glob(os.path.join(O,s,"**","*_agent_info.csv"),recursive=True): os.remove(f)
PY
ls -R $O | head -40
