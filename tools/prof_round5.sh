# Round-5 profile set (run on the GPU box through gpurun: `bash tools/prof_round5.sh [recursive|lde|all]`); summaries land in gpurun_out/prof_r05 and are
# copied into profiles/ afterwards.  rocprofv3 --kernel-trace --stats and the --pmc passes are SEPARATE runs (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
WHAT=${1:-all}
mkdir -p $O
cd $R
ONE="python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline"
if [ $WHAT != lde ]; then
  python tools/valu_probe_run.py > $O/valu_probe.json 2> $O/valu_probe.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python bench.py --steps 1 --warmup 1 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_rocprof_1stream.json 2> $O/stats1.err
  for c in FETCH_SIZE WRITE_SIZE; do
    GL355_BENCH_NO_AGGREGATE=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $ONE > /dev/null 2> $O/pmc_$c.err
  done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- $ONE > $O/bench_under_pmc_sq.json 2> $O/pmc_sq.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc_probe -- python tools/valu_probe_run.py > $O/valu_probe_under_pmc.json 2> $O/pmc_probe.err
fi
if [ $WHAT != recursive ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lde -- python tools/prof_lde.py 80 > /dev/null 2> $O/stats_lde.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/ldepmc_$c -- python tools/prof_lde.py > /dev/null 2> $O/ldepmc_$c.err
  done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/ldepmc_sq -- python tools/prof_lde.py > /dev/null 2> $O/ldepmc_sq.err
fi
python - <<'PY'
import csv, glob, os, collections, shutil
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_r05")
for d in ("pmc_FETCH_SIZE","pmc_WRITE_SIZE","pmc_sq","pmc_probe","ldepmc_FETCH_SIZE","ldepmc_WRITE_SIZE","ldepmc_sq"):
    if not os.path.isdir(os.path.join(O,d)): continue
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
ls -R $O | head -60
