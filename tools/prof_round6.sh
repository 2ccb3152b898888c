# Round-6 profile set (run on the GPU box through gpurun: `bash tools/prof_round6.sh [probe|recursive|lde|all]`); summaries land in gpurun_out/prof_r06 and are
# copied into profiles/ afterwards.  rocprofv3 --kernel-trace --stats and the --pmc passes are SEPARATE runs (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
WHAT=${1:-all}
mkdir -p $O
cd $R
# steady state of the PMC pass: the same one-context job with 1 step and with 3 steps; (3 steps) - (1 step) = two steps' kernels without the set-up
# (key hashes, group tree, the two circuits' preprocessed commitments), VERDICT r5 #1 (c)
ONE="python bench.py --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline"
if [ $WHAT = probe ] || [ $WHAT = all ]; then
  python tools/valu_probe_run.py > $O/valu_probe.json 2> $O/valu_probe.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES --output-format csv -d $O/pmc_probe -- python tools/valu_probe_run.py > $O/valu_probe_under_pmc.json 2> $O/pmc_probe.err
fi
if [ $WHAT = recursive ] || [ $WHAT = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python bench.py --steps 1 --warmup 1 --proofs-per-step 16 --threads 1 --no-cpu-baseline > $O/bench_under_rocprof_1stream.json 2> $O/stats1.err
  for c in FETCH_SIZE WRITE_SIZE; do
    GL355_BENCH_NO_AGGREGATE=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $ONE --steps 1 > /dev/null 2> $O/pmc_$c.err
  done
  for n in 1 3; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq$n -- $ONE --steps $n > $O/bench_under_pmc_sq$n.json 2> $O/pmc_sq$n.err
  done
fi
if [ $WHAT = lde ] || [ $WHAT = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lde -- python tools/prof_lde.py 80 > /dev/null 2> $O/stats_lde.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/ldepmc_$c -- python tools/prof_lde.py > /dev/null 2> $O/ldepmc_$c.err
  done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/ldepmc_sq -- python tools/prof_lde.py > /dev/null 2> $O/ldepmc_sq.err
fi
python - <<'PY'
import csv, glob, os, collections, shutil
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_r06")
for d in ("pmc_FETCH_SIZE","pmc_WRITE_SIZE","pmc_sq1","pmc_sq3","pmc_probe","ldepmc_FETCH_SIZE","ldepmc_WRITE_SIZE","ldepmc_sq"):
    if not os.path.isdir(os.path.join(O,d)): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    with open(os.path.join(O,d+"_summary.txt"),"w") as out:
        for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
            out.write(k[:110]+": "+", ".join("%s=%.6g (n=%d)"%(c,v/cnt[(k,c)],cnt[(k,c)]) for c,v in agg[k].items())+"\n")
    shutil.rmtree(os.path.join(O,d),ignore_errors=True)
for s in ("stats","stats1","stats_lde"):
    for f in glob.glob(os.path.join(O,s,"**","*kernel_trace.csv"),recursive=True): os.remove(f)
    for f in glob.glob(os.path.join(O,s,"**","*_agent_info.csv"),recursive=True): os.remove(f)
PY
ls -R $O | head -60
