# per-kernel time + HBM-side traffic of ONE kernel of the recursive workload (one context, lock-step batches of 8):
#   bash tools/prof_kernel.sh quotient_kernel      -> prints average duration, FETCH_SIZE, WRITE_SIZE per launch
K=${1:-quotient_kernel}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=/tmp/pk; rm -rf $O; mkdir -p $O; cd $R
CMD="python bench.py --steps 1 --warmup 0 --proofs-per-step 16 --threads 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- $CMD > /dev/null 2> $O/st.err
grep -h "$K" $O/st/*/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- $CMD > /dev/null 2> $O/$c.err
  python - "$O/$c" "$K" "$c" <<'P'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]:
            tot += float(r["Counter_Value"]); n += 1
print("%s per launch of %s: %.1f MB reported (n=%d)%s" % (sys.argv[3], sys.argv[2], tot / max(n, 1) / 1024, n, "  x2 on gfx950 for the HBM-side figure" if sys.argv[3] == "FETCH_SIZE" else ""))
P
done
