cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tl; rm -rf $O; mkdir -p $O; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python bench.py --steps 8 --warmup 1 --proofs-per-step 24 --threads 12 --no-cpu-baseline > /dev/null 2>&1
python tools/timeline_stats.py $O/tr > $O/timeline_12streams.txt 2>&1
rm -rf $O/tr
cat $O/timeline_12streams.txt
