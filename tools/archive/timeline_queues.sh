cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tl; mkdir -p $O; cd $GRAFT_REPO_ROOT
for q in 8 16; do
  rm -rf $O/tr
  GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python bench.py --steps 8 --warmup 1 --proofs-per-step 24 --threads 12 --no-cpu-baseline > $O/b$q.json 2>/dev/null
  echo "GPU_MAX_HW_QUEUES=$q"; python tools/timeline_stats.py $O/tr 2>&1 | head -3
  python -c "import json;print(json.loads(open('$O/b$q.json').read().strip().splitlines()[-1])['value'])"
done
rm -rf $O/tr
