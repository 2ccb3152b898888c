# ONE rank confined to k CPUs drives the whole device: units/s per GPU at k usable cores per rank -- the figure an 8-GPU node's core count has to be read
# against (8 ranks x k cores; the ranks of a node share nothing else on the host side).  bench.py picks the rank's wait / replay mode from its usable cores.
# usage (GPU box): bash tools/hostside_one_rank.sh > gpurun_out/r06/hostside_one_rank.txt
cd $GRAFT_REPO_ROOT
for K in 1 2 3 4 6 8 12 16; do
  line=$(taskset -c 0-$((K-1)) python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$K" "$line" <<'PY'
import json, sys
k, line = int(sys.argv[1]), sys.argv[2]
try:
    d = json.loads(line)
    c = d["config"]
    print("1 rank x %2d cores: %7.1f units/s | mode %s | expected %s | host CPU %.2f ms/unit | latency one unit %.2f ms"
          % (k, d["value"], json.dumps(c.get("host_mode")), c.get("expected_units_per_s_per_gpu"), c["host_cpu_ms_per_unit"], d["latency_single_unit_ms"]["median_ms"]))
except Exception as exc:
    print("1 rank x %d cores: FAILED (%r) %s" % (k, exc, line[:200]))
PY
done
