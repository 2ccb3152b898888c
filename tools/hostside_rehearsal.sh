# 8-GPU dress rehearsal of the HOST side on one device: `bench.py --gpus N` with GL355_BENCH_ONE_DEVICE=1 (every rank on cuda:0, exchange
# over TCP), the job confined by taskset to N x k CPUs, for the (ranks, cores per rank) pairs the box's CPU quota allows.  What it shows: which wait /
# replay mode each rank picks at k cores (bench.py main_recursive), the host CPU per unit in that mode, and that N ranks' host threads keep ONE device
# busy (the device is time-sliced, so the aggregate rate is one device's rate: the per-rank figure is a host-side figure only).
# usage (GPU box): bash tools/hostside_rehearsal.sh > gpurun_out/r06/hostside_8rank.txt
cd $GRAFT_REPO_ROOT
export GL355_BENCH_ONE_DEVICE=1
echo "# $(nproc) CPUs visible, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for cfg in "8 1" "8 2" "4 4" "2 8" "1 16"; do
  set -- $cfg; N=$1; K=$2; C=$((N*K-1))
  line=$(taskset -c 0-$C python bench.py --gpus $N --steps 3 --warmup 1 --proofs-per-step 96 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$N" "$K" "$line" <<'PY'
import json, sys
n, k, line = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
try:
    d = json.loads(line)
    c = d["config"]
    print("ranks %d x %2d cores: %7.1f units/s on the one device | mode %s | expected per GPU in this mode %s | host CPU %.2f ms/unit (rank 0) | exchange: %s"
          % (n, k, d["value"], json.dumps(c.get("host_mode")), c.get("expected_units_per_s_per_gpu"), c["host_cpu_ms_per_unit"], c["exchange"]))
except Exception as exc:
    print("ranks %d x %d cores: FAILED (%r) %s" % (n, k, exc, line[:200]))
PY
done
