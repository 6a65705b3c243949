set -e
OUT=gpurun_out/hwq; mkdir -p $OUT
B="bench.py --no-cpu-baseline --no-single-scan --steps 100 --warmup 20"
run() { tag=$1; shift; timeout -k 10 120 "$@" > $OUT/$tag.log 2> $OUT/$tag.err; python3 - $OUT/$tag.log $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "| value %.0f" % d["value"], "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "fit %.4f" % d["roofline"]["fitness"]["ms"])
PY
}
run q_default python3 $B
GPU_MAX_HW_QUEUES=2 run q2 python3 $B
GPU_MAX_HW_QUEUES=8 run q8 python3 $B
GPU_MAX_HW_QUEUES=8 run q8_in2 python3 $B --inflight 2
GPU_MAX_HW_QUEUES=8 run q8_in3 python3 $B --inflight 3
GPU_MAX_HW_QUEUES=16 run q16_in2 python3 $B --inflight 2
GPU_MAX_HW_QUEUES=16 run q16_in3h0 python3 $B --inflight 3 --max-helpers 0
run qd_in2 python3 $B --inflight 2
