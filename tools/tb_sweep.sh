for tb in 1082 2164 3246 4096; do
  IFR_TARGET_BLOCKS=$tb timeout -k 10 120 python bench.py --no-extra --no-cpu --steps 20 --warmup 3 > gpurun_out/tb_$tb.json 2> gpurun_out/tb_$tb.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/tb_$tb.json").read().strip().splitlines()[-1])
print($tb, d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("kernel_ms"))
PY
done
