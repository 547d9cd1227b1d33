# dev: parity tests of the fused kernel + C2 / C2 K=20 / C3 bench lines after a kernel change (tag = $1)
tag=$1
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || { tail -5 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
timeout -k 10 200 python bench.py --no-extra --no-cpu > gpurun_out/${tag}_c2.json 2>gpurun_out/${tag}_c2.err || exit 1
timeout -k 10 200 python bench.py --no-extra --no-cpu --topk 20 > gpurun_out/${tag}_c2k20.json 2>gpurun_out/${tag}_c2k20.err || exit 1
timeout -k 10 300 python bench.py --no-extra --no-cpu --config c3_mf_shard --steps 2 --warmup 1 > gpurun_out/${tag}_c3.json 2>gpurun_out/${tag}_c3.err || exit 1
python - <<PY
import json
for f in ["c2", "c2k20", "c3"]:
    d = json.loads(open("gpurun_out/${tag}_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4))
PY
