for IL in 0 1; do
echo "== interleave $IL, fb15k one-sided"; KGE_V4_INTERLEAVE=$IL python tools/one_sided.py --steps 200
echo "== interleave $IL, wikidata5m forced-dist"; KGE_V4_INTERLEAVE=$IL KGE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --repeats 3 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wikidata ms/step', d['ms_per_step'], 'launch us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'fb15k weak launch ms', d['fb15k_weak']['scoring_launch_ms'])"
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "bf16 or c2_full or c5" 2>&1 | tail -2
KGE_V4_INTERLEAVE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "bf16 or c2_full or c5" 2>&1 | tail -2
