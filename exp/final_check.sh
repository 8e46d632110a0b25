cd $GRAFT_REPO_ROOT
set -o pipefail
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu -k "rgat or packs or version or matmul" 2>&1 | tail -2 || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic'))"
