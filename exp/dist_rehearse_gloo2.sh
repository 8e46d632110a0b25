cd $GRAFT_REPO_ROOT
HET_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 2 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err; echo rc=$?; tail -c 700 gpurun_out/bench_gloo2.json; tail -3 gpurun_out/bench_gloo2.err
