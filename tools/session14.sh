#!/bin/bash
# round 2, sixth session: bench.py's control plane — the RCCL probe's fallback to gloo (two ranks on one GPU: RCCL refuses), and the
# driver's torchrun form with one rank
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6g
mkdir -p $O
timeout -k 10 200 python bench.py --gpus 2 --rehearse-one-gpu --try-rccl --batch 16 --steps 2 --warmup 1 > $O/bench_2rank_rccl_refused.json 2> $O/bench_2rank_rccl_refused.err; echo "rc $?"
tail -c 1500 $O/bench_2rank_rccl_refused.json; tail -5 $O/bench_2rank_rccl_refused.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "rc $?"
cut -c1-300 $O/bench_torchrun_1rank.json
