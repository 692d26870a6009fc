#!/bin/bash
# Run ON the GPU box (one GPU): rehearsal of bench.py's N-rank path with every rank on device 0 (DATUM_BENCH_DEVICES=1, --rendezvous gloo)
mkdir -p gpurun_out/r04q
export DATUM_BENCH_DEVICES=1
{
echo "== own launcher, 2 ranks, no gather"
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --rendezvous gloo --gather none --cpu-seconds 0 --no-frame --no-regime 2> gpurun_out/r04q/a.err; echo "exit $?"
echo "== torch.distributed.run, 2 ranks, no gather"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --warmup 5 --rendezvous gloo --gather none --cpu-seconds 0 --no-frame --no-regime 2> gpurun_out/r04q/b.err; echo "exit $?"
echo "== own launcher, 4 ranks, no gather, 2048^2 x 1 (configs[3]'s tile)"
timeout 300 python bench.py --gpus 4 --resolution 2048 --cascades 1 --steps 20 --warmup 5 --rendezvous gloo --gather none --cpu-seconds 0 --no-frame --no-regime 2> gpurun_out/r04q/c.err; echo "exit $?"
echo "== own launcher, 2 ranks, pipelined gather: RCCL must refuse two ranks on one GPU, bench.py must fail cleanly"
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --rendezvous gloo --cpu-seconds 0 --no-frame --no-regime 2> gpurun_out/r04q/d.err; echo "exit $?"; grep -h "OceanError\|ranks failed" gpurun_out/r04q/d.err | head -4
} > gpurun_out/r04q/rehearsal.txt 2>&1
cat gpurun_out/r04q/rehearsal.txt | cut -c1-900
