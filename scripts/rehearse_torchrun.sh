export LIP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --probes 64 --samples 50 --no-cpu-baseline --no-resnet50 > gpurun_out/reh.json 2> gpurun_out/reh.err
tail -1 gpurun_out/reh.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['per_shard_products_per_s'], d['config']['examples_total'], d['config']['backend'], d['posterior_samples']['value'], d['scaling'])"
