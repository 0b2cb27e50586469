export TMPDIR=/tmp
P=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
echo "== torch.distributed.run, 2 ranks on the one GPU (port $P)"
time (timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 3 --warmup 1 --docs 3000 --vocab 2000 --topics 20 --cpu-sample 0 > gpurun_out/torchrun.out 2> gpurun_out/torchrun.err)
echo rc $?; tail -1 gpurun_out/torchrun.out | cut -c1-600; tail -5 gpurun_out/torchrun.err | cut -c1-300
echo "== same, full-size weak (what the driver would launch at N=2, on one GPU: host reduction)"
P=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
time (timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/torchrun2.out 2> gpurun_out/torchrun2.err)
echo rc $?; tail -1 gpurun_out/torchrun2.out | cut -c1-700; tail -3 gpurun_out/torchrun2.err | cut -c1-300
