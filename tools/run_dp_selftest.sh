# bench.py's data-parallel lines through its self-test hooks: two ranks on ONE GPU over gloo (functional evidence of the N > 1
# plumbing and of the row-block step -- NOT performance numbers: both ranks share the device and the collectives run on the host)
cd $GRAFT_REPO_ROOT
export EMX_BENCH_ONE_DEVICE=1 EMX_BENCH_BACKEND=gloo
for extra in "--dp-mode replica" "--dp-mode rowblock" "--dp-mode rowblock --global-batch 0"; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 100 --warmup 20 $extra 2>/dev/null | tail -1
done
