cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in VALUBusy "SQ_INSTS_VALU SQ_WAVES" FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_k2/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/time_sk.py 8192 512 6.283185307179586 > /dev/null 2>&1
  python tools/pmc_summary.py $d sk_tile_kernel
done
