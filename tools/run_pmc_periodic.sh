# counters of the periodic all-pairs tile kernel at the C4 shape: float form (no span given), phase form (span given)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for form in float phase; do
  unset EMX_TIME_SK_SPAN EMX_SK_PHASE
  if [ $form != float ]; then export EMX_TIME_SK_SPAN=6.2831; fi
  echo "== $form"
  for c in "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "LDSBankConflict" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    d=gpurun_out/pmc_per_$form/$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/time_sk.py 8192 512 6.283185307179586 > $d.log 2>&1
    python3 tools/pmc_summary.py $d sk_tile_kernel
  done
done
