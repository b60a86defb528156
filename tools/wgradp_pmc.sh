cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=$R/gpurun_out/wgp_pmc_$(echo $c | tr ' ' '_')
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d $d -- $R/tools/exp_wgradp 65536 784 784 > $R/gpurun_out/wgp_pmc.log 2>&1 || exit 1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $f
done
