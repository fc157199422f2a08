# the fixed-grid passes alone (CM_QUANT=0), to compare k3_local<..., false, false> with round 2's figure
export CM_QUANT=0
bash scripts/kstats_run.sh $1
