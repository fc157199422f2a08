#!/bin/bash
# usage: soak_final.sh TAG — the round's differential runs against the oracle once more, on the build that is measured and shipped
# (about 15 minutes of GPU box time; one line per run in gpurun_out/TAG_soak.txt).
T=${1:-soak}
mkdir -p gpurun_out
O=gpurun_out/${T}_soak.txt; : > $O
run() { echo "== $*" >> $O; timeout -k 10 "$1" "${@:2}" 2>&1 | tail -2 >> $O || { echo "FAILED: $*" >> $O; tail -20 $O; exit 1; }; echo "$(date +%T) done: ${*:2}"; }
run 200 python3 scripts/fuzz_parity.py 150 61 0.0 0.2
run 150 python3 scripts/fuzz_parity.py 100 62 0.15 0.5
run 150 env CM_LDS_RANK=0 python3 scripts/fuzz_parity.py 100 63 0.0 0.2
run 150 python3 scripts/fuzz_concurrent.py 100 11
run 150 python3 scripts/fuzz_api.py 100 5
run 150 python3 scripts/fuzz_fused.py 80 7
run 300 python3 scripts/fuzz_shared_bins.py 200 1
run 200 python3 scripts/fuzz_quantile_stream.py 120 1
cat $O
