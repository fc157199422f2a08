#!/bin/bash
# usage: [MP=0] [BENCH_ARGS="--config 3 --dense"] pmc_traffic.sh TAG
# HBM traffic of one frame from the PMC counters (separate rocprofv3 --pmc passes, as
# MI355X_MICROARCH.md §HBM prescribes). Writes gpurun_out/${TAG}_traffic.json. The stream runs without box misses
# (--jump-every 0): every frame but a context's first takes the steady-state path, whose kernels the per-frame figures are for
# (a kernel launched for fewer than half of the frames — the first frames' fixed-grid passes — is left out).
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_$C
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/${TAG}_$C -- \
      python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --profile-frames 2 --jump-every 0 --min-pts ${MP:-2} ${BENCH_ARGS:-} > gpurun_out/${TAG}_$C.json 2> gpurun_out/${TAG}_$C.err || { tail -5 gpurun_out/${TAG}_$C.err; exit 1; }
done
python3 scripts/pmc_account.py "$TAG"
