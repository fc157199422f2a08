#!/bin/bash
# Small frames / the live node: cfg4's shape (4 sensors x 120 k points) and the reference node's own
# configuration (--live: six sensors, ROI, 10 cm, min 2 points, zone-wise ground removal + per-slab outlier filter) through
# cloudmerge_replay from host .pcd payloads: one thread (tick latency p50 / p99), subscriber threads, subscriber threads +
# pipelined publish, the deferred wait on top of it (--defer: frame n waited for during tick n + 1), DMA-able input buffers (--pin), submits that do not wait for their copy (--async). usage: live_node.sh TAG   -> gpurun_out/TAG_live_node.jsonl, TAG_live_node_kernels.txt
TAG=${1:-live}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SEQ4=/tmp/cfg4_seq; SEQ6=/tmp/live_seq
python -m cloud_merger_amd.replay_data $SEQ4 --frames 100 --sensors 4 > /dev/null
python -m cloud_merger_amd.replay_data $SEQ6 --frames 60 --sensors 6 > /dev/null
make -C cloud_merger_amd/host -s
R=./cloud_merger_amd/host/bin/cloudmerge_replay
OUT=gpurun_out/${TAG}_live_node.jsonl; : > $OUT
run() { echo "# $*" >> $OUT; for rep in 1 2 3; do timeout -k 5 60 "$@" >> $OUT; done; }
CROP="--crop -15 -5 -0.5 60 5 3"
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --pipeline
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --pipeline --pin
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --defer
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --defer --pin
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --threads --repeat 4
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --threads --repeat 4 --pipeline
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --threads --repeat 4 --pipeline --pin --async
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 $CROP --threads --repeat 4 --defer --pin --async
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 --threads --repeat 4
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 --threads --repeat 4 --pipeline
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 --threads --repeat 4 --defer --pin --async
run $R --dir $SEQ4 --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 --defer --pin
run $R --dir $SEQ6 --live --frames 60
run $R --dir $SEQ6 --live --frames 60 --threads --repeat 4
grep -v "^#" $OUT | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-92s %8.0f frames/s  tick p50 %s p99 %s' % (d.get('mode','')[:92], d.get('steady_frames_per_s', d['frames_per_s']), d.get('tick_ms_p50','-'), d.get('tick_ms_p99','-')))
"
# kernels per tick of the live configuration
rm -rf gpurun_out/${TAG}_live_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_live_prof -- $R --dir $SEQ6 --live --frames 60 > /dev/null 2>&1 || true
f=$(find gpurun_out/${TAG}_live_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > gpurun_out/${TAG}_live_node_kernels.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot_calls = sum(int(r["Calls"]) for r in rows); tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
print("launches per tick %.1f, kernel time per tick %.1f us (60 ticks)" % (tot_calls / 60.0, tot_ns / 60.0 / 1e3))
for r in rows[:25]:
    m = re.search(r"(k[234g]?_\w+(<[\w, ]+>)?|__amd\w+)", r["Name"]); nm = m.group(1) if m else r["Name"][:40]
    print("%-44s calls %5s avg_us %8.2f" % (nm, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat gpurun_out/${TAG}_live_node_kernels.txt 2>/dev/null | head -30
