#!/bin/bash
# BASELINE config 4: 4 sensors, 100 frames @10 Hz recorded as .pcd (synthesised: the reference ships
# no data), replayed through the C++ CloudMergerNode. One GPU here; `--shard r/N` per rank on N GPUs.
# Twice each: everything on one thread (submit x4, fuse, publish, one after the other) and with the reference's
# threading (--threads: one subscriber thread per sensor beside the loop thread; the slots are double-buffered).
set -e
SEQ=/tmp/cfg4_seq
python -m cloud_merger_amd.replay_data $SEQ --frames 100 --sensors 4 > /dev/null
make -C cloud_merger_amd/host -s
for T in "" "--threads" "--threads --repeat 4"; do
  for rep in 1 2 3; do
    ./cloud_merger_amd/host/bin/cloudmerge_replay --dir $SEQ --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 $T
    ./cloud_merger_amd/host/bin/cloudmerge_replay --dir $SEQ --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 --crop -15 -5 -0.5 60 5 3 $T
  done
done
