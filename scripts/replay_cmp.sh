#!/bin/bash
# cfg4 replay, bucket path vs general path, three runs each (ROI + 10 cm, and 5 cm without ROI)
set -e
SEQ=/tmp/cfg4_seq
python -m cloud_merger_amd.replay_data $SEQ --frames 100 --sensors 4 > /dev/null
make -C cloud_merger_amd/host -s
for P in auto classic; do for i in 1 2 3; do
  echo -n "$P roi: "; CM_PATH=$P ./cloud_merger_amd/host/bin/cloudmerge_replay --dir $SEQ --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 --crop -15 -5 -0.5 60 5 3 | grep -o '"frames_per_s": [0-9.]*'
  echo -n "$P 5cm: "; CM_PATH=$P ./cloud_merger_amd/host/bin/cloudmerge_replay --dir $SEQ --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 | grep -o '"frames_per_s": [0-9.]*'
done; done
