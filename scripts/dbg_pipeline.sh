python -m cloud_merger_amd.replay_data /tmp/cfg4_seq --frames 100 --sensors 4 > /dev/null
R=./cloud_merger_amd/host/bin/cloudmerge_replay
for i in 1 2 3; do CM_NODE_TRACE=1 $R --dir /tmp/cfg4_seq --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 --crop -15 -5 -0.5 60 5 3 --pipeline 2>&1 | cut -c1-400; done
