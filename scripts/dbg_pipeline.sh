python -m cloud_merger_amd.replay_data /tmp/cfg4_seq --frames 100 --sensors 4 > /dev/null
R=./cloud_merger_amd/host/bin/cloudmerge_replay
$R --dir /tmp/cfg4_seq --sensors 4 --frames 100 --leaf 0.05 --min-pts 2 | cut -c1-600
$R --dir /tmp/cfg4_seq --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 | cut -c1-600
$R --dir /tmp/cfg4_seq --sensors 4 --frames 100 --leaf 0.1 --min-pts 2 --crop -15 -5 -0.5 60 5 3 | cut -c1-600
CM_VERBOSE=1 $R --dir /tmp/cfg4_seq --sensors 4 --frames 30 --leaf 0.05 --min-pts 2 2>&1 | grep cloudmerge | head -5
