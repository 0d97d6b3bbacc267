set -e
python tools/diagnostics/rank_frames.py 1 0 100
python tools/diagnostics/rank_frames.py 2 1 100
python tools/diagnostics/rank_frames.py 4 3 100
python tools/diagnostics/rank_frames.py 8 0 100
python tools/diagnostics/rank_frames.py 8 7 100
