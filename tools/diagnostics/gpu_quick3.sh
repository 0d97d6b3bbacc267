set -e
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 300
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 40
timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 12
