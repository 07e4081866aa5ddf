timeout 300 python tools/_diag_align.py 2>&1 | tail -30
