python tools/pin_probe.py
