python tools/ssbank_probe.py 64
PGX_SS_SEGS=1 python tools/ssbank_probe.py 64 | head -3
