python tools/ssbank_probe.py 64
PGX_SS_SEGS=1 python tools/ssbank_probe.py 64
PGX_SS_SEGS=2 python tools/ssbank_probe.py 64
PGX_SS_SEGS=6 python tools/ssbank_probe.py 64
python tools/ssbank_probe.py 256
