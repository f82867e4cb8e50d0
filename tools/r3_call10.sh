python tools/ssbank_probe.py 64 | head -2
PGX_SS_SEG_NW=4 python tools/ssbank_probe.py 64 | head -2
PGX_SS_SEG_NW=4 PGX_SS_SEGS=6 python tools/ssbank_probe.py 64 | head -2
PGX_SS_SEG_NW=4 PGX_SS_SEGS=12 python tools/ssbank_probe.py 64 | head -2
timeout -k 10 200 python tools/shard_probe.py supersaw | tail -2
PGX_SS_SEG_NW=4 timeout -k 10 200 python tools/shard_probe.py supersaw | tail -2
