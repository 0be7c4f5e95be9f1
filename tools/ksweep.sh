#!/bin/bash
# k sweep of the global counting path on 10 M synthetic reads (diagnostics): ms/step, k-mers/s,
# distinct keys, level-2 record totals / max stream / capacity / spills
for k in "$@"; do
  CFRK_BENCH_INFO=1 python bench.py --reads 10000000 --k $k --steps 2 --warmup 1 --cpu-reads 0 2>/dev/null > /tmp/ks.json
  python3 - "$k" <<'PY'
import json,sys
d=json.loads(open('/tmp/ks.json').readline()); i=d.get("msp_info") or {}
print("k=%s ms=%.2f kmers/s=%.3g ok=%s D=%d rec=%s maxstream=%s cap=%s spilled=%s/%s" % (sys.argv[1], d["ms_per_step"], d["value"], d["sum_count_ok"], d["distinct"], i.get("l2_records"), i.get("l2_max_leaf"), i.get("l2_cap"), i.get("spilled_records"), i.get("spilled_kmers")))
PY
done
