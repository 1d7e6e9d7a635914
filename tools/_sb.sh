#!/bin/bash
# usage: _sb.sh n shape [extra env...]: prints the key numbers of one seed_bench run
python tools/seed_bench.py $1 $2 3 16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', 'identical', d['identical'], '/', d['compared'], 'lib_s', d['library_s'], 'cold', d['library_cold_s'], 'ref_s', d['reference_s'], 'marshal', d['marshal_s'], 'batches', d['batches'], 'upload_ms', d['upload_ms'], 'idle_ms', d['walks_ms'], 'lib/s', d['library_pairs_per_s'], 'ref/s', d['reference_pairs_per_s'])"
