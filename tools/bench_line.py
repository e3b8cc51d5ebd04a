#!/usr/bin/env python
"""One-line digest of a bench.py JSON line on stdin: tag, rounds/s (timed window), sustained rounds/s, tie-break resolutions, nodes, chain documents,
host milliseconds.  usage: python bench.py ... | python tools/bench_line.py <tag>"""
import json,sys
d=json.loads(sys.stdin.read()); tb=d["config"]["tie_break"]
print(sys.argv[1], round(d["value"],1), round(d["config"].get("sustained_rounds_per_s",0),1), tb["resolutions"], tb["nodes"], tb["chain_documents"], round(tb["host_ms"],1))
