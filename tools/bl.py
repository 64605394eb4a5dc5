"""Condense a bench.py JSON line (stdin) to: label, GCUPS, kernel avg/min ms, kernel config."""
import json
import sys

d = json.loads(sys.stdin.read())
r = d["roofline"]
print(" ".join(sys.argv[1:]), d["config"]["pairs_per_gpu"], round(d["value"], 1), round(r["kernel_avg_ms"], 4),
      round(r["kernel_min_ms"], 4), d["config"]["kernel_config"][:70])
