#!/usr/bin/env python3
"""Print selected fields of a bench.py JSON line read from stdin: python bench.py | python scripts/bench_field.py"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(d["ms_per_step"], d["value"], r.get("achieved"), r.get("launches_per_iteration"), r.get("avg_launch_us"))
