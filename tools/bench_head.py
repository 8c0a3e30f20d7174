"""one line of a bench.py JSON line on stdin: ms/step, in-step roofline fraction, the first launches"""
import json
import sys
b = json.loads(sys.stdin.read())
print(b["ms_per_step"], b["roofline"]["frac"] if b.get("roofline") else None,
      [(l["name"][5:28], l["us"]) for l in (b.get("roofline") or {}).get("step_launches", [])][:2])
