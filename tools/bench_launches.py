"""bench.py JSON line on stdin -> ms/step and every launch of the step (name, us) from roofline.step_launches"""
import json
import sys
b = json.loads(sys.stdin.read())
print(b["ms_per_step"], " ".join(f"{l['name'][5:].replace('_f32', '')}={l['us']:.1f}"
                                 for l in (b.get("roofline") or {}).get("step_launches", [])))
