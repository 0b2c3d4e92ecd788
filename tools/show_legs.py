"""Print chosen extra legs of a bench.py log (the JSON line) without their `what` texts.
Usage: python tools/show_legs.py LOG [leg ...]"""
import json
import sys


def strip(v):
    if isinstance(v, dict):
        return {a: strip(b) for a, b in v.items() if a != "what"}
    return v


for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        j = json.loads(line)
        print(json.dumps({k: j[k] for k in ("value", "ms_per_step")}), json.dumps(strip(j["roofline"])))
        for k, v in j.get("extra_legs", {}).items():
            if len(sys.argv) < 3 or k in sys.argv[2:]:
                print(k, json.dumps(strip(v)))
