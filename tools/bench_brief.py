"""Print the headline numbers of a bench.py JSON line read from stdin (diagnostic helper)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["kernel_us"])
