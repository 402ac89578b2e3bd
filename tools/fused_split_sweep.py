import json, os, sys
sys.path.insert(0, os.getcwd())
from tools.split_variants_check import timing_fused
res = {}
for n in (32768, 65536, 98304, 131072, 196608, 262144):
    res[n] = {s: [timing_fused(s, n), timing_fused(s, n)] for s in (0, 2, 4)}
    print(n, res[n], file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
