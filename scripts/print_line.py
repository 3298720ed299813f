import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), "%.4g" % d["value"], {k:round(v,4) for k,v in d["roofline"]["stage_ms"].items()})
