"""One line per secondary workload of a bench.py JSON: value, ms per frame, traversal kernel times, parity flag.
usage: python tools/bench_line.py gpurun_out/some_bench.json"""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,s in d["secondary"].items(): print(sys.argv[1], k, round(s["value"],1), s.get("ms_per_frame"), s.get("trace_closest_ms"), s.get("trace_any_ms"), s["parity"].get("parity_sample_ok"))
