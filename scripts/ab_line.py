"""One line of scripts/ab_bench.sh: variant, scene, throughput and the stage times of a bench.py output file."""
import json, sys
name, sc, path = sys.argv[1:4]
j = json.load(open(path))  # the detail file of the run (bench.py --detail)
k = j["kernel_ms_per_step"]
print(f"{name:>14s} {sc:8s} {j['value']:8.1f} Msamples/s {j['ms_per_step']:9.1f} ms | closest {k['trace_closest']:7.1f} any {k['trace_any']:6.1f} mis {k['trace_mis']:6.1f} "
      f"shade {k['shade']:7.1f} resolve {k['resolve']:5.1f} raygen {k['raygen']:5.1f} film {k['film']:5.1f} | sclk {((j.get('gpu_clocks') or {}).get('sclk_MHz') or {}).get('after')} W {((j.get('gpu_clocks') or {}).get('power_W') or {}).get('after')}", flush=True)
