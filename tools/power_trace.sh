#!/bin/bash
# Clock / power of the GPU while a command runs: polls rocm-smi every 0.25 s beside `$@`, prints the samples and their summary.
#   tools/power_trace.sh <label> <command...>      -> gpurun_out/<label>_power.txt
# (reading the sensors needs no privileges; used for DESIGN 4.1.7: which clock the step's MFMA kernels actually run at)
label=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/${label}_power.txt
"$@" > gpurun_out/${label}_cmd.out 2> gpurun_out/${label}_cmd.err &
pid=$!
: > $out.raw
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n' >> $out.raw
  echo >> $out.raw
  sleep 0.25
done
wait $pid
python - "$out.raw" > $out <<'PY'
import json, sys, re
rows = []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get("card0", {})
    def num(pat):
        for k, v in c.items():
            if re.search(pat, k, re.I):
                m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                if m:
                    return float(m.group(0))
        return None
    rows.append((num(r"sclk"), num(r"mclk"), num(r"power"), num(r"temperature.*(hotspot|junction)") or num(r"temperature")))
print("# samples: sclk MHz, mclk MHz, power W, temperature C")
for r in rows:
    print("  ".join("%8s" % ("-" if x is None else "%.0f" % x) for x in r))
def col(i):
    return sorted(x[i] for x in rows if x[i] is not None)
for i, name in enumerate(["sclk MHz", "mclk MHz", "power W", "temp C"]):
    v = col(i)
    if v:
        print("# %-9s min %.0f  median %.0f  max %.0f  (n = %d)" % (name, v[0], v[len(v) // 2], v[-1], len(v)))
PY
rm -f $out.raw
tail -6 $out
