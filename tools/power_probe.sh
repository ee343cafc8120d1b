#!/bin/bash
# GPU power / clock while the headline loop runs (rocm-smi polled beside bench.py): is the chip at its power cap?
python bench.py --steps 120 --warmup 2 --no-cpu-baseline --no-roofline --no-parity "$@" > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower --json 2>/dev/null | tr -d '\n' ; echo
  sleep 0.2
done > gpurun_out/power_samples.txt
wait $BP
python - <<'PY'
import json,re
vals=[];clk=[];cap=None
for ln in open('gpurun_out/power_samples.txt'):
    ln=ln.strip()
    if not ln.startswith('{'): continue
    try: d=json.loads(ln)
    except Exception: continue
    for c in d.values():
      for k,v in c.items():
        if 'Power' in k and 'Max' not in k and 'Cap' not in k:
            try: vals.append(float(v))
            except: pass
        if 'Max Graphics Package Power' in k or 'Cap' in k:
            try: cap=float(v)
            except: pass
        if k.startswith('sclk clock speed'):
            m=re.search(r'(\d+)Mhz',str(v))
            if m: clk.append(int(m.group(1)))
busy=[v for v in vals if v>400]
print('samples',len(vals),'power W min/max',(min(vals),max(vals)) if vals else None,'| while loaded (>400 W): n',len(busy),'mean',round(sum(busy)/len(busy),1) if busy else None,'max',max(busy) if busy else None,'| cap',cap,'| sclk MHz min/max',(min(clk),max(clk)) if clk else None)
print(sorted(clk)[-12:])
print(open('gpurun_out/power_bench.json').read()[:200])
PY
head -c 600 gpurun_out/power_samples.txt
