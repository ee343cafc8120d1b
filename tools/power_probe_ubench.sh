#!/bin/bash
# marginal power of the consumer loop's ingredients: tools/ubench/consumer_loop <variant> on all CUs, rocm-smi sampled beside it
for v in ${@:-mfma mfma_valu4 loop_lds_only loop_w_only loop loop_rnd loop_sibling}; do
  timeout -k 5 60 tools/ubench/consumer_loop $v 5 > gpurun_out/pp_$v.txt 2>&1 &
  BP=$!
  sleep 1.5
  : > gpurun_out/pp_$v.smi
  while kill -0 $BP 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> gpurun_out/pp_$v.smi; echo >> gpurun_out/pp_$v.smi
    sleep 0.25
  done
  wait $BP
  python - $v <<'PY'
import json,re,sys
v=sys.argv[1]; pw=[]; ck=[]
for ln in open(f'gpurun_out/pp_{v}.smi'):
    ln=ln.strip()
    if not ln.startswith('{'): continue
    try: d=json.loads(ln)
    except Exception: continue
    for c in d.values():
        for k,x in c.items():
            if 'Power' in k:
                try: pw.append(float(x))
                except: pass
            if k.startswith('sclk clock speed'):
                m=re.search(r'(\d+)Mhz',str(x))
                if m: ck.append(int(m.group(1)))
busy=[p for p in pw if p>350]
print(f"{v:16s} power while loaded: mean {sum(busy)/max(len(busy),1):7.1f} W max {max(busy) if busy else 0:7.1f} (n={len(busy)}); sclk max {max(ck) if ck else 0} MHz; ", open(f'gpurun_out/pp_{v}.txt').read().strip())
PY
done
