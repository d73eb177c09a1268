#!/bin/bash
# Dev: A/B of dispatch knobs by the per-family times of bench.py's measuring passes (more stable than the step time)
for kv in none conv.glds3_pp_mink=512 conv.stream_percu=3 bn.vpt=4 conv.big_minblocks=200 wgrad.pp_mink=512; do
  if [ "$kv" = none ]; then T=""; else T="--tune $kv"; fi
  python bench.py --no-cpu-baseline $T 2>/dev/null | tail -1 > /tmp/l.json
  python - "$kv" <<'PY'
import json, sys
d = json.load(open('/tmp/l.json'))
f = d['roofline']['all_families']
tot = sum(v['ms_per_step'] for v in f.values())
keys = ['conv_glds3_kernel<pp>', 'conv_glds3_kernel', 'conv_glds3_kernel<splitk>', 'conv1x1_stream_kernel', 'bn_apply_kernel', 'conv_wgrad_pp_kernel', 'conv_wgrad_kernel<bf16>', 'conv_glds4_kernel<pp>']
print(f'{sys.argv[1]:28s} {d["value"]:7.1f} img/s  families {tot:6.2f} ms  ' + '  '.join(f'{k.replace("conv_","").replace("_kernel","")}={f[k]["ms_per_step"]:.2f}' for k in keys if k in f))
PY
done
