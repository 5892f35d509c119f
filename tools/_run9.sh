mkdir -p gpurun_out/r04
for rep in 1 2; do
  for tree in new old; do
    if [ $tree = old ]; then cd ab_old; fi
    python tools/model_bench.py --precisions f32 f16x3 --out /tmp/pm_${tree}_$rep.json > /dev/null 2>/tmp/pm_${tree}_$rep.err || tail -5 /tmp/pm_${tree}_$rep.err
    if [ $tree = old ]; then cd ..; fi
    python - <<PY
import json
d=json.load(open('/tmp/pm_${tree}_$rep.json'))
rows=d if isinstance(d,list) else d.get('rows',d)
print('$tree', $rep)
for r in (rows if isinstance(rows,list) else rows.values()):
    print('   ', {k:(round(v,2) if isinstance(v,float) else v) for k,v in r.items() if k in ('config','name','precision','images_per_s','ms_per_call','conv_ms','other_ms','conv_tflops','model','case')})
PY
  done
done
