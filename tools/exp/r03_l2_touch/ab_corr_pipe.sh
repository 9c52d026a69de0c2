mkdir -p gpurun_out
for i in 1 2; do
  for v in 0 1; do
    echo "VFML_CORR_PIPE=$v" 
    VFML_CORR_PIPE=$v timeout -k 10 200 python bench.py --no-jobs --no-roofline --no-cpu-baseline --steps 40 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('plans',{}).get('f16x3',{}).get('value'))" || exit 1
  done
done
