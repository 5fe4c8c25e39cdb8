# usage: ab.sh "ENV1=.. ENV2=.." ...   -> bench line summary per environment setting
for e in "$@"; do
  env $e python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], 'conv', d['roofline']['achieved'], 'wgrad', d['roofline_wgrad']['achieved'])"
done
