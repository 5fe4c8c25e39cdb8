for v in 64 256 512; do
  echo "== CVCS_HALO_NARROW_CIN=$v"
  CVCS_HALO_NARROW_CIN=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], 'halo', d['roofline']['achieved'], d['roofline']['ms_per_step'], 'enc', d['roofline_encoder']['achieved'], d['roofline_encoder']['ms_per_step'])"
done
echo "== CVCS_FUSE_BN_BWD=0"
CVCS_FUSE_BN_BWD=0 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'])"
