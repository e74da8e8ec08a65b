for v in "" "NNPOPS_ANI_FWD_APG=3" "NNPOPS_ANI_FWD_WPA=1" "NNPOPS_ANI_FWD_WPA=1 NNPOPS_ANI_FWD_APG=2" "NNPOPS_ANI_FWD_ROWLDS=0" "NNPOPS_ANI_STORE=0"; do
  echo "== $v"
  env $v python tools/probe_ani.py --lib tools/_probe/libnnpops_hip.so --rounds 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for p in d['probes']:
    if p['mask'] in (0,64,128,192,448): print(p['mask'], p['us']['angular_forward'])
"
done
