cd /root/repo
cp annlite_amd/libannlite_hip.so /tmp/lib_orig.so
cp build_exp/lib_exp5.so annlite_amd/libannlite_hip.so
for rows in 1250000 10000000; do
ANNLITE_DEBUG_COUNTERS=1 timeout 120 python - <<PY
import sys, os
sys.argv=['prof_scan.py','--rows','$rows','--data','lowrank','--fused','--iters','5']
sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
exec(open('scripts/prof_scan.py').read().split("if os.environ.get('ANNLITE_DEBUG_COUNTERS')")[0])
from annlite_amd import _capi
c=_capi.debug_counters()
print('rows $rows phases per WG (us): pop %.1f exact %.1f queues %.1f rounds %.1f publish %.1f | total inside %.1f batches %d' % tuple([c[i]/256/2400. for i in (0,1,2,3,5,4)]+[c[6]]))
PY
done
cp /tmp/lib_orig.so annlite_amd/libannlite_hip.so
