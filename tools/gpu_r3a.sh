set -x
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_fuzz.py tests/test_gpu_shells.py -x -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -14
H=orb_ygz_slam_amd/csrc/host; L=orb_ygz_slam_amd/lib
g++ -std=c++17 -O2 -pthread -I $H -I $H/standalone tests/cpp/shell_latency.cc $H/ORBextractor.cc $H/ORBmatcher.cc $H/SparseImageAlign.cc $H/ygzf_pool.cc -L $L -lygzf -Wl,-rpath,$PWD/$L -o /tmp/shell_latency
python - <<'PY'
import sys
sys.path.insert(0,'.')
from orb_ygz_slam_amd.scene import two_view_scene
from orb_ygz_slam_amd import EUROC
a,b,_,_=two_view_scene(9,752,480,EUROC,Z=4.0)
a.tofile('/tmp/a.u8'); b.tofile('/tmp/b.u8')
PY
YGZF_SIA_DEBUG=1 /tmp/shell_latency /tmp 3 2>&1 | grep "ygzf sia" | tail -2
python bench.py --steps 3 --warmup 1 --passes 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('end_to_end',{})); 
for k,v in d.get('other_workloads',{}).items(): print(k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a in ('value','unit','ms_per_step','roofline')})"
