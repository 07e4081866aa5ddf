timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_shells.py tests/test_gpu_frustum.py tests/test_gpu_fuzz.py tests/test_gpu_repeat.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -8
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
for s in 1 2 4 8; do echo split $s; YGZF_MATCH_SPLIT=$s YGZF_MATCH_DEBUG=1 /tmp/shell_latency /tmp 2 2>&1 | grep "ygzf match" | tail -1; YGZF_MATCH_SPLIT=$s /tmp/shell_latency /tmp 200 | grep search_by; done
timeout 200 python tools/call_latency.py 2>/dev/null | grep -i "search"
python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --batch 1 --sub-batch 1 --steps 200 --warmup 10 2>&1 | tail -1 | cut -c1-140
for b in 4 16 64; do python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --batch $b --sub-batch $b --steps 100 --warmup 10 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b',$b, d['ms_per_step'], d['value'])"; done
