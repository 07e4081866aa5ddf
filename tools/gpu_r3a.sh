timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_golden.py tests/test_gpu_fast_plans.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
YGZF_FUZZ_SEEDS=64 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "special or carry" 2>&1 | grep -v "^$" | tail -2
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
YGZF_OCT_DEBUG=1 /tmp/shell_latency /tmp 2 2>&1 | grep "ygzf octree" | tail -8
python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --batch 1 --sub-batch 1 --steps 200 --warmup 10 2>&1 | tail -1 | cut -c1-140
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_isolated_avg_us'))"
