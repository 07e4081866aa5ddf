timeout 900 python -m pytest tests/test_gpu_shells.py tests/test_gpu_boundary.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
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
/tmp/shell_latency /tmp 200
