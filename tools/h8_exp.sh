R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out
cd $R
cp quber_amd/libquber_hip.so /tmp/libquber_hip.so.keep
(cd quber_amd/csrc && make -B conv_h8.o H8X=-DH8_EXPERIMENT > /dev/null 2>&1 && make H8X=-DH8_EXPERIMENT > /dev/null 2>&1)
for D in 0 2 3; do echo "pk_debug $D"; python3 - <<P
import sys; sys.argv=['x','--only','3x3','--iters','20']
sys.path.insert(0,'$R/tools')
from quber_amd import _lib
_lib.load().quber_set_tuning(16, $D)
import h8_bench; h8_bench.main()
P
done
cp /tmp/libquber_hip.so.keep quber_amd/libquber_hip.so
