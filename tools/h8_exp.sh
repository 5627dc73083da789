R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out
cd $R
DL=$(tools/diag_build.sh h8exp H8X=-DH8_EXPERIMENT) || exit 1
for D in 0 2 3; do echo "pk_debug $D"; QUBER_LIB=$DL python3 - <<P
import sys; sys.argv=['x','--only','3x3','--iters','20']
sys.path.insert(0,'$R/tools')
from quber_amd import _lib
_lib.load().quber_set_tuning(16, $D)
import h8_bench; h8_bench.main()
P
done
