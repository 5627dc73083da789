# stamps of the narrow-tile kernel on two layers; usage (GPU box): tools/h8n_stamps.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${1:-x}
cd $R
python3 tools/h8_bench.py --iters 20 --only "64" > $O/${TAG}_h8n_layers.md 2>/dev/null
DL=$(tools/diag_build.sh h8stamps H8X=-DH8_STAMPS) || exit 1
(cd tools && QUBER_LIB=$DL python3 h8_stamps.py "stem.conv3" > $O/${TAG}_stamps_stem3.txt 2>/dev/null; QUBER_LIB=$DL python3 h8_stamps.py "res2.conv2" > $O/${TAG}_stamps_res2c2.txt 2>/dev/null)
cat $O/${TAG}_h8n_layers.md; tail -5 $O/${TAG}_stamps_stem3.txt | cut -c1-260; tail -5 $O/${TAG}_stamps_res2c2.txt | cut -c1-260
