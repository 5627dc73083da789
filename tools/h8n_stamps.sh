# stamps of the narrow-tile kernel on two layers; usage (GPU box): tools/h8n_stamps.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${1:-x}
cd $R
python3 tools/h8_bench.py --iters 20 --only "64" > $O/${TAG}_h8n_layers.md 2>/dev/null
cp quber_amd/libquber_hip.so /tmp/libquber_hip.so.keep
(cd quber_amd/csrc && make -B conv_h8.o plan.o H8X=-DH8_STAMPS > /dev/null 2>&1 && make H8X=-DH8_STAMPS > /dev/null 2>&1)
(cd tools && python3 h8_stamps.py "stem.conv3" > $O/${TAG}_stamps_stem3.txt 2>/dev/null; python3 h8_stamps.py "res2.conv2" > $O/${TAG}_stamps_res2c2.txt 2>/dev/null)
cp /tmp/libquber_hip.so.keep quber_amd/libquber_hip.so
cat $O/${TAG}_h8n_layers.md; tail -5 $O/${TAG}_stamps_stem3.txt | cut -c1-260; tail -5 $O/${TAG}_stamps_res2c2.txt | cut -c1-260
