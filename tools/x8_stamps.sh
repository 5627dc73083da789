R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${1:-x}
cd $R
cp quber_amd/libquber_hip.so /tmp/libquber_hip.so.keep
(cd quber_amd/csrc && make -B conv_x8.o plan.o X8X=-DX8_STAMPS > /dev/null 2>&1 && make X8X=-DX8_STAMPS > /dev/null 2>&1)
(cd tools && python3 x8_stamps.py "fusion_res5.conv" > $O/${TAG}_x8_stamps.txt 2>&1; python3 x8_stamps.py "wino GEMM 512" >> $O/${TAG}_x8_stamps.txt 2>&1)
cp /tmp/libquber_hip.so.keep quber_amd/libquber_hip.so
cat $O/${TAG}_x8_stamps.txt | grep -v amdgpu.ids
