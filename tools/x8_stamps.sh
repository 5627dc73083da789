R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${1:-x}
cd $R
DL=$(tools/diag_build.sh x8stamps X8X=-DX8_STAMPS) || exit 1
(cd tools && QUBER_LIB=$DL python3 x8_stamps.py "fusion_res5.conv" > $O/${TAG}_x8_stamps.txt 2>&1; QUBER_LIB=$DL python3 x8_stamps.py "wino GEMM 512" >> $O/${TAG}_x8_stamps.txt 2>&1)
cat $O/${TAG}_x8_stamps.txt | grep -v amdgpu.ids
