R=${GRAFT_REPO_ROOT}; O=$R/gpurun_out
cd $R
DL=$(tools/diag_build.sh h8stamps H8X=-DH8_STAMPS) || exit 1
(cd tools && QUBER_LIB=$DL python3 h8_stamps.py "fusion_layers.1" 2>/dev/null | tail -4 | cut -c1-200; QUBER_LIB=$DL python3 h8_stamps.py "head.0" 2>/dev/null | tail -3 | cut -c1-200)
