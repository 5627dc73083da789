R=${GRAFT_REPO_ROOT}; O=$R/gpurun_out
cd $R
cp quber_amd/libquber_hip.so /tmp/libquber_hip.so.keep
(cd quber_amd/csrc && make -B conv_h8.o plan.o H8X=-DH8_STAMPS > /dev/null 2>&1 && make H8X=-DH8_STAMPS > /dev/null 2>&1)
(cd tools && python3 h8_stamps.py "fusion_layers.1" 2>/dev/null | tail -4 | cut -c1-200; python3 h8_stamps.py "head.0" 2>/dev/null | tail -3 | cut -c1-200)
cp /tmp/libquber_hip.so.keep quber_amd/libquber_hip.so
