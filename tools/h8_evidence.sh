#!/bin/bash
# Evidence of the fp16 path's 256 x 256 LDS-DMA kernel (csrc/conv_h8.hip), on the GPU box:
#   usage: tools/h8_evidence.sh <tag>   -> gpurun_out/<tag>_h8_*.{md,txt}
# (the undilated 3x3 layers run on the patch kernels, option key 38: the stamps are taken on a dilated layer (DMA-gather kernel) and on a head layer (patch kernel))
# stand-alone layer tables (conv_igemm against conv_h8), PMC counters of two wide layers, in-kernel tile stamps
# (the stamps need a diagnostic build: tools/diag_build.sh makes it in a scratch copy of the sources and QUBER_LIB points the tools at it)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}; O=$R/gpurun_out
cd $R
python3 tools/h8_bench.py --iters 30 > $O/${TAG}_h8_layers.md 2>/dev/null
rm -f $O/${TAG}_h8a_pmc.txt $O/${TAG}_h8b_pmc.txt
tools/h8_pmc.sh ${TAG}_h8a "fusion_res2.conv0" > /dev/null 2>&1
tools/h8_pmc.sh ${TAG}_h8b "fusion_res5.conv" > /dev/null 2>&1
tools/h8_pmc.sh ${TAG}_h8c "fusion_res2.conv0" conv_igemm > /dev/null 2>&1
DL=$(tools/diag_build.sh h8stamps H8X=-DH8_STAMPS) || exit 1
(cd tools && QUBER_LIB=$DL python3 h8_stamps.py "res5.conv2" > $O/${TAG}_h8_stamps_res5_conv2_dma_gather.txt 2>/dev/null; QUBER_LIB=$DL python3 h8_stamps.py "head.0" > $O/${TAG}_h8_stamps_head0_patch.txt 2>/dev/null)
tail -4 $O/${TAG}_h8_layers.md; cat $O/${TAG}_h8a_pmc.txt | head -12; tail -3 $O/${TAG}_h8_stamps_head0_patch.txt | cut -c1-250
