#!/bin/bash
# Diagnostic build of libquber_hip.so in a SCRATCH COPY of the sources (never in quber_amd/csrc: objects built with -D..._STAMPS
# there would be newer than their sources and the next incremental `make` would link them into the product library).
#   usage: LIB=$(tools/diag_build.sh <name> VAR=-DFLAG ...)    e.g. tools/diag_build.sh h8stamps H8X=-DH8_STAMPS
#   then:  QUBER_LIB=$LIB python3 tools/h8_stamps.py ...       (quber_amd/_lib.py loads the library QUBER_LIB names)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
D=${TMPDIR:-/tmp}/quber_diag_$NAME
rm -rf $D && mkdir -p $D/quber_amd $D/include || exit 1
cp $R/include/*.h $D/include/ && mkdir $D/quber_amd/csrc && cp $R/quber_amd/csrc/*.hip $R/quber_amd/csrc/*.h $R/quber_amd/csrc/Makefile $D/quber_amd/csrc/ || exit 1
make -C $D/quber_amd/csrc -j16 "$@" > $D/build.log 2>&1 || { echo "diagnostic build failed: $D/build.log" >&2; exit 1; }
echo $D/quber_amd/libquber_hip.so
