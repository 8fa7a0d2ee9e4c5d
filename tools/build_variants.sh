#!/bin/bash
# Several A/B libraries side by side (one ALT body each): tools/build_variants.sh name1 'json1' name2 'json2' ...
# -> sink-flash-attention-kernel_amd/sink_attention/libsfa_<name>.so ; run with SFA_LIB_PATH=... python tools/ab.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/sink-flash-attention-kernel_amd
while [ $# -ge 2 ]; do
  name=$1; json=$2; shift 2
  rm -f $P/csrc/gen/*.inc $P/build/sfa_bwd_mfma.o $P/build/sfa_fwd_mfma.o
  ASMGEN_ALT="${ASMGEN_ALT_KIND_DKDV:+$json}" ASMGEN_ALT_DQ="${ASMGEN_ALT_KIND_DQ:+$json}" ASMGEN_ALT_FWD="${ASMGEN_ALT_KIND_FWD:+$json}" \
    make -C $P -j8 EXTRA=-DSFA_AB 2>&1 | grep -E "error|Error" || true
  cp $P/sink_attention/libsfa.so $P/sink_attention/libsfa_$name.so
  echo "built libsfa_$name.so ($json)"
done
rm -f $P/csrc/gen/*.inc $P/build/sfa_bwd_mfma.o $P/build/sfa_fwd_mfma.o
make -C $P -j8 2>&1 | grep -E "error|Error" || true
