#!/bin/bash
# development: libsfa.so variants whose short-window dK/dV bodies are generated with other parameters (knock-outs, scheduler
# knobs), for same-box comparisons through SFA_LIB_PATH (tools/kbench.py).  usage: tools/build_variants.sh name='{"json"}' ...
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/sink-flash-attention-kernel_amd
mkdir -p $P/build_ab
for spec in "$@"; do
  name=${spec%%=*}; js=${spec#*=}
  ASMGEN_ALT_SKEW="$js" make -C $P > /tmp/build_$name.log 2>&1 || { tail -5 /tmp/build_$name.log; exit 1; }
  cp $P/sink_attention/libsfa.so $P/build_ab/libsfa_$name.so
  echo "built $name"
done
ASMGEN_ALT_SKEW= make -C $P > /tmp/build_base.log 2>&1
echo "rebuilt the shipped library"
