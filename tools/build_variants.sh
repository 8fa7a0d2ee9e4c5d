#!/bin/bash
# development: libsfa.so variants whose hand-placed bodies are generated with other parameters (knock-outs, scheduler knobs), for
# same-box comparisons through SFA_LIB_PATH (tools/run_variants.sh).
# usage: tools/build_variants.sh name=kind:'{"json"}' ...     kind: skew (ASMGEN_ALT_SKEW: the short-window dK/dV bodies) or
#        fwd (ASMGEN_ALL_FWD: the work-list forward bodies)
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/sink-flash-attention-kernel_amd
mkdir -p $P/build_ab
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; kind=${rest%%:*}; js=${rest#*:}
  sk='{}'; fw='{}'
  if [ "$kind" = skew ]; then sk="$js"; else fw="$js"; fi
  ASMGEN_ALT_SKEW="$sk" ASMGEN_ALL_FWD="$fw" make -C $P > /tmp/build_$name.log 2>&1 || { tail -5 /tmp/build_$name.log; exit 1; }
  cp $P/sink_attention/libsfa.so $P/build_ab/libsfa_$name.so
  echo "built $name"
done
ASMGEN_ALT_SKEW= ASMGEN_ALL_FWD= make -C $P > /tmp/build_base.log 2>&1
echo "rebuilt the shipped library"
