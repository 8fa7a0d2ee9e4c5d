#!/bin/bash
# A/B library: the default body + an alternate one generated with ASMGEN_ALT (JSON: DkdvGen keyword arguments, and
# sched_<NAME> overrides of tools/asmgen/sched.py constants), compiled side by side (-DSFA_AB); tools/ab.py times them.
# usage: tools/build_ab.sh '{"npool": 8, "sched_LDS_LAT": 128}'      (no argument: back to the release build)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/sink-flash-attention-kernel_amd
rm -f $P/csrc/gen/*.inc $P/build/sfa_bwd_mfma.o $P/build/sfa_fwd_mfma.o
if [ -n "${1:-}" ]; then
  if [ "${2:-}" = "stamps" ]; then export ASMGEN_STAMPS=1; fi
  if [ "${2:-}" = "phases" ]; then export ASMGEN_STAMPS=phases; fi   # fenced s_memtime between the MFMA phases of a trip      # the ALT body carries s_memtime stamps (tools/stamps_dkdv.py)
  ASMGEN_ALT="$1" ASMGEN_ALT_DQ="${ASMGEN_ALT_DQ:-{\}}" ASMGEN_ALT_FWD="${ASMGEN_ALT_FWD:-{\}}" make -C $P -j8 EXTRA=-DSFA_AB 2>&1 | grep -E "error|Error|asmgen" || true
else
  make -C $P -j8 2>&1 | grep -E "error|Error|asmgen" || true
fi
ls -la $P/sink_attention/libsfa.so
