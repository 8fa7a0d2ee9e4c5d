#!/bin/bash
# development: per-kernel times (tools/ab.py stage events) of every library variant under build_ab/, same box, same process order
R=$(cd $(dirname $0)/.. && pwd)
for cfg in ${CFGS:-C4 oss_swa}; do
  for f in $R/sink-flash-attention-kernel_amd/build_ab/libsfa_*.so; do
    n=$(basename $f .so); n=${n#libsfa_}
    echo -n "$cfg $n: "
    SFA_LIB_PATH=$f timeout -k 10 120 python $R/tools/ab.py --cfg $cfg --values 0 --rounds ${ROUNDS:-20} 2>&1 | grep "^variant" | cut -c1-140
  done
done
