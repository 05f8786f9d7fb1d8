#!/bin/bash
# attention backward variants (round 6): tools/_diag/libdevit_$v.so for every argument (default: the ablation set), two interleaved passes, cold
VARS=${@:-anov anodelta anovd anomain}
for i in 1 2; do
  timeout 120 python tools/attn_ab.py
  for v in $VARS; do DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so timeout 120 python tools/attn_ab.py; done
done
