#!/bin/bash
# attention backward ablations (round 6): default library vs tools/_diag/libdevit_a{nov,nodelta,novd,nomain}.so, two interleaved passes, cold
for i in 1 2; do
  timeout 120 python tools/attn_ab.py
  for v in anov anodelta anovd anomain; do DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so timeout 120 python tools/attn_ab.py; done
done
